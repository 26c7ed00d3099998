// Drives the ASan/UBSan build of the N-API addon against the host-memory stub library (tests/test_sanitizers_cpu.py).
// argv: <addon.node> <stub.so> <tiny checkpoint file>.  Prints "ok <n checks>" and exits 0; any sanitizer report aborts the process.
const [, , addonPath, libPath, ckpt] = process.argv;
const a = require(addonPath);
let checks = 0;
function ok(c, what) { if (!c) { console.error("FAILED: " + what); process.exit(2); } ++checks; }
function throws(fn, part, what) {
  try { fn(); } catch (e) { ok(String(e.message).includes(part), what + ": message was " + e.message); return; }
  ok(false, what + ": did not throw");
}
a.open(libPath);
const hdr = new Int32Array([64, 176, 2, 4, 4, 512, 64]);
const ctx = a.create(hdr, 0);
const V = 512, d = 64, h = 176, S = 64, hs2 = 8;
const counts = [V * d, d, d * d, d * d, d * d, d * d, d, h * d, h * d, h * d, d, S * hs2, S * hs2, V * d];
const layered = [0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0];
// every array is a VIEW with byteOffset != 0 into a larger buffer whose other bytes are poison (llama2.ts:56: getF32Array returns
// new Float32Array(buffer.buffer, buffer.byteOffset, ...)): the addon must hand over the view's bytes, not the buffer's
const logits = new Float32Array(V);
for (let kind = 0; kind < 14; ++kind) {
  for (let l = 0; l < (layered[kind] ? 2 : 1); ++l) {
    const n = counts[kind], off = 4 * (3 + kind);            // 12, 16, 20, ... bytes
    const buf = new ArrayBuffer(off + n * 4 + 64);
    new Float32Array(buf).fill(1e30);                          // poison
    const view = new Float32Array(buf, off, n);
    for (let i = 0; i < n; ++i) view[i] = (kind + 1) * 0.5 + i * 1e-6;
    ok(view.byteOffset == off, "byteOffset");
    a.upload(ctx, kind, layered[kind] ? l : -1, view);
    a.forward(ctx, 1, 0, logits);                              // the stub adds the first element of the last upload to every logit
    ok(Math.abs(logits[0] - ((7 % 1009) * 0.001 + (kind + 1) * 0.5)) < 1e-4, "upload of kind " + kind + " read the view, not the buffer start (" + logits[0] + ")");
  }
}
// a Buffer-backed view the way the reference makes them
{
  const b = Buffer.alloc(16 + d * 4);
  const view = new Float32Array(b.buffer, b.byteOffset + 16, d);
  view.fill(2.25);
  a.upload(ctx, 10, -1, view);
  a.forward(ctx, 2, 5, logits);
  ok(Math.abs(logits[3] - (((3 * 31 + 2 * 7 + 5) % 1009) * 0.001 + 2.25)) < 1e-4, "Buffer-backed view");
}
throws(() => a.upload(ctx, 99, -1, new Float32Array(4)), "tensor kind out of range", "kind out of range");
throws(() => a.upload(ctx, 2, 0, new Float32Array(d * d - 1)), "wrong float count", "short array");
throws(() => a.upload(ctx, 2, 0, new Int32Array(d * d)), "expected a Float32Array", "wrong array type");
throws(() => a.upload(ctx, 2, 0, null), "expected a Float32Array", "null array");
throws(() => a.upload({}, 2, 0, new Float32Array(d * d)), "expected a context handle", "not a handle");
throws(() => a.upload(ctx, "x", 0, new Float32Array(d * d)), "expected an integer", "kind not a number");
throws(() => a.forward(ctx, 1, 0, new Float32Array(V - 1)), "logits array too small", "short logits array is refused before the library writes");
throws(() => a.forward(ctx, 1, 0, new Float32Array(new ArrayBuffer(V * 4 + 8), 8, V - 1)), "logits array too small", "short logits view");
throws(() => a.forward(ctx, 1, S, logits), "pos out of range", "pos");
throws(() => a.forward(ctx, V, 0, logits), "token out of range", "token");
a.forward(ctx, 1, 0, null);
a.forward(ctx, 1, 0, undefined);
{ // a logits VIEW with an offset: exactly V floats from byte 32
  const buf = new ArrayBuffer(32 + V * 4);
  const lv = new Float32Array(buf, 32, V);
  a.forward(ctx, 3, 1, lv);
  ok(new Float32Array(buf, 0, 8).every((x) => x === 0), "bytes in front of the logits view untouched");
}
{ // prefill: Int32Array view with an offset, logits out
  const tb = new ArrayBuffer(8 + 5 * 4);
  const toks = new Int32Array(tb, 8, 5);
  toks.set([1, 5, 9, 2, 7]);
  a.prefill(ctx, toks, 0, logits);
  a.prefill(ctx, toks, 3, null);
  throws(() => a.prefill(ctx, toks, S - 2, logits), "positions out of range", "prefill past the context");
  throws(() => a.prefill(ctx, new Float32Array(5), 0, logits), "expected Int32Array tokens", "prefill tokens type");
  throws(() => a.prefill(ctx, toks, 0, new Float32Array(7)), "logits array too small", "prefill short logits");
  throws(() => a.prefill(ctx, new Int32Array(0), 0, logits), "positions out of range", "empty prompt");
}
{ // device loops
  const t = a.decodeGreedy(ctx, 1, 0, 16);
  ok(t instanceof Int32Array && t.length == 16 && t[15] == 16, "decodeGreedy");
  ok(a.decodeGreedy(ctx, 1, 0, 0).length == 0, "zero steps");
  throws(() => a.decodeGreedy(ctx, 1, 60, 16), "positions out of range", "decodeGreedy past the context");
  throws(() => a.decodeGreedy(ctx, 1, 0, -1), "steps < 0", "negative steps");
  const rng = new Uint32Array([42, 0]);
  const s = a.decodeSample(ctx, 1, 0, 8, 0.9, 0.9, rng);
  ok(s.length == 8 && (rng[0] != 42 || rng[1] != 0), "decodeSample advances the state in place");
  throws(() => a.decodeSample(ctx, 1, 0, 8, 0.9, 0.9, new Uint32Array(1)), "Uint32Array of length 2", "short rng");
  throws(() => a.decodeSample(ctx, 1, 0, 8, "hot", 0.9, rng), "must be numbers", "temperature type");
}
{ // state reads
  const x = new Float32Array(d);
  a.readState(ctx, 0, -1, x);
  ok(x[d - 1] == d - 1, "readState");
  throws(() => a.readState(ctx, 0, -1, new Float32Array(d + 1)), "wrong float count", "readState size");
  const tb = new ArrayBuffer(16 + 10 * 4);
  const tv = new Float32Array(tb, 16, 10);
  a.readTensor(ctx, 2, 0, d * d - 10, tv);
  ok(tv[9] == d * d - 1 && new Float32Array(tb, 0, 4).every((x) => x === 0), "readTensor into an offset view");
  throws(() => a.readTensor(ctx, 2, 0, d * d - 9, tv), "read past the end", "readTensor bounds");
  throws(() => a.readTensor(ctx, 2, 0, -1, tv), "non-negative", "readTensor negative offset");
  a.setOption(ctx, 1, 1);
  throws(() => a.setOption(ctx, 77, 1), "unknown option", "option key");
  if (a.getOption(ctx, 6) !== 106) throw new Error("getOption: wrong value");
  throws(() => a.getOption(ctx, 77), "unknown option", "read of an unknown option");
}
{ // native loader entry
  const r = a.loadCheckpoint(ckpt, 0);
  ok(r.header instanceof Int32Array && r.header.length == 7 && r.header[0] == 64, "loadCheckpoint header");
  a.destroy(r.ctx);
  throws(() => a.loadCheckpoint(ckpt + ".missing", 0), "cannot open checkpoint", "missing file");
}
throws(() => a.create(new Int32Array(6), 0), "expected Int32Array(7)", "short header");
throws(() => a.create(new Int32Array([64, 176, 2, 5, 4, 512, 64]), 0), "bad header", "library refusal surfaces as an Error");
// zero-copy logits view, then destroy while the view is alive, then use-after-destroy
const ab = a.logitsBuffer(ctx, V);
const lview = new Float32Array(ab);
a.forward(ctx, 4, 2, null);
ok(lview.length == V && lview[1] > 0, "logitsBuffer view sees the forward");
if (process.env.L2_STUB_OVERRUN_CALL) a.forward(ctx, 1, 0, new Float32Array(V));   // negative control: the stub writes V + 1 floats
a.destroy(ctx);
ok(lview[1] > 0, "a live view keeps the memory alive after destroy()");
throws(() => a.forward(ctx, 1, 0, logits), "context already destroyed", "use after destroy");
a.destroy(ctx);                                                 // twice is harmless
console.log("ok " + checks);
