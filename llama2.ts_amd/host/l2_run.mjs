// @ts-check
// l2_run.mjs -- drive the forward pass from Node with TOKEN IDS in and out (no tokenizer, no text):
//
//   node l2_run.mjs <checkpoint> [--steps N] [--prompt 12,7,99] [--temperature T] [--topp P] [--seed S]
//                   [--loop host|device] [--prefill] [--native-loader] [--metrics]
//
// Prints ONE JSON line {"tokens":[...], "tok_s":...}: the ids the reference's loop (llama2.ts:465-508) would have
// printed as text for the same checkpoint, prompt, flags and seed -- tests/test_cli_gpu.py turns them into text with
// the tokenizer's vocabulary and compares with what the reference printed.
//   --metrics      also print ONE JSON line on STDERR next to where the reference prints its tok/s (llama2.ts:511): tok/s, the
//                  algorithmic bytes one token streams (SURVEY.md 8(d)), GB/s and the fraction of the 8 TB/s HBM peak;
//   --loop host    one transformer() call per position through the drop-in boundary, greedy pick here (temperature 0);
//   --loop device  the library's device-resident loops: l2_decode_greedy, or l2_decode_sample when temperature > 0
//                  (temperature / softmax / sample / top-p / RNG on the GPU, same ids as the reference for the seed).
import { openBackend, loadModel, transformer, algorithmicBytesPerToken } from "./l2_backend.mjs";

function options(argv) {
  const o = { steps: 256, prompt: [], temperature: 0, topp: 1, seed: 1n, loop: "host", prefill: false, nativeLoader: false, metrics: false };
  for (let i = 0; i < argv.length; ++i) {
    const k = argv[i];
    if (k == "--prefill") o.prefill = true;
    else if (k == "--native-loader") o.nativeLoader = true;
    else if (k == "--metrics") o.metrics = true;
    else if (i + 1 >= argv.length) throw new Error("missing value after " + k);
    else if (k == "--steps") o.steps = parseInt(argv[++i]);
    else if (k == "--prompt") o.prompt = argv[++i].split(",").filter((t) => t.length).map((t) => parseInt(t));
    else if (k == "--temperature") o.temperature = parseFloat(argv[++i]);
    else if (k == "--topp") o.topp = parseFloat(argv[++i]);
    else if (k == "--seed") o.seed = BigInt(argv[++i]);
    else if (k == "--loop") o.loop = argv[++i];
    else throw new Error("unknown option " + k);
  }
  return o;
}

function firstMaximum(values) {          // the reference's argmax keeps the first of equal maxima (llama2.ts:364-366)
  let at = 0;
  for (let i = 1; i < values.length; ++i) if (values[i] > values[at]) at = i;
  return at;
}

function run() {
  const [, , file, ...rest] = process.argv;
  if (!file) throw new Error("usage: node l2_run.mjs <checkpoint> [--steps N] [--prompt ids] [--temperature T] [--topp P] [--seed S] [--loop host|device] [--prefill] [--native-loader] [--metrics]");
  const o = options(rest);
  const be = openBackend();
  const { config, weights, state } = loadModel(file, be, { device: parseInt(process.env.L2_DEVICE || "0"), nativeLoader: o.nativeLoader });
  const steps = (o.steps <= 0 || o.steps > config.seq_len) ? config.seq_len : o.steps;
  const out = [];
  let token = 1, pos = 0, t0 = 0, pos_t0 = 0;          // position 0 is fed BOS (llama2.ts:463)
  // The clock starts where the reference starts its own: AFTER the first iteration of the loop (llama2.ts:507 -- "the first iteration
  // can be slower": here it holds the graph capture), whatever that iteration was (a prompt position or a sampled one); the rate is
  // then (positions - 1) / elapsed like llama2.ts:511.
  const startClock = () => { if (!t0 && pos > 0) { t0 = Date.now(); pos_t0 = pos; } };

  // the teacher-forced prompt positions (llama2.ts:471-473): one transformer() each, or one batched l2_prefill
  const forced = Math.min(o.prompt.length, steps);
  if (o.prefill && forced > 1) {
    be.prefill(weights.ctx, Int32Array.from([1, ...o.prompt.slice(0, forced - 1)]), 0, null);
    for (; pos < forced; ++pos) out.push(token = o.prompt[pos]);
    startClock();                                       // (a batched prompt is one "first iteration")
  }
  for (; pos < forced; ) { transformer(token, pos, config, state, weights, be); out.push(token = o.prompt[pos]); ++pos; startClock(); }

  const seed = new Uint32Array([Number(o.seed & 0xffffffffn), Number(o.seed >> 32n)]);
  while (pos < steps) {
    let ids;
    if (o.loop == "device") {
      // one position alone while the clock is not running yet (the reference's first iteration), then the rest in calls of up to 256
      // positions: the loop stays on the device, the host only looks for BOS (llama2.ts:499) in what comes back
      const n = t0 ? Math.min(256, steps - pos) : 1;
      ids = o.temperature == 0 ? be.decodeGreedy(weights.ctx, token, pos, n) : be.decodeSample(weights.ctx, token, pos, n, o.temperature, o.topp, seed);
    } else {
      if (o.temperature != 0) throw new Error("--loop host picks greedily; sampling runs on the device (--loop device)");
      transformer(token, pos, config, state, weights, be);
      ids = [firstMaximum(state.logits)];
    }
    let stop = false;
    for (const id of ids) {
      ++pos;
      if (id == 1) { stop = true; break; }   // BOS ends the sequence (llama2.ts:499)
      out.push(token = id);
    }
    if (stop) break;
    startClock();
  }
  const ms = t0 ? Date.now() - t0 : 0;
  if (o.metrics) {
    // what the timed positions streamed, by SURVEY.md 8(d)'s count (weights + KV rows of the position + logits), against the HBM peak
    let bytes = 0;
    for (let p = pos_t0; p < pos; ++p) bytes += algorithmicBytesPerToken(config, p);
    const n = pos - pos_t0, sec = ms / 1000;
    process.stderr.write(JSON.stringify({ metrics: { tokens_timed: n, tok_s: sec > 0 ? n / sec : null, algorithmic_bytes_per_token: n ? Math.round(bytes / n) : null,
      hbm_gb_s: sec > 0 ? bytes / sec / 1e9 : null, hbm_peak_gb_s: 8000, hbm_frac: sec > 0 ? bytes / sec / 1e9 / 8000 : null,
      loop: o.loop, timer: "Date.now(), started after the first iteration like llama2.ts:507; first timed position " + pos_t0,
      sampler: o.loop == "device" && o.temperature != 0 ? { tokens: be.getOption(weights.ctx, 6), by_serial_loop: be.getOption(weights.ctx, 7) } : undefined } }) + "\n");
  }
  be.destroy(weights.ctx);
  process.stdout.write(JSON.stringify({ tokens: out, tok_s: ms > 0 ? (pos - pos_t0) / ms * 1000 : null }) + "\n");
}

try {
  run();
} catch (e) {
  console.error(String(e && e.message ? e.message : e));
  process.exit(1);
}
