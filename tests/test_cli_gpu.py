"""End-to-end drop-in check through the JavaScript boundary: Node -> N-API addon -> C ABI -> HIP.

`llama2.ts_amd/host/l2_run.mjs` drives the forward pass with token ids in and out (the tokenizer, the RNG and the
printing of the reference's CLI are host-only code outside the hot path).  The ids it returns,
turned into text here with the tokenizer's vocabulary and the reference's printing rule (llama2.ts:502: the piece
after BOS loses one leading space), must be exactly what the TRUE reference printed for the same checkpoint, prompt,
flags and seed -- tests/golden/cli_*.json holds that stdout (oracle/make_goldens.py ran /root/reference/llama2.ts
under Node with the synthetic tokenizer of tests/synth_tokenizer.py in its working directory).
Covers: the per-token drop-in call with a greedy pick, a teacher-forced prompt, the device-resident greedy loop, the
device sampler (temperature, top-p, RNG), the batched prompt prefill and the native checkpoint loader.
"""
import json
import os
import re
import shutil
import subprocess

import pytest

import oracle_lib as O
import synth_tokenizer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
HOST = os.path.join(ROOT, "llama2.ts_amd", "host", "l2_run.mjs")
TRAILER = re.compile(r"\n\nachieved tok/s: [^\n]*\n\n$")
VOCAB, _ = synth_tokenizer.build_vocab()


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    if shutil.which("node") is None:
        pytest.skip("no node on this box")
    import __graft_entry__ as graft
    graft.build()
    d = tmp_path_factory.mktemp("cli")
    meta = json.load(open(os.path.join(GOLD, "cli_greedy.json")))
    O.synth_write(meta["header"], meta["seed"], str(d / "model.bin"))
    return d


def reference_run(name):
    """(flags of the reference's run, prompt ids, the text it printed without the tok/s trailer)."""
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    flags = dict(zip(meta["argv"][::2], meta["argv"][1::2]))
    fed = meta["tokens_fed"]
    prompt_ids = []
    if meta["prompt"] is not None:      # the teacher-forced ids are the fed tokens whose pieces spell the prompt (llama2.ts:471-473)
        text = ""
        for t in fed[1:]:
            if text == meta["prompt"]:
                break
            text += VOCAB[t]
            prompt_ids.append(t)
        assert text == meta["prompt"]
    return flags, prompt_ids, TRAILER.sub("", meta["stdout"])


def text_of(ids):
    out, prev = "", 1
    for t in ids:
        piece = VOCAB[t]
        out += piece[1:] if (prev == 1 and piece.startswith(" ")) else piece     # llama2.ts:502
        prev = t
    return out


def run_ids(workdir, argv, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(["node", HOST, str(workdir / "model.bin"), *argv], cwd=str(workdir), env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    out = r.stdout.decode("utf8")
    return r.returncode, (json.loads(out) if r.returncode == 0 else None), r.stderr.decode("utf8")


def argv_for(flags, prompt_ids, loop, extra=()):
    argv = ["--steps", flags["-n"], "--temperature", flags.get("-t", "1.0"), "--topp", flags.get("-p", "1.0"),
            "--seed", flags.get("-s", "1"), "--loop", loop, *extra]
    if prompt_ids:
        argv += ["--prompt", ",".join(map(str, prompt_ids))]
    return argv


@pytest.mark.parametrize("name", ["cli_greedy", "cli_prompt"])
def test_drop_in_call_per_token_prints_what_the_reference_printed(workdir, name):
    flags, prompt_ids, want = reference_run(name)
    rc, res, err = run_ids(workdir, argv_for(flags, prompt_ids, "host"))
    assert rc == 0, err
    assert text_of(res["tokens"]) == want


@pytest.mark.parametrize("name", ["cli_greedy", "cli_prompt", "cli_temp", "cli_topp"])
@pytest.mark.parametrize("extra", [(), ("--prefill",)])
def test_device_loops_print_what_the_reference_printed(workdir, name, extra):
    """l2_decode_greedy / l2_decode_sample (temperature, softmax, sample / sample_topp and the RNG on the GPU), with
    and without the batched prompt prefill in front."""
    flags, prompt_ids, want = reference_run(name)
    rc, res, err = run_ids(workdir, argv_for(flags, prompt_ids, "device", extra))
    assert rc == 0, err
    assert text_of(res["tokens"]) == want


def test_native_loader_prints_the_same(workdir):
    flags, prompt_ids, want = reference_run("cli_prompt")
    rc, res, err = run_ids(workdir, argv_for(flags, prompt_ids, "host", ("--native-loader",)))
    assert rc == 0, err
    assert text_of(res["tokens"]) == want


def test_metrics_line_on_stderr(workdir):
    """--metrics: one JSON line on stderr (tok/s, algorithmic bytes per token, GB/s, fraction of the 8 TB/s peak) -- stdout stays the ids."""
    from llama2_ts_amd import configs
    flags, prompt_ids, want = reference_run("cli_greedy")
    for loop in ("host", "device"):
        rc, res, err = run_ids(workdir, argv_for(flags, prompt_ids, loop, ("--metrics",)))
        assert rc == 0, err
        assert text_of(res["tokens"]) == want
        m = json.loads([l for l in err.splitlines() if l.startswith("{")][-1])["metrics"]
        hdr = json.load(open(os.path.join(GOLD, "cli_greedy.json")))["header"]
        n = m["tokens_timed"]
        # the clock starts after the first iteration, like the reference's (llama2.ts:507): positions 1 .. n are the timed ones
        assert n == int(flags["-n"]) - 1 and m["loop"] == loop and m["hbm_peak_gb_s"] == 8000 and "llama2.ts:507" in m["timer"]
        want_bytes = sum(configs.algorithmic_bytes_per_token(tuple(hdr), p) for p in range(1, n + 1)) / n
        assert abs(m["algorithmic_bytes_per_token"] - want_bytes) <= 1
        assert m["tok_s"] > 0 and abs(m["hbm_gb_s"] - m["tok_s"] * m["algorithmic_bytes_per_token"] / 1e9) / m["hbm_gb_s"] < 1e-3
        assert abs(m["hbm_frac"] - m["hbm_gb_s"] / 8000) < 1e-9


def test_metrics_line_counts_the_sampler(workdir):
    """A sampled device loop with --metrics also says how its tokens were picked (getOption 6 / 7 through the addon): every sampled token by
    the margin rule here, none by the serial loop -- and the text is still the reference's."""
    for name in ("cli_temp", "cli_topp"):
        flags, prompt_ids, want = reference_run(name)
        rc, res, err = run_ids(workdir, argv_for(flags, prompt_ids, "device", ("--metrics",)))
        assert rc == 0, err
        assert text_of(res["tokens"]) == want
        m = json.loads([l for l in err.splitlines() if l.startswith("{")][-1])["metrics"]
        assert m["sampler"]["tokens"] == int(flags["-n"]) - len(prompt_ids) and m["sampler"]["by_serial_loop"] == 0, m


def test_offset_views_through_the_real_addon(workdir, tmp_path):
    """llama2.ts:56 makes its Float32Arrays as views (buffer.buffer, buffer.byteOffset, n): the addon must hand the VIEW's bytes to
    l2_upload.  Arrays with byteOffset != 0 (a plain ArrayBuffer view, and a pooled Buffer the way the reference builds it) go up
    through the real addon into HBM and come back through readTensor; the bytes around the views are poison."""
    js = tmp_path / "offset.js"
    js.write_text("""
const a = require(%r); a.open(%r);
const ctx = a.create(new Int32Array([64, 176, 2, 4, 4, 512, 64]), 0);
const n = 64 * 64, off = 40;
const buf = new ArrayBuffer(off + n * 4 + 24);
new Float32Array(buf, 0, 10).fill(9e9); new Float32Array(buf, off + n * 4, 6).fill(-9e9);
const v = new Float32Array(buf, off, n);
for (let i = 0; i < n; ++i) v[i] = i * 0.25 - 3;
a.upload(ctx, 2, 1, v);                                        // wq[1]
const back = new Float32Array(new ArrayBuffer(n * 4 + 8), 8, n);
a.readTensor(ctx, 2, 1, 0, back);
let bad = 0;
for (let i = 0; i < n; ++i) if (back[i] !== v[i]) ++bad;
// the reference's own construction on a pooled Buffer (small allocUnsafe buffers share one ArrayBuffer: byteOffset != 0)
let b = Buffer.allocUnsafe(64 * 4), tries = 0;
while (b.byteOffset == 0 && ++tries < 64) b = Buffer.allocUnsafe(64 * 4);
const w = new Float32Array(b.buffer, b.byteOffset, 64);
for (let i = 0; i < 64; ++i) w[i] = 1 + i / 64;
a.upload(ctx, 1, 0, w);                                        // rms_att_weight[0]
const back2 = new Float32Array(64);
a.readTensor(ctx, 1, 0, 0, back2);
let bad2 = 0;
for (let i = 0; i < 64; ++i) if (back2[i] !== w[i]) ++bad2;
const tail = new Float32Array(3);
a.readTensor(ctx, 2, 1, n - 3, tail);                          // an offset read of the tensor's end
console.log(JSON.stringify({ viewOffset: v.byteOffset, bad, pooledOffset: w.byteOffset, bad2, tail: Array.from(tail), want: [v[n - 3], v[n - 2], v[n - 1]] }));
a.destroy(ctx);
""" % (os.path.join(ROOT, "llama2.ts_amd", "host", "l2_napi.node"), os.path.join(ROOT, "llama2.ts_amd", "lib", "libllama2hip.so")))
    r = subprocess.run(["node", str(js)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    j = json.loads(r.stdout.decode())
    assert j["viewOffset"] == 40 and j["bad"] == 0 and j["pooledOffset"] != 0 and j["bad2"] == 0 and j["tail"] == j["want"]


def test_errors_surface_as_exit_code_one(workdir):
    r = subprocess.run(["node", HOST], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and "usage" in r.stderr.decode()
    rc, _, err = run_ids(workdir, ["--bogus", "1"])
    assert rc == 1 and "unknown option" in err
    rc, _, err = run_ids(workdir, ["--steps", "4", "--prompt", "99999999"])
    assert rc == 1 and "token" in err                       # the library rejects ids outside the vocabulary
    rc, _, err = run_ids(workdir, ["--steps", "4", "--temperature", "0.5"])
    assert rc == 1 and "device" in err                      # host loop picks greedily only
    r = subprocess.run(["node", HOST, str(workdir / "missing.bin")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1
