"""End-to-end drop-in check: `node llama2.mjs <ckpt> ...` (JS host -> N-API -> C ABI -> HIP) must print
exactly what the TRUE reference printed for the same checkpoint, tokenizer, flags and seed.

The expected text comes from tests/golden/cli_*.json (oracle/make_goldens.py ran /root/reference/llama2.ts
under Node with the synthetic tokenizer of tests/synth_tokenizer.py in its working directory).  Covers the
greedy path, a BPE-encoded prompt, temperature sampling and top-p sampling (host sampler + RNG on GPU logits).
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
HOST = os.path.join(ROOT, "llama2.ts_amd", "host", "llama2.mjs")
TOKS = re.compile(r"\n\nachieved tok/s: [^\n]*\n\n$")


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    if shutil.which("node") is None:
        pytest.skip("no node on this box")
    import __graft_entry__ as graft
    graft.build()
    d = tmp_path_factory.mktemp("cli")
    meta = json.load(open(os.path.join(GOLD, "cli_greedy.json")))
    O.synth_write(meta["header"], meta["seed"], str(d / "model.bin"))
    synth_tokenizer.write(str(d / "tokenizer.bin"))
    return d


def run_cli(workdir, argv, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(["node", HOST, str(workdir / "model.bin"), *argv], cwd=str(workdir), env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return r.returncode, r.stdout.decode("utf8"), r.stderr.decode("utf8")


@pytest.mark.parametrize("name", ["cli_greedy", "cli_prompt", "cli_temp", "cli_topp"])
def test_cli_prints_what_the_reference_printed(workdir, name):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    rc, out, err = run_cli(workdir, meta["argv"])
    assert rc == 0, err
    assert TOKS.search(out), out[-80:]               # same trailer format as llama2.ts:511
    assert TOKS.sub("", out) == TOKS.sub("", meta["stdout"])


def test_cli_device_greedy_extra_prints_the_same(workdir):
    meta = json.load(open(os.path.join(GOLD, "cli_greedy.json")))
    rc, out, err = run_cli(workdir, meta["argv"], {"L2_DEVICE_GREEDY": "1"})
    assert rc == 0, err
    assert TOKS.sub("", out) == TOKS.sub("", meta["stdout"])


@pytest.mark.parametrize("name", ["cli_temp", "cli_topp"])
def test_cli_device_sampler_extra_prints_the_same(workdir, name):
    """L2_DEVICE_SAMPLER=1: temperature, softmax, sample / sample_topp and the RNG on the GPU (l2_decode_sample),
    with and without the batched prompt prefill in front of it."""
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    for extra in ({}, {"L2_PREFILL": "1"}):
        rc, out, err = run_cli(workdir, meta["argv"], dict({"L2_DEVICE_SAMPLER": "1"}, **extra))
        assert rc == 0, err
        assert TOKS.sub("", out) == TOKS.sub("", meta["stdout"])


def test_cli_stats_line_goes_to_stderr_only(workdir):
    meta = json.load(open(os.path.join(GOLD, "cli_greedy.json")))
    rc, out, err = run_cli(workdir, meta["argv"], {"L2_STATS": "1"})
    assert rc == 0, err
    assert TOKS.sub("", out) == TOKS.sub("", meta["stdout"])          # stdout untouched
    stats = json.loads(err.strip().splitlines()[-1])
    assert stats["tokens_timed"] == meta["steps_run"] - 1 and stats["algorithmic_bytes_per_token"] > 60_000_000
    assert 0 < stats["hbm_frac_of_8tbs"] < 1


def test_cli_usage_and_errors(workdir):
    r = subprocess.run(["node", HOST], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stderr.decode().startswith("Usage: ... llama2.ts <checkpoint> [options]")
    rc, _, err = run_cli(workdir, ["-x", "1"])
    assert rc == 1 and "Usage:" in err               # unknown flag (llama2.ts:421)
    rc, _, err = run_cli(workdir, ["-t"])
    assert rc == 1 and "Usage:" in err               # flag without value (llama2.ts:410)
    rc, _, err = run_cli(workdir, ["-t", "0", "-i", "中"])
    assert rc == 1 and "character not found in vocab" in err   # llama2.ts:310


def test_cli_native_loader_prints_the_same(workdir):
    meta = json.load(open(os.path.join(GOLD, "cli_prompt.json")))
    rc, out, err = run_cli(workdir, meta["argv"], {"L2_NATIVE_LOADER": "1"})
    assert rc == 0, err
    assert TOKS.sub("", out) == TOKS.sub("", meta["stdout"])


@pytest.mark.parametrize("name", ["cli_prompt", "cli_topp"])
def test_cli_batched_prefill_prints_the_same(workdir, name):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    rc, out, err = run_cli(workdir, meta["argv"], {"L2_PREFILL": "1"})
    assert rc == 0, err
    assert TOKS.sub("", out) == TOKS.sub("", meta["stdout"])
