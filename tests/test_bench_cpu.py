"""bench.py without a GPU: the launcher half of `--gpus N` and the honesty of the failure path.  On a box with no gfx950 device
the ranks cannot create a context, so there is NO result line and the exit code is not 0 -- the benchmark never prints a number
it did not measure (there is no CPU fallback behind it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible here: tests/test_bench_gpu.py covers the launcher with real ranks")
def test_gpus_flag_starts_ranks_and_relays_their_failure():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "stories15M", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]          # no JSON line was invented
    err = r.stderr.decode()
    assert "torch.distributed" in err or "ChildFailedError" in err or "no HIP device" in err  # the child launcher ran and its ranks said why


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible here")
def test_single_process_fails_loudly_without_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "stories15M", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"no HIP device visible" in r.stderr
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


def test_reference_js_figures_are_fixtures_with_provenance():
    """cpu_baseline.reference_js_tok_s comes from tests/golden/reference_speed.json (oracle/make_goldens.py --speed): every entry says
    which CPU / Node / argv produced it."""
    sys.path.insert(0, ROOT)
    import bench
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_speed.json")))
    for name in ("stories15M", "stories110M", "llama2_7b"):
        assert ref[name]["tok_s"] > 0 and ref[name]["threads"] == 1 and ref[name]["node"].startswith("v") and ref[name]["cpu"]
        got = bench.reference_js_figure(name)
        assert got["reference_js_tok_s"] == round(ref[name]["tok_s"], 4) and "build container" in got["reference_js_measured"]
    assert bench.reference_js_figure("tiny") == {"reference_js_tok_s": None}
