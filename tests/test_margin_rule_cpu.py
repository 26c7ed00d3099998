"""CPU check of the device sampler's margin rule (llama2.ts_amd/csrc/margin_rule.h, used by csrc/sampler_margin.hip.h): tree sums plus a
proven margin decide which index the reference's sequential loops return (llama2.ts:368-394); what they cannot decide goes to the loop
itself.  tests/margin_rule_host.cc runs the same inline functions the kernels call against the loops written out, on softmax-like and
adversarial vectors, with the tree total pushed to the edge of what the rule allows for: a decided index is never different from the
loop's, every probability a different total rounds differently is accounted for, and the rule decides nearly always."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("mr") / "margin_rule_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "margin_rule_host.cc")])
    return exe


def _exps(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "softmax":                      # logits ~ N(0, 3) at temperature 0.9
        x = (rng.standard_normal(n) * 3.0).astype(np.float32)
    elif kind == "peaked":                     # one token far ahead, a long tail down to subnormal exps
        x = (rng.standard_normal(n) * 12.0).astype(np.float32)
        x[rng.integers(n)] += 40.0
    elif kind == "flat":                       # a huge temperature: thousands of EQUAL probabilities
        x = (rng.standard_normal(n) * 1e-6).astype(np.float32)
    elif kind == "dyadic":                     # exps that are powers of two: every sum exact, thresholds land ON running sums
        x = (np.log(2.0) * rng.integers(-12, 1, n)).astype(np.float32)
        e = np.exp2(np.round(x / np.log(2.0))).astype(np.float32)
        e[0] = 1.0
        return e
    x = (x.astype(np.float64) / 0.9).astype(np.float32)
    return np.exp(x.astype(np.float64) - float(x.max())).astype(np.float32)


@pytest.mark.parametrize("kind,n", [("softmax", 32000), ("softmax", 1000), ("softmax", 128256), ("peaked", 32000), ("flat", 32000), ("dyadic", 4096),
                                    ("dyadic", 32000), ("softmax", 5121)])
def test_a_decided_index_is_the_loops_index(harness, tmp_path, kind, n):
    src = str(tmp_path / "e.f32")
    for seed in (1, 2):
        _exps(kind, n, seed).tofile(src)
        for skew in (0.0, 1.0, -1.0):
            r = subprocess.run([harness, src, "400", str(seed * 7919), str(skew)], capture_output=True, text=True)
            assert r.returncode == 0, (kind, n, seed, skew, r.stdout, r.stderr)
            f = r.stdout.split()
            decided, undecided, wrong, tdec, tund, twrong, und_random, tund_random = (int(v) for v in f[3:11])
            assert wrong == 0 and twrong == 0
            if kind in ("softmax", "peaked"):  # thresholds that were not aimed at a running sum are all decided; most of the aimed ones too (sample: to within 2^-25)
                assert und_random == 0 and tund_random == 0 and undecided <= 40, r.stdout
            if kind == "dyadic":               # thresholds ON a running sum: the rule must step aside, not guess
                assert undecided > 0 or tund > 0, r.stdout
