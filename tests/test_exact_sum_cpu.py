"""CPU check of the arithmetic the device sampler's running sums rest on (llama2.ts_amd/csrc/exact_sum.h): regular /
serial classification from an approximate prefix, grid composites with tie parity, the checked chain over the runs.
tests/exact_sum_host.cc drives the same functions the kernels call with plain loops; the result must be BIT-identical
to the reference's serial `cumProb += x[i]` loop (llama2.ts:369-373, :384-391) on the adversarial vectors the GPU test
uses, for several tile sizes, with the approximate prefix perturbed (another summation order) and with predictions
falsified on purpose (the chain's check has to catch them)."""
import os
import subprocess

import numpy as np
import pytest

import sum_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("xs") / "exact_sum_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "exact_sum_host.cc")])
    return exe


def test_run_chain_arithmetic_is_bit_identical_to_the_serial_loop(harness, tmp_path):
    cases = sum_cases.adversarial()
    src, dst = str(tmp_path / "in.f32"), str(tmp_path / "out.f64")
    for n, v in enumerate(cases):
        want = sum_cases.serial_sums(v)
        v.tofile(src)
        for tile, noise, sabotage, mb in ((1024, 0, 0, 32), (1024, 77, 0, 32), (256, 0, 0, 32), (64, 12345, 0, 32), (1024, 0, 3, 32), (256, 9, 2, 32),
                                          (1024, 0, 0, 20), (1024, 4242, 0, 20)):
            out = subprocess.check_output([harness, src, dst, str(tile), str(noise), str(sabotage), str(mb)]).decode()
            got = np.fromfile(dst, dtype=np.float64)
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (n, tile, noise, sabotage, mb, int(np.argmax(got != want)), out)
            if mb == 20 and n in (0, 1, 2):
                assert int(out.split()[6]) == 0, out         # the cruder prefix with its wider margin still predicts every run
            if sabotage and n == 0:
                assert int(out.split()[6]) > 0, out          # the falsified predictions were caught and re-added
            if noise == 0 and sabotage == 0 and mb == 32 and v.size > 20000 and n in (0, 1, 2):      # softmax-like vectors: nearly everything is on a grid
                fields = out.split()
                assert int(fields[4]) < 400 and int(fields[6]) == 0, out
