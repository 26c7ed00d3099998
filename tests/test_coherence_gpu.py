"""The no-acquire dispatch is correct because of a RULE (csrc/kernels.hip.h: on the library's own queue no launch of a token but its first
acquires, so every load of a byte an earlier launch wrote goes past L1) -- and a rule kept by discipline alone can be broken by the next
edit.  Two guards:

  * the TYPE: such bytes travel in the kernel-argument structs as Mut<T>, which has no operator* / operator[] -- a plain load of an
    activation does not compile (tests/test_abi_cpu.py holds that on the CPU);
  * the ADVERSARY, here: L2_DEBUG_POLLUTE=1 puts a kernel behind EVERY launch of the recorded step that makes every CU pull every mutable
    line of the step into its vector L1 with plain loads.  Tokens, logits (bit for bit) and sampled tokens of such a run on the queue must
    equal those of replayed hipGraphs without the adversary -- deterministic, unlike a soak.

What the adversary found (round 6, profiles/r06/coherence_adversary.txt): with L2_TEST_COHERENCE_BREAK=1 the same probe runs against a
build in which every load of the rule is a PLAIN cached one (`make coherence_break`) -- and that build is bit-identical too.  The direct
measurement says why (tools/aql/microbench_aql.cpp, profiles/r06/l1_across_dispatches.txt): 2.8e9 plain loads of lines the same CUs had
pulled two dispatches earlier, no acquire, release none / agent / system: 0 stale.  On gfx950 with ROCm 7.2's firmware a dispatch does not
see vector-L1 lines of an earlier dispatch.  The rule is what the HSA memory model requires of a queue without acquire fences, not what
this silicon needs; it stays (it costs nothing: profiles/r06/plain_vs_sc1_loads_ab.txt), enforced by the type, and the adversary stays as
the test that would catch a chip or firmware that does carry lines over.  The opt-in test below RECORDS how the broken build fares; it
cannot demand a failure the hardware does not produce."""
import json
import os
import subprocess
import sys

import pytest

import coherence_probe as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape", list(P.SHAPES))
def test_every_cu_holding_stale_lines_changes_nothing(shape):
    r = P.probe([shape])[shape]
    assert r["queue"] == 1, "the library's own queue was not in use: this run did not test the no-acquire dispatch"
    assert r["tokens_equal"] and r["logits_equal"] and r["sampled_equal"], r


@pytest.mark.skipif(os.environ.get("L2_TEST_COHERENCE_BREAK") != "1", reason="builds a second library on the box (40 s): L2_TEST_COHERENCE_BREAK=1")
def test_record_how_a_build_with_plain_loads_fares_under_the_adversary():
    subprocess.run(["make", "-C", os.path.join(ROOT, "llama2.ts_amd", "csrc"), "coherence_break"], check=True, stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "gpurun_out", "diag", "libllama2hip_break.so")
    env = dict(os.environ, L2_LIB_PATH=lib, L2_TEST_HOOKS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "coherence_probe.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
    broken = [s for s, v in out.items() if not (v["tokens_equal"] and v["logits_equal"] and v["sampled_equal"])]
    print("\n[build with PLAIN loads under the adversary] shapes that came out different: %s   %s" % (broken or "none", json.dumps(out)))
    assert all(v["queue"] == 1 for v in out.values())      # the run was on the library's own queue (no acquire between the launches of a token)
