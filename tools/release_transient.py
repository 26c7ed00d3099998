"""Decode rate over the seconds after the first step of a repacked model (Llama-2-7B): the first step gives the row-major tensors back
(one copy of the weights) and the driver scrubs the released memory in the background -- L2_ONE_COPY=0 (behind L2_TEST_HOOKS=1) keeps
both copies for comparison.  profiles/r04/one_copy_release_transient.txt."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from llama2_ts_amd import configs, runtime
hdr = configs.header("llama2_7b")
ctx = runtime.Context(hdr); ctx.synth_fill(1)
t0 = time.perf_counter()
ctx.bench_decode(1, 0, 8)
print("first steps (pack + release) %.2f s, weights MiB %d" % (time.perf_counter() - t0, ctx.get_option(runtime.OPT_WEIGHT_MIB)))
for i in range(14):
    ms = ctx.bench_decode(1, 0, 128)
    print("t=%5.2f s  %.2f tok/s" % (time.perf_counter() - t0, 128e3 / ms), flush=True)
ctx.close()
