"""Decode a few hundred tokens with ANY build of the library (only the round-1 entry points), for same-box A/B runs of two builds under
rocprofv3:  L2_USE_GRAPH=0 rocprofv3 --kernel-trace --stats ... -- python3 tools/ab_lib_decode.py <lib.so> <config> [tokens] [pos0]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llama2_ts_amd import configs
L = C.CDLL(sys.argv[1])
hdr = configs.header(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 64; pos0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
h = C.c_void_p()
assert L.l2_create((C.c_int32 * 7)(*hdr), 0, C.byref(h)) == 0
assert L.l2_synth_fill(h, C.c_uint32(1)) == 0
out = (C.c_int32 * (pos0 + n))()
L.l2_decode_greedy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
assert L.l2_decode_greedy(h, 1, 0, pos0 + n, out) == 0
for _ in range(3):
    t0 = time.perf_counter()
    assert L.l2_decode_greedy(h, 1, pos0, n, out) == 0
    dt = time.perf_counter() - t0
print("%s %s: %.1f tok/s (%d tokens from pos %d)" % (os.path.basename(sys.argv[1]), sys.argv[2], n / dt, n, pos0))
L.l2_destroy.argtypes = [C.c_void_p]; L.l2_destroy(h)
