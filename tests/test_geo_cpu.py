"""The launch geometry is a pure function (launch.hip.h: pick_geo_pure / use_small_pure, reachable through l2_debug_pick_geo without a
GPU): every phase of every named config, of the tensor-parallel shards of Llama-2-7B, of the random / wide headers the GPU tests run
and of a grid of odd shapes must select a template point the library INSTANTIATES -- and the library must instantiate no point of the
streaming kernel that nothing selects, by default or with the latency form switched off (round 5 removed 51 such instances: U = 1, PRE = 2, U = 4 with PRE = 12, the attention kernels'
second wave count, the prompt GEMM's LDS variant)."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from llama2_ts_amd import configs, runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODE_QKV, MODE_WO, MODE_W13, MODE_W2, MODE_CLS = range(5)
N_CUS, SMALL_MAX = 256, 8 << 20


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    import __graft_entry__ as graft
    graft.build()
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("no ROCm llvm tools here")
    tmp = tmp_path_factory.mktemp("co")
    so = tmp / "lib.so"
    shutil.copy(runtime.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp))
    names = set()
    for o in os.listdir(tmp):
        if "gfx950" in o:
            notes = subprocess.check_output([readelf, "--notes", str(tmp / o)]).decode()
            names |= set(re.findall(r"\.name:\s+(_Z\S+)", notes))
    stream = {tuple(int(v) for v in m) for n in names for m in re.findall(r"12phase_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])E", n)}
    small = {tuple(int(v) for v in m) for n in names for m in re.findall(r"18phase_small_kernelILi(\d+)ELi(\d+)ELi(\d+)E", n)}
    scalar = {int(m) for n in names for m in re.findall(r"19phase_kernel_scalarILi(\d+)E", n)}
    assert len(stream) > 30 and len(small) > 20 and scalar == {0, 1, 2, 3, 4}, (len(stream), len(small), scalar)
    return {"stream": stream, "small": small, "n": len(names)}


def phases(hdr, G=1, gqa=False):
    """(mode, rows, columns) of the five phases as the library launches them: q / k / v has d_loc + 2 kv_dim_loc rows (llama2_hip.hip:
    qkv_args) -- 3 d_loc unless the context honours n_kv_heads < n_heads (L2_F_GQA)."""
    d, h, _L, H, KVH, V, _S = hdr
    V = abs(V)
    dl, hl, Vl = d // G, h // G, V // G
    kvl = (KVH * (d // H) if gqa else d) // G
    return [(MODE_QKV, dl + 2 * kvl, d), (MODE_WO, d, dl), (MODE_W13, hl, d), (MODE_W2, d, hl), (MODE_CLS, Vl, d)]


def pick(mode, rows, n, small_max=SMALL_MAX):
    out = (C.c_int * 6)()
    L = runtime.lib()
    L.l2_debug_pick_geo.argtypes = [C.c_int] * 5 + [C.c_void_p]
    assert L.l2_debug_pick_geo(mode, rows, n, N_CUS, small_max, out) == 0
    return list(out)


def shapes():
    import test_hip_parity as T
    out = [(configs.header(n), 1) for n in configs.CONFIGS]
    out += [(configs.header("llama2_7b"), G) for G in (2, 4, 8)] + [(configs.header("tiny"), 2), (configs.header("tiny"), 4), (configs.header("stories15M"), 2)]
    out += [(hdr, 1) for hdr in T._random_headers(14, 20261003) + T.WIDE_SHAPES + T.FUSED_SHAPES]
    out += [((1280, 2560, 2, 10, 10, -1000, 48), 1), ((1280, 2572, 2, 10, 10, 1000, 48), 1), ((2048, 5632, 1, 16, 16, -777, 32), 1), ((1280, 1280, 1, 10, 10, -140001, 16), 1)]
    # narrow hidden sizes and wide-but-short matrices: the streaming kernel's short-row points (U = 4 with two waves, U = 2 / 4 with one staging round in w2)
    out += [((768, 300, 1, 12, 12, 300, 32), 1), ((512, 300, 1, 8, 8, -300, 32), 1), ((1024, 768, 1, 8, 8, 300, 32), 1), ((1024, 8192, 1, 8, 8, -300, 32), 1), ((640, 200, 1, 10, 10, 120, 32), 1)]
    # grouped-query shapes (the third entry: honoured, L2_F_GQA): q / k / v phases with fewer rows than 3 d -- the round-5 advisor's
    # d = 640 with 128-wide k / v, multi-query 7B width, the GQA fixtures of tests/gqa_cases.py
    out += [((640, 1728, 1, 10, 2, -300, 32), 1, True), ((4096, 11008, 1, 32, 1, -320, 16), 1, True), ((4096, 11008, 1, 32, 8, -320, 16), 8, True),
            ((1024, 2816, 1, 16, 2, 300, 32), 1, True), ((768, 2048, 1, 12, 4, 300, 32), 1, True), ((64, 176, 2, 4, 2, 512, 64), 1, True), ((256, 704, 2, 4, 1, -512, 320), 1, True)]
    rng = np.random.default_rng(5)
    for _ in range(300):      # odd widths, tiny and huge row counts
        H = int(rng.choice([1, 2, 4, 8, 16, 32]))
        hs = int(rng.choice([2, 4, 8, 16, 32, 64, 128, 256]))
        d = H * hs
        if d > 8192:
            continue
        out.append(((d, int(rng.integers(d // 2 + 1, 4 * d + 2)), 1, H, H, int(rng.integers(17, 140000)) * int(rng.choice([-1, 1])), 64), 1))
    return out


def test_every_selected_template_point_is_instantiated_and_no_streaming_point_is_dead(kernels):
    used = set()
    # (small_max 0: the development configuration L2_SMALL_MAX=0 the GPU tests use to run the streaming form on shapes that default to
    # the latency form -- a narrow qkv / wo phase reaches the streaming kernel's short-row points only that way)
    for hdr, G, gqa, small_max in [(sh[0], sh[1], len(sh) > 2 and sh[2], sm) for sh in shapes() for sm in (SMALL_MAX, 0)]:
        for mode, rows, n in phases(hdr, G, gqa):
            if rows <= 0 or n <= 0:
                continue
            form, a, b, nw, grid, packable = pick(mode, rows, n, small_max)
            assert grid >= 1 and nw in (1, 2, 4, 8)
            if form == 2:
                continue                                   # the scalar kernel: one instance per mode (checked in the fixture)
            if form == 1:
                assert (mode, a, b) in kernels["small"], ("latency form", hdr, G, mode, a, b)
                continue
            pushes = [0, 1] if (G > 1 and mode in (MODE_WO, MODE_W2)) else [0]      # tensor-parallel wo / w2: the push instance and the partial one (RCCL)
            for push in pushes:
                pts = [(mode, 2, a, b, 0, push)] + ([(mode, 2, 2, b, 1, push)] if packable else [])      # row-major (first step, no memory) and repacked
                for p in pts:
                    assert p in kernels["stream"], ("streaming form", hdr, G, p)
                    used.add(p)
    # wo / w2 push instances exist for every point a one-GPU shape selects (any width can be sharded): compare without the push bit
    used_nopush = {p[:5] for p in used}
    dead = sorted(p for p in kernels["stream"] if p[:5] not in used_nopush)
    assert not dead, "instantiated, never selected: %s" % dead
    print("\n%d kernels in the library; streaming kernel: %d instances, %d selected by %d shapes" % (kernels["n"], len(kernels["stream"]), len(used), len(shapes())))


def test_geometry_of_the_benchmark_shapes_is_what_design_md_says():
    """Llama-2-7B, one GPU: every layer phase streams a repacked copy (U = 2), w1 / w3 on 459 x 4 waves x 3 row groups; stories110M: the
    latency form for every layer phase, the classifier on three float4 per lane."""
    for mode, rows, n in phases(configs.header("llama2_7b"))[:4]:
        form, U, pre, nw, grid, packable = pick(mode, rows, n)
        assert (form, U, nw, packable) == (0, 2, 4, 1)
    assert pick(MODE_W13, 11008, 4096)[4] == 459
    for mode, rows, n in phases(configs.header("stories110M"))[:4]:
        assert pick(mode, rows, n)[0] == 1
    assert pick(MODE_CLS, 32000, 768)[:3] == [0, 3, 1]
