"""CPU sanitizer runs ("ASan on host shim"; there is no GPU sanitizer on this pool).

* the N-API addon (llama2.ts_amd/host/l2_napi.cc: the pointer / length handling that stands where FileHandleReader.getF32Array views
  and the state.logits hand-off of llama2.ts:44-68, 468 were) built with -fsanitize=address,undefined and driven by Node against
  a host-memory stand-in for the library (tests/stub/l2_stub.c) that touches exactly the bytes the real one would: offset views,
  short logits arrays, wrong kinds and types, use after destroy -- plus a NEGATIVE control (the stub told to write one float too
  many) that has to end in an AddressSanitizer report, so a silent pass means something;
* the oracle's forward / generator / samplers (`make -C oracle asan`);
* the host harness of the sampler's exact running sums (tests/exact_sum_host.cc over csrc/exact_sum.h)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "llama2.ts_amd", "host")


def _runtime_libs():
    libs = []
    for name in ("libasan.so", "libubsan.so"):
        p = subprocess.check_output(["gcc", "-print-file-name=" + name]).decode().strip()
        if not os.path.isabs(p) or not os.path.exists(p):
            pytest.skip("gcc has no %s here" % name)
        libs.append(p)
    return ":".join(libs)


@pytest.fixture(scope="module")
def asan_addon():
    if shutil.which("node") is None or not os.path.exists("/usr/include/node/node_api.h"):
        pytest.skip("no node / node headers")
    subprocess.run(["make", "-C", HOST, "asan"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(HOST, "build", "l2_napi_asan.node"), os.path.join(HOST, "build", "libllama2hip_stub.so")


def _drive(asan_addon, tmp_path, **extra):
    ck = tmp_path / "tiny.bin"
    ck.write_bytes(struct.pack("<7i", 64, 176, 2, 4, 4, 512, 64) + b"\0" * 64)
    env = dict(os.environ, LD_PRELOAD=_runtime_libs(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", **extra)
    return subprocess.run(["node", os.path.join(ROOT, "tests", "stub", "napi_asan_driver.js"), asan_addon[0], asan_addon[1], str(ck)],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)


def test_napi_addon_is_clean_under_asan_and_ubsan(asan_addon, tmp_path):
    r = _drive(asan_addon, tmp_path)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-3000:]
    assert "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
    out = r.stdout.decode().split()
    assert out[0] == "ok" and int(out[1]) >= 70          # every check of the driver ran (offset views of all 14 kinds, short arrays, wrong types ...)


def test_the_sanitizer_is_live_on_typed_array_memory(asan_addon, tmp_path):
    """Negative control: the stub writes V + 1 floats into a V-float Float32Array handed over by the addon."""
    r = _drive(asan_addon, tmp_path, L2_STUB_OVERRUN="1", L2_STUB_OVERRUN_CALL="1")
    err = r.stderr.decode()
    assert r.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in err and "l2_forward" in err, err[-2000:]


def test_oracle_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    out = r.stdout.decode()
    assert r.returncode == 0 and "oracle: ASan/UBSan clean" in out and "runtime error" not in out, out[-3000:]


def test_exact_sum_harness_under_asan_and_ubsan(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sum_cases
    exe = str(tmp_path / "exact_sum_host_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-fno-sanitize-recover=undefined", "-o", exe, os.path.join(ROOT, "tests", "exact_sum_host.cc")])
    src, dst = str(tmp_path / "in.f32"), str(tmp_path / "out.f64")
    for v in sum_cases.adversarial()[:4]:
        want = sum_cases.serial_sums(v)
        v.tofile(src)
        for tile, noise, sabotage, mb in ((1024, 0, 0, 32), (64, 12345, 0, 32), (256, 9, 2, 32), (1024, 4242, 0, 20)):
            r = subprocess.run([exe, src, dst, str(tile), str(noise), str(sabotage), str(mb)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, UBSAN_OPTIONS="halt_on_error=1"))
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            assert np.array_equal(np.fromfile(dst, dtype=np.float64).view(np.uint64), want.view(np.uint64))
