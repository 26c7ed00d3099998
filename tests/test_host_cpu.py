"""CPU-side checks of the JS host + N-API addon: builds, loads under the box's node, rejects bad arguments, and
fails loudly (exit 1, no fallback) when no GPU is present."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "llama2.ts_amd", "host", "l2_run.mjs")

pytestmark = pytest.mark.skipif(shutil.which("node") is None, reason="no node")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as graft
    graft.build()
    assert os.path.exists(os.path.join(ROOT, "llama2.ts_amd", "host", "l2_napi.node"))


def test_driver_argument_errors(built, tmp_path):
    r = subprocess.run(["node", HOST], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stdout == b"" and b"usage: node l2_run.mjs <checkpoint>" in r.stderr
    for argv in (["m.bin", "--steps"], ["m.bin", "--bogus", "1"], ["m.bin", "steps", "1"]):
        r = subprocess.run(["node", HOST, *argv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path))
        assert r.returncode == 1 and r.stdout == b"", argv


def test_driver_fails_loudly_without_gpu(built, tmp_path):
    """No CPU path behind the JS boundary either: a checkpoint that opens but no gfx950 device => exit 1."""
    import struct
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    p = tmp_path / "m.bin"
    p.write_bytes(struct.pack("<7i", 64, 176, 2, 4, 4, 512, 64) + b"\0" * 64)
    r = subprocess.run(["node", HOST, str(p), "--steps", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and r.stdout == b"" and b"no HIP device visible" in r.stderr


def test_addon_exports_and_fails_without_gpu(built, tmp_path):
    js = ("const a=require(%r); console.log(Object.keys(a).sort().join(','));"
          "a.open(%r); try{a.create(new Int32Array([64,176,2,4,4,512,64]),0); console.log('created')}"
          "catch(e){console.log('ERR '+e.message)}") % (
        os.path.join(ROOT, "llama2.ts_amd", "host", "l2_napi.node"),
        os.path.join(ROOT, "llama2.ts_amd", "lib", "libllama2hip.so"))
    r = subprocess.run(["node", "-e", js], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = r.stdout.decode().splitlines()
    assert out[0] == "create,decodeGreedy,decodeSample,destroy,deviceCount,forward,getOption,loadCheckpoint,logitsBuffer,open,prefill,readState,readTensor,setOption,synthFill,upload"
    import torch
    if not torch.cuda.is_available():
        assert out[1].startswith("ERR libllama2hip: no HIP device visible")


def test_synthetic_tokenizer_is_deterministic(tmp_path):
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_tokenizer
    v, s = synth_tokenizer.write(str(tmp_path / "t.bin"))
    assert len(v) == 32000 and v[1] == "\n<s>\n" and v[3 + ord("a")] == "a" and len(set(v)) == 32000
    h = hashlib.sha256(open(tmp_path / "t.bin", "rb").read()).hexdigest()
    v2, _ = synth_tokenizer.write(str(tmp_path / "t2.bin"))
    assert v2 == v and hashlib.sha256(open(tmp_path / "t2.bin", "rb").read()).hexdigest() == h


def test_typescript_declarations_cover_the_addon(built):
    """l2_napi.d.ts (for hosts that stay TypeScript) declares exactly the functions the addon exports."""
    import re
    dts = open(os.path.join(ROOT, "llama2.ts_amd", "host", "l2_napi.d.ts")).read()
    declared = sorted(set(re.findall(r"export function (\w+)\(", dts)))
    js = "console.log(Object.keys(require(%r)).sort().join(','))" % os.path.join(ROOT, "llama2.ts_amd", "host", "l2_napi.node")
    r = subprocess.run(["node", "-e", js], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert declared == r.stdout.decode().strip().split(",")


def test_typescript_twin_erases_to_the_module(built, tmp_path):
    """`host stays TypeScript` taken literally: l2_backend.ts is the same module in erasable-syntax TypeScript.  Stripping its types
    with the stripper the reference itself ships (sucrase inside /root/reference/t348.mjs -- what the reference's own loader does to
    llama2.ts at import) must give l2_backend.mjs back token for token (comments and whitespace aside), and the stripped file must
    load under this box's Node with the same exports."""
    import re
    import sys
    if not os.path.exists("/root/reference/t348.mjs"):
        pytest.skip("the reference (and its bundled type stripper) is not on this box")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_goldens as mg
    host = os.path.join(ROOT, "llama2.ts_amd", "host")
    out = tmp_path / "l2_backend.stripped.mjs"
    mg.strip_types(os.path.join(host, "l2_backend.ts"), str(out))

    def code(src):
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r'(^|[^:"])//[^\n]*', r"\1", src)
        return re.sub(r"\s+", "", src)
    assert code(out.read_text()) == code(open(os.path.join(host, "l2_backend.mjs")).read())
    ts = open(os.path.join(host, "l2_backend.ts")).read()
    for banned in ("enum ", "namespace ", "constructor(private", "constructor(public", "import type"):     # erasable syntax only
        assert banned not in ts
    js = ("import * as a from %r; import * as b from %r; console.log(Object.keys(a).sort().join(',') == Object.keys(b).sort().join(','), Object.keys(a).length)"
          % (str(out), os.path.join(host, "l2_backend.mjs")))
    probe = tmp_path / "probe.mjs"
    probe.write_text(js)
    r = subprocess.run(["node", str(probe)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.decode().split()[0] == "true", r.stderr.decode()[-800:]
