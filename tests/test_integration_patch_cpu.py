"""INTEGRATION.md section 1, EXECUTED: the four-edit patch applied to the real /root/reference/llama2.ts (in /tmp), types
stripped with the reference's own sucrase, run under Node with a recording stub in place of the N-API addon
(tools/run_integration_patch.py).  Build-container only: the GPU box has neither /root/reference nor this need -- there
tests/test_cli_gpu.py drives the real addon.  Reference lines: llama2.ts:433 (readConfig), :435 (readWeights), :468 (the call)."""
import os
import shutil
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.exists("/root/reference/llama2.ts") or shutil.which("node") is None,
                                reason="needs the reference checkout and node (build container only)")


def fnv(b):
    h = 0x811c9dc5
    for x in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ x) * 0x01000193) & 0xffffffff
    return h


@pytest.mark.parametrize("hdr", [(64, 176, 2, 4, 4, -512, 64), (64, 176, 3, 4, 4, 512, 64)])
def test_patched_reference_drives_the_addon_interface(hdr):
    import run_integration_patch as rip
    steps = 16
    rec = rip.run(hdr, 1, steps)
    d, h, L, H, _, V, S = hdr
    shared, V = V > 0, abs(V)
    # the library is opened once, the context created from the 7 header ints verbatim (sign of vocab_size kept)
    assert rec["open"] == ["/nonexistent/libllama2hip.so"]
    assert rec["create"] == [{"header": list(hdr), "device": 0}]
    # uploads: the 14 kinds (13 for a shared classifier) in checkpoint order, per-layer tensors layer by layer, sizes of
    # readWeights (llama2.ts:112-129), and the BYTES of the file at the tensor's offset whatever Buffer the view sits in
    hs2 = (d // H) // 2
    shapes = [(0, 0, V * d), (1, L, d), (2, L, d * d), (3, L, d * d), (4, L, d * d), (5, L, d * d), (6, L, d), (7, L, h * d), (8, L, d * h),
              (9, L, h * d), (10, 0, d), (11, 0, S * hs2), (12, 0, S * hs2)] + ([] if shared else [(13, 0, V * d)])
    want = []
    blob = open(rec["checkpoint"], "rb").read()
    assert struct.unpack("<7i", blob[:28]) == hdr
    off = 28
    for kind, layers, count in shapes:
        for layer in range(max(layers, 1)):
            want.append({"kind": kind, "layer": layer if layers else -1, "floats": count, "fnv": fnv(blob[off:off + 4 * count])})
            off += 4 * count
    assert off == len(blob)
    got = [{k: u[k] for k in ("kind", "layer", "floats", "fnv")} for u in rec["upload"]]
    assert got == want
    # one forward per position, pos = 0, 1, 2, ...; the first token is BOS (llama2.ts:463) and every next one is what the
    # reference's own argmax (llama2.ts:478) made of the logits the stub wrote into state.logits
    fw = rec["forward"]
    assert [f["pos"] for f in fw] == list(range(steps)) and all(f["logits_len"] == V for f in fw)
    tok = 1
    for f in fw:
        assert f["token"] == tok
        tok = (tok * 7 + f["pos"] * 13 + 3) % V
    assert "achieved tok/s" in rec["stdout"]
