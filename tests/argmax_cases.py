"""Edge cases of the reference's `argmax` (llama2.ts:364-366: `arr.reduce((maxIdx, val, idx, array) => (val > array[maxIdx] ? idx : maxIdx), 0)`)
as MODELS: a seeded synthetic checkpoint whose classifier rows (unshared: vocab_size < 0) are patched so that the logits the forward
pass produces hit them -- exact ties placed in different workgroups and on different argmax keys of the device's classifier launch,
-0 beside +0, +-inf, NaN, nothing but NaN, and NaN at index 0 (what the reduce() does there: `val > NaN` is never true, so the pick
stays 0 WHATEVER the other logits are).  Shared by tests/test_argmax_edges_gpu.py (the device's picks against the oracle's on the
same patched tensors), tests/test_oracle_golden.py (the oracle against what the REAL reference picked) and oracle/make_goldens.py
(which writes each model as a v0 file and runs the reference on it)."""
import struct

import numpy as np

import oracle_lib as O

# d = 256: the classifier takes the streaming kernel with ONE column batch per row; 2048 rows = 1024 row groups = 256 workgroups of
# four waves, folded into the eight keys by (workgroup & 7).  `odd`: n % 4 != 0 everywhere = the scalar kernels + the one-wave pick.
SHAPES = {"vec": (256, 512, 2, 4, 4, -2048, 48), "odd": (66, 170, 2, 3, 3, -259, 33)}
SEED = 11
DENORM = np.float32(1.401298464324817e-45)      # the smallest fp32 denormal: its product with |x| < 0.5 rounds to +-0

CASES = ("ties", "specials", "nan0", "allnan", "zeros")


def _spread(V):
    """Row indices far apart: different row groups, different workgroups, different (workgroup & 7) key lines."""
    return [9, 10, V // 3 + 17, (2 * V) // 3 - 64, V - 1]


def patch(case, shape, tensors):
    """Modify {kind: flat float32 array} in place for `case`; returns a short description."""
    d, h, L, H, KVH, V, S = SHAPES[shape]
    V = abs(V)
    wcls = tensors[13].reshape(V, d)
    if case == "ties":
        # five copies of one strong row (64 x an ordinary one) and two copies of its negative: whatever the sign of its dot product
        # with x, the maximum is an EXACT tie between rows in different workgroups -- the reference keeps the smallest index
        base = (64.0 * wcls[V // 2 + 5]).astype(np.float32)
        for r in _spread(V):
            wcls[r] = base
        for r in (5, (3 * V) // 5 + 2):
            wcls[r] = -base
        return "exact ties of the maximum across workgroups and key lines"
    if case == "specials":
        wcls[3, 0] = np.nan                       # a NaN logit in the middle: never wins
        for r in (4, V // 3 + 33, V - 5):         # +inf * x[1]: +inf (a tie of infinities), -inf, or NaN when x[1] == 0
            wcls[r, 1] = np.inf
        for r in (6, V - 9):                      # ... and the same tie for the other sign of x[1]
            wcls[r, 1] = -np.inf
        wcls[7, 1], wcls[7, 2] = np.inf, -np.inf  # inf - inf
        return "NaN, +inf (tied), -inf and inf - inf logits"
    if case == "nan0":
        wcls[0, 0] = np.nan                       # logits[0] = NaN: `val > NaN` is false for every val -> the pick stays 0
        return "NaN at index 0: the reference's reduce() never leaves it"
    if case == "allnan":
        tensors[10][d // 2] = np.nan              # rms_final_weight -> the final-normed x holds a NaN -> every logit is NaN
        return "nothing but NaN -> index 0"
    if case == "zeros":
        wcls[:] = 0.0
        tensors[10][7] = tensors[10][9] = 0.01    # |x[7]|, |x[9]| < 0.5 after the final norm: DENORM * x rounds to a zero that keeps the sign
        wcls[0, 7], wcls[1, 7] = -DENORM, DENORM  # the two roundings to zero have opposite signs: one of logits[0], logits[1] is -0
        wcls[V - 2, 9] = -DENORM
        return "-0 beside +0: `>` does not tell them apart -> index 0"
    raise KeyError(case)


def tensors_of(case, shape):
    """{kind: flat float32 array} of the patched model."""
    hdr = SHAPES[shape]
    o = O.Oracle(hdr, SEED)
    t = {k: np.array(o.weights(k), copy=True) for k in range(14)}
    o.close()
    patch(case, shape, t)
    return t


def patched_oracle(case, shape):
    """An oracle model holding the patched tensors (its weight views are writable)."""
    o = O.Oracle(SHAPES[shape], SEED)
    t = tensors_of(case, shape)
    for k in range(14):
        o.weights(k)[:] = t[k]
    return o, t


def write_v0(path, case, shape):
    t = tensors_of(case, shape)
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", *SHAPES[shape]))
        for k in range(14):
            f.write(np.ascontiguousarray(t[k], dtype="<f4").tobytes())
    return t


def oracle_run(case, shape, steps):
    """(tokens fed, picks, logits per step) of the oracle's greedy loop on the patched model (llama2.ts:463-479, 496)."""
    o, _ = patched_oracle(case, shape)
    fed, picks, logits = [], [], []
    tok = 1
    for pos in range(steps):
        lg = o.forward(tok, pos)
        nxt = O.argmax(lg)
        fed.append(tok); picks.append(nxt); logits.append(lg)
        tok = nxt
    o.close()
    return fed, picks, logits
