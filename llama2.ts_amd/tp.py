"""Tensor-parallel sharding plan for the forward pass; mirrors tensor_slice() in
csrc/llama2_hip.hip.  Rank r of G owns whole attention heads and FFN rows:

  wq, wk, wv : rows  [r*d/G, (r+1)*d/G)      (H/G heads; RoPE, KV cache and attention are head-local)
  wo         : cols  [r*d/G, (r+1)*d/G)      -> fp64 partial of xb2, all-reduce(sum), ONE fp32 rounding
  w1, w3     : rows  [r*h/G, (r+1)*h/G)      (SwiGLU is elementwise)
  w2         : cols  [r*h/G, (r+1)*h/G)      -> fp64 partial of xb,  all-reduce(sum), ONE fp32 rounding
  wcls       : rows  [r*V/G, (r+1)*V/G)      -> logits slice, all-gather
  everything else (embedding table, norm weights, RoPE tables, x) is replicated.

Two all-reduces of d doubles per layer and one all-gather of V floats per token: latency-bound messages
(32 KB at d = 4096), which is why the library talks to RCCL directly on its own stream.
"""


def shards(cfg, G):
    """True if the header shards over G ranks (7B does for G in 1,2,4,8; stories15M/110M are replica-only)."""
    d, h, _L, H, _kv, V, _S = cfg
    V = abs(V)
    return H % G == 0 and h % G == 0 and V % G == 0 and (d // G) % 2 == 0


def tensor_slice(cfg, kind, rank, G):
    """(rows, cols, row0, col0) of rank's slice of one layer of tensor `kind` (kinds as in runtime.T_*)."""
    d, h, _L, H, _kv, V, S = cfg
    V = abs(V)
    hs2 = (d // H) // 2
    dl, hl, Vl = d // G, h // G, V // G
    if kind == 0:
        return (V, d, 0, 0)
    if kind in (1, 6, 10):
        return (1, d, 0, 0)
    if kind in (2, 3, 4):
        return (dl, d, rank * dl, 0)
    if kind == 5:
        return (d, dl, 0, rank * dl)
    if kind in (7, 9):
        return (hl, d, rank * hl, 0)
    if kind == 8:
        return (d, hl, 0, rank * hl)
    if kind in (11, 12):
        return (S, hs2, 0, 0)
    if kind == 13:
        return (Vl, d, rank * Vl, 0)
    raise ValueError(kind)
