"""Named checkpoint shapes (llama2.c-v0 header ints, llama2.ts:80-93) used by tests and bench.

The header is the 7 int32 of the `.bin` file: dim, hidden_dim, n_layers, n_heads, n_kv_heads,
vocab_size (negative => unshared classifier, llama2.ts:90), seq_len.  Shapes follow SURVEY.md
section 8's table; the weights are synthetic (the real stories*.bin / Llama-2 files are not
available offline) and come from the repo's deterministic generator.
"""

CONFIGS = {
    # name: (dim, hidden_dim, n_layers, n_heads, n_kv_heads, vocab_size, seq_len)
    "tiny": (64, 176, 2, 4, 4, 512, 64),
    "ragged": (66, 170, 2, 3, 3, -259, 33),          # n % 4 != 0 everywhere, unshared classifier
    "tinylong": (64, 176, 2, 4, 4, 512, 1280),       # long context at tiny width: crosses every attention split level
    "stories15M": (288, 768, 6, 6, 6, 32000, 256),
    "stories110M": (768, 2048, 12, 12, 12, 32000, 1024),
    "llama2_7b_L2": (4096, 11008, 2, 32, 32, -32000, 2048),   # 7B width, 2 layers (golden-sized)
    "llama2_7b": (4096, 11008, 32, 32, 32, -32000, 2048),
}

DEFAULT_SEED = 1


def header(name):
    return tuple(int(v) for v in CONFIGS[name])


def checkpoint_bytes(hdr):
    d, h, L, H, _kv, V, S = hdr
    shared = V > 0
    V = abs(V)
    hs = d // H
    n = V * d + L * d + 4 * L * d * d + L * d + 3 * L * d * h + d + 2 * S * (hs // 2)
    if not shared:
        n += V * d
    return 28 + 4 * n


def algorithmic_bytes_per_token(hdr, pos):
    """weights + norms + embedding row + KV read/write + RoPE row + logits write."""
    d, h, L, H, _kv, V, S = hdr
    V = abs(V)
    hs = d // H
    return 4 * (L * (4 * d * d + 3 * d * h + 2 * d) + d + V * d + d + L * (2 * (pos + 1) * d + 2 * d) + hs) + 4 * V
