"""Deterministic synthetic `tokenizer.bin` in the reference's format (read at llama2.ts:444-449):
int32 max_token_length, then per token: float32 score, int32 byte length, UTF-8 bytes.

The real tokenizer.bin ships with the reference and is not copied into this repo; the CLI tests and the
CLI goldens (oracle/make_goldens.py runs the TRUE reference with this file in its working directory) use
this stand-in instead.  Layout mirrors the real one: ids 0-2 specials, ids 3-258 the code points
U+0000-U+00FF, then BPE-style merges (every later token is the concatenation of two earlier ones, so
bpe_encode's greedy pair merging (llama2.ts:305-344) has real work to do)."""
import struct

BASE = " etaoinshrdlucmfwypvbgkqjxz"


def _hash32(a):
    a &= 0xFFFFFFFF
    a ^= a >> 16; a = (a * 0x7FEB352D) & 0xFFFFFFFF
    a ^= a >> 15; a = (a * 0x846CA68B) & 0xFFFFFFFF
    a ^= a >> 16
    return a


def build_vocab(vocab_size=32000):
    vocab = ["<unk>", "\n<s>\n", "\n</s>\n"] + [chr(i) for i in range(256)]
    scores = [0.0] * len(vocab)
    seen = set(vocab)
    pool = [3 + ord(ch) for ch in BASE]          # ids usable as merge operands
    ctr = 0
    while len(vocab) < vocab_size:
        ctr += 1
        h1, h2 = _hash32(ctr * 2 + 1), _hash32(ctr * 2 + 2)
        a, b = pool[h1 % len(pool)], pool[h2 % len(pool)]
        s = vocab[a] + vocab[b]
        if s in seen or len(s) > 10 or s.count(" ") > 1 or (" " in s[1:]):
            continue
        seen.add(s)
        vocab.append(s)
        scores.append(-float(len(vocab) - 259))
        pool.append(len(vocab) - 1)
    return vocab[:vocab_size], scores[:vocab_size]


def write(path, vocab_size=32000):
    vocab, scores = build_vocab(vocab_size)
    enc = [v.encode("utf8") for v in vocab]
    with open(path, "wb") as f:
        f.write(struct.pack("<i", max(len(e) for e in enc)))
        for e, sc in zip(enc, scores):
            f.write(struct.pack("<fi", sc, len(e)))
            f.write(e)
    return vocab, scores


if __name__ == "__main__":
    import sys
    v, _ = write(sys.argv[1])
    print(len(v), v[259:280], v[-5:])
