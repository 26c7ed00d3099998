"""ctypes binding to the CPU oracle (oracle/build/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "build", "liboracle.so")

T = dict(TOKEN_EMBEDDING=0, RMS_ATT=1, WQ=2, WK=3, WV=4, WO=5, RMS_FFN=6, W1=7, W2=8, W3=9,
         RMS_FINAL=10, FREQ_REAL=11, FREQ_IMAG=12, WCLS=13)
S = dict(x=0, xb=1, xb2=2, hb=3, hb2=4, q=5, k=6, v=7, att=8, logits=9, key_cache=10, value_cache=11)


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size",
                                      "seq_len", "shared_weights", "head_size")]


def build():
    src = [os.path.join(ROOT, "oracle", f) for f in ("llama2_oracle.c", "llama2_oracle.h", "Makefile")]
    if not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_create_synth.restype = C.c_void_p
        L.orc_create_synth.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_open.restype = C.c_void_p
        L.orc_open.argtypes = [C.c_char_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_get_config.restype = C.POINTER(OrcConfig)
        L.orc_get_config.argtypes = [C.c_void_p]
        L.orc_weights.restype = C.c_void_p
        L.orc_weights.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_state.restype = C.c_void_p
        L.orc_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.orc_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_forward_tp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_argmax.restype = C.c_int
        L.orc_argmax.argtypes = [C.c_void_p, C.c_int]
        L.orc_random_u32.restype = C.c_uint32
        L.orc_random_u32.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_random_f32.restype = C.c_float
        L.orc_random_f32.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_sample.restype = C.c_int
        L.orc_sample.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        L.orc_sample_topp.restype = C.c_int
        L.orc_sample_topp.argtypes = [C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_uint64)]
        L.orc_next_token.restype = C.c_int
        L.orc_next_token.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_uint64)]
        L.orc_tensor_count.restype = C.c_uint64
        L.orc_tensor_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_tensor_offset.restype = C.c_uint64
        L.orc_tensor_offset.argtypes = [C.c_void_p, C.c_int]
        L.orc_checkpoint_floats.restype = C.c_uint64
        L.orc_checkpoint_floats.argtypes = [C.c_void_p]
        L.orc_read_config.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_synth_tensor.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.orc_synth_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_float, C.c_float]
        L.orc_synth_write.restype = C.c_int
        L.orc_synth_write.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p]
        L.orc_time_forward.restype = C.c_double
        L.orc_time_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_softmax.argtypes = [C.c_void_p, C.c_int]
        L.orc_matmul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def _hdr(hdr):
    return np.asarray(hdr, dtype=np.int32)


def _view(ptr, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(int(n),))


class Oracle:
    """One model instance of the CPU restatement (RunState included)."""

    def __init__(self, hdr, seed=1, path=None):
        L = lib()
        self.hdr = tuple(int(v) for v in hdr)
        h = _hdr(hdr)
        self._m = L.orc_open(path.encode()) if path else L.orc_create_synth(h.ctypes.data, seed)
        if not self._m:
            raise RuntimeError("oracle model creation failed")
        self.cfg = L.orc_get_config(self._m).contents
        self.V = self.cfg.vocab_size

    def close(self):
        if self._m:
            lib().orc_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor_layers(self, kind):
        return self.cfg.n_layers if 1 <= kind <= 9 else 1

    def weights(self, kind, layer=-1):
        """numpy view of one tensor (one layer of it when layer >= 0)."""
        L = lib()
        total = L.orc_tensor_count(C.byref(self.cfg), kind)
        if kind == T["WCLS"] and self.cfg.shared_weights:
            total = L.orc_tensor_count(C.byref(self.cfg), T["TOKEN_EMBEDDING"])
        n = total // self.tensor_layers(kind) if layer >= 0 else total
        return _view(L.orc_weights(self._m, kind, max(layer, 0)), n)

    def state(self, name):
        n = C.c_size_t()
        p = lib().orc_state(self._m, S[name], C.byref(n))
        return _view(p, n.value)

    def forward(self, token, pos):
        out = np.empty(self.V, dtype=np.float32)
        lib().orc_forward(self._m, int(token), int(pos), out.ctypes.data)
        return out

    def forward_tp(self, token, pos, g):
        out = np.empty(self.V, dtype=np.float32)
        lib().orc_forward_tp(self._m, int(token), int(pos), int(g), out.ctypes.data)
        return out

    def time_forward(self, steps):
        toks = np.zeros(steps, dtype=np.int32)
        sec = lib().orc_time_forward(self._m, 0, steps, toks.ctypes.data)
        return sec, toks


class Rng:
    """The reference's BigInt xorshift* state (llama2.ts:348-355) as a uint64."""

    def __init__(self, seed):
        self.state = C.c_uint64(int(seed))

    def u32(self):
        return lib().orc_random_u32(C.byref(self.state))

    def f32(self):
        return lib().orc_random_f32(C.byref(self.state))


def next_token(logits, temperature, topp, rng):
    """llama2.ts:476-493 on a COPY of `logits` (the reference mutates state.logits); returns (token, probabilities)."""
    v = np.array(logits, dtype=np.float32, copy=True)
    tok = lib().orc_next_token(v.ctypes.data, v.size, float(temperature), float(topp), C.byref(rng.state))
    return tok, v


def argmax(v):
    v = np.ascontiguousarray(v, dtype=np.float32)
    return lib().orc_argmax(v.ctypes.data, v.size)


def synth_tensor(hdr, seed, kind, layer=-1):
    L = lib()
    cfg = OrcConfig()
    h = _hdr(hdr)
    L.orc_read_config(h.ctypes.data, C.byref(cfg))
    total = L.orc_tensor_count(C.byref(cfg), kind)
    layered = 1 <= kind <= 9
    n = total // cfg.n_layers if (layer >= 0 and layered) else total
    out = np.empty(n, dtype=np.float32)
    L.orc_synth_tensor(C.byref(cfg), seed, kind, layer, out.ctypes.data)
    return out


def set_gqa(on):
    """SURVEY.md 8(f4), parity unpinned by the reference: make the restatement honour n_kv_heads < n_heads (grouped-query
    attention as llama2.c defines it).  Off = the reference's behaviour (the field is parsed and ignored)."""
    lib().orc_set_gqa(int(bool(on)))


def rope_runc(hdr):
    """(freq_cis_real, freq_cis_imag) as llama2.c's run.c computes them per position (fp32 powf / cosf / sinf), each (S * hs/2)."""
    L = lib()
    cfg = OrcConfig()
    h = _hdr(hdr)
    L.orc_read_config(h.ctypes.data, C.byref(cfg))
    n = cfg.seq_len * (cfg.head_size // 2)
    re, im = np.empty(n, dtype=np.float32), np.empty(n, dtype=np.float32)
    L.orc_rope_runc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_rope_runc(C.byref(cfg), re.ctypes.data, im.ctypes.data)
    return re, im


def synth_write(hdr, seed, path):
    h = _hdr(hdr)
    if lib().orc_synth_write(h.ctypes.data, seed, path.encode()) != 0:
        raise RuntimeError("orc_synth_write failed")
