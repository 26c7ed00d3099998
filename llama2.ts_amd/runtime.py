"""Python host layer over the C ABI (include/llama2_hip.h) -- the same seam the N-API binding uses.

It mirrors the reference's names for this path so the parity tests read like the reference:
`readConfig` (llama2.ts:80-93), `readWeights` (:112-129), `newRunState` (:147-163),
`transformer(token, pos, config, state, weights)` (:205-303, call site :468) and `argmax` (:364-366).
Weights and RunState live in HBM; `state.logits` is the only host-visible field, exactly what the
sampling loop (llama2.ts:470-508) consumes.

There is NO CPU fallback: if libllama2hip.so is missing or no gfx950 device is visible every entry
point raises.  (torch is not needed here; the library owns its HIP stream.)
"""
import ctypes as C
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("L2_LIB_PATH") or os.path.join(_HERE, "lib", "libllama2hip.so")

T_TOKEN_EMBEDDING, T_RMS_ATT, T_WQ, T_WK, T_WV, T_WO, T_RMS_FFN, T_W1, T_W2, T_W3, T_RMS_FINAL, \
    T_FREQ_REAL, T_FREQ_IMAG, T_WCLS = range(14)
TENSOR_NAMES = ["token_embedding_table", "rms_att_weight", "wq", "wk", "wv", "wo", "rms_ffn_weight", "w1", "w2",
                "w3", "rms_final_weight", "freq_cis_real", "freq_cis_imag", "wcls"]
S_X, S_XB, S_XB2, S_HB, S_HB2, S_Q, S_K, S_V, S_ATT, S_LOGITS, S_KEY_CACHE, S_VALUE_CACHE = range(12)
STATE_IDS = dict(x=S_X, xb=S_XB, xb2=S_XB2, hb=S_HB, hb2=S_HB2, q=S_Q, k=S_K, v=S_V, att=S_ATT, logits=S_LOGITS,
                 key_cache=S_KEY_CACHE, value_cache=S_VALUE_CACHE)
OPT_EXACT_ATTENTION, OPT_USE_GRAPH, OPT_KEEP_STATE, OPT_PACKED_MIB, OPT_WEIGHT_MIB, OPT_SAMPLED_TOKENS, OPT_SAMPLED_SERIAL, OPT_AQL_QUEUE, OPT_PREFILL_F32_MFMA, OPT_CHECK_POS = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
F_GQA, F_GENERATE_ROPE = 1, 2     # l2_create_ex flags
TP_SOLO_ID = b"L2-SOLO-SHARD-TIMING"   # l2_create_tp id of a shard-timing context (include/llama2_hip.h: L2_TP_SOLO_ID)

# every symbol include/llama2_hip.h declares (tests check the .so exports them all)
ABI_SYMBOLS = ["l2_abi_version", "l2_device_count", "l2_last_error", "l2_create", "l2_destroy", "l2_tp_unique_id",
               "l2_create_tp", "l2_upload", "l2_synth_fill", "l2_read_tensor", "l2_forward", "l2_logits_host",
               "l2_decode_greedy", "l2_decode_sample", "l2_debug_running_sums", "l2_read_state", "l2_set_option", "l2_get_option", "l2_timer_start",
               "l2_timer_stop", "l2_bench_gemv", "l2_bench_decode", "l2_load_checkpoint", "l2_get_header", "l2_prefill", "l2_bench_dominant_in_situ", "l2_tp_mode", "l2_create_ex", "l2_bench_tokens", "l2_dispatch_reason"]


class L2Error(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libllama2hip: %s (code %d)" % (text, code))
        self.code = code


_lib = None


def lib():
    """Load libllama2hip.so (raises if it was not built: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP extension missing: %s (run __graft_entry__.build())" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    L.l2_abi_version.restype = i32
    L.l2_device_count.restype = i32
    L.l2_last_error.restype = C.c_char_p
    L.l2_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.l2_destroy.argtypes = [vp]
    L.l2_destroy.restype = None
    L.l2_tp_unique_id.argtypes = [vp]
    L.l2_create_tp.argtypes = [vp, i32, i32, i32, vp, C.POINTER(vp)]
    L.l2_upload.argtypes = [vp, i32, i32, vp, sz]
    L.l2_synth_fill.argtypes = [vp, u32]
    L.l2_read_tensor.argtypes = [vp, i32, i32, sz, vp, sz]
    L.l2_forward.argtypes = [vp, i32, i32, vp]
    L.l2_logits_host.argtypes = [vp]
    L.l2_logits_host.restype = vp
    L.l2_decode_greedy.argtypes = [vp, i32, i32, i32, vp]
    L.l2_debug_running_sums.argtypes = [i32, vp, sz, vp]
    L.l2_decode_sample.argtypes = [vp, i32, i32, i32, C.c_double, C.c_double, C.POINTER(C.c_uint64), vp]
    L.l2_read_state.argtypes = [vp, i32, i32, vp, sz]
    L.l2_set_option.argtypes = [vp, i32, i32]
    L.l2_get_option.argtypes = [vp, i32, C.POINTER(i32)]
    L.l2_timer_start.argtypes = [vp]
    L.l2_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.l2_bench_gemv.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float)]
    L.l2_bench_decode.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float)]
    L.l2_load_checkpoint.argtypes = [C.c_char_p, i32, i32, i32, vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.l2_get_header.argtypes = [vp, vp]
    L.l2_prefill.argtypes = [vp, vp, i32, i32, vp]
    L.l2_bench_dominant_in_situ.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(i32)]
    L.l2_tp_mode.argtypes = [vp]
    L.l2_create_ex.argtypes = [vp, i32, u32, C.POINTER(vp)]
    L.l2_tp_mode.restype = i32
    L.l2_bench_tokens.argtypes = [vp, vp, i32]
    L.l2_dispatch_reason.argtypes = [vp]
    L.l2_dispatch_reason.restype = C.c_char_p
    for name in ABI_SYMBOLS:   # fail at load time, not at first use, if the .so is stale
        getattr(L, name)
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise L2Error(rc, lib().l2_last_error().decode("utf8", "replace"))


class Config:
    """`Config` of the reference (llama2.ts:69-79)."""

    def __init__(self, hdr):
        hdr = tuple(int(v) for v in hdr)
        assert len(hdr) == 7
        self.header = hdr
        (self.dim, self.hidden_dim, self.n_layers, self.n_heads, self.n_kv_heads, vocab, self.seq_len) = hdr
        self.vocab_size = abs(vocab)
        self.shared_weights = vocab > 0
        self.head_size = self.dim // self.n_heads


def running_sums(values, device=0):
    """S_i = fl(S_{i-1} + values[i]) in fp64, on the device (diagnostic for the sampler's exact parallel accumulation)."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    out = np.empty(v.size, dtype=np.float64)
    _check(lib().l2_debug_running_sums(int(device), v.ctypes.data, v.size, out.ctypes.data))
    return out


def readConfig(buf):
    """readConfig (llama2.ts:80-93): 7 little-endian int32."""
    return Config(struct.unpack("<7i", bytes(buf[:28])))


def tensor_shapes(cfg, gqa=False):
    """[(kind, n_layers_or_0, per-array float count)] in checkpoint order (llama2.ts:114-127).  `gqa`: wk / wv have
    n_kv_heads * head_size rows (contexts created with F_GQA); the reference always reads (d, d)."""
    d, h, L, V, S, hs2 = cfg.dim, cfg.hidden_dim, cfg.n_layers, cfg.vocab_size, cfg.seq_len, cfg.head_size // 2
    kvd = cfg.n_kv_heads * cfg.head_size if gqa else d
    out = [(T_TOKEN_EMBEDDING, 0, V * d), (T_RMS_ATT, L, d), (T_WQ, L, d * d), (T_WK, L, kvd * d), (T_WV, L, kvd * d),
           (T_WO, L, d * d), (T_RMS_FFN, L, d), (T_W1, L, h * d), (T_W2, L, d * h), (T_W3, L, h * d),
           (T_RMS_FINAL, 0, d), (T_FREQ_REAL, 0, S * hs2), (T_FREQ_IMAG, 0, S * hs2)]
    if not cfg.shared_weights:
        out.append((T_WCLS, 0, V * d))
    return out


class Context:
    """One l2_ctx: weights + RunState of one model on one MI355X (or one rank of a TP group)."""

    def __init__(self, cfg, device=0, tp_rank=0, tp_size=1, nccl_id=None, flags=0):
        self.cfg = cfg if isinstance(cfg, Config) else Config(cfg)
        hdr = (C.c_int32 * 7)(*self.cfg.header)
        h = C.c_void_p()
        self.flags = flags
        if flags and tp_size > 1:
            raise ValueError("l2_create_ex flags (F_GQA / F_GENERATE_ROPE) cannot be combined with tp_size > 1: use l2_load_checkpoint for sharded version-1 files")
        if flags:
            _check(lib().l2_create_ex(hdr, device, int(flags), C.byref(h)))
        elif tp_size > 1:
            idbuf = C.create_string_buffer(bytes(nccl_id), 128)
            _check(lib().l2_create_tp(hdr, device, tp_rank, tp_size, idbuf, C.byref(h)))
        else:
            _check(lib().l2_create(hdr, device, C.byref(h)))
        self._h = h
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self._logits_view = None

    def close(self):
        if getattr(self, "_h", None):
            lib().l2_destroy(self._h)     # frees the pinned logits buffer: views handed out by logits_host() die here
            self._h = None
            self._logits_view = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights
    def upload(self, kind, layer, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        _check(lib().l2_upload(self._h, kind, layer, a.ctypes.data, a.size))

    def synth_fill(self, seed):
        _check(lib().l2_synth_fill(self._h, int(seed)))

    def read_tensor(self, kind, layer, offset, n):
        out = np.empty(n, dtype=np.float32)
        _check(lib().l2_read_tensor(self._h, kind, layer, offset, out.ctypes.data, n))
        return out

    # -- forward
    def forward(self, token, pos, out=None, view=False):
        """transformer(token, pos, ...) (llama2.ts:468).  Returns the logits as a fresh array; `out` (float32,
        contiguous, >= vocab_size) receives them in place; `view=True` returns the library's pinned buffer itself --
        zero copy, but the NEXT forward / prefill overwrites it and close() frees it (like state.logits in the
        reference, which every transformer() call rewrites)."""
        if out is not None:
            if out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"] or out.size < self.cfg.vocab_size:
                raise ValueError("logits array must be contiguous float32 with at least vocab_size elements")
            _check(lib().l2_forward(self._h, int(token), int(pos), out.ctypes.data))
            return out
        _check(lib().l2_forward(self._h, int(token), int(pos), None))
        return self.logits_host() if view else np.array(self.logits_host(), copy=True)

    def prefill(self, tokens, pos0=0):
        """Feed a run of (prompt) tokens at pos0.. in chunks of up to 64 tokens; returns the logits of the last position."""
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        _check(lib().l2_prefill(self._h, t.ctypes.data, t.size, int(pos0), None))
        return np.array(self.logits_host(), copy=True)

    def logits_host(self):
        """The pinned host buffer l2_forward fills (V floats), as a numpy VIEW: valid until close(), rewritten by
        every forward / prefill."""
        if self._h is None:
            raise L2Error(-4, "context is closed")
        if self._logits_view is None:
            p = lib().l2_logits_host(self._h)
            self._logits_view = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.cfg.vocab_size,))
        return self._logits_view

    def decode_greedy(self, first_token, pos0, steps):
        out = np.zeros(steps, dtype=np.int32)
        _check(lib().l2_decode_greedy(self._h, int(first_token), int(pos0), int(steps), out.ctypes.data))
        return out

    def decode_sample(self, first_token, pos0, steps, temperature, topp, rng_state):
        """The sampled branch of the loop (llama2.ts:480-493) on the device.  `rng_state`: the reference's rng_seed as an
        int; returns (tokens, advanced rng state)."""
        out = np.zeros(steps, dtype=np.int32)
        st = C.c_uint64(int(rng_state))
        _check(lib().l2_decode_sample(self._h, int(first_token), int(pos0), int(steps), float(temperature), float(topp),
                                      C.byref(st), out.ctypes.data))
        return out, int(st.value)

    def read_state(self, name, layer=-1):
        c = self.cfg
        dl, hl, Hl = c.dim // self.tp_size, c.hidden_dim // self.tp_size, c.n_heads // self.tp_size
        kvl = (c.n_kv_heads * c.head_size if getattr(self, "flags", 0) & F_GQA else c.dim) // self.tp_size
        slab = c.seq_len * kvl
        n = {"x": c.dim, "xb": dl, "xb2": c.dim, "hb": hl, "hb2": hl, "q": dl, "k": kvl, "v": kvl, "att": Hl * c.seq_len,
             "logits": c.vocab_size, "key_cache": slab * (c.n_layers if layer < 0 else 1),
             "value_cache": slab * (c.n_layers if layer < 0 else 1)}[name]
        out = np.empty(n, dtype=np.float32)
        _check(lib().l2_read_state(self._h, STATE_IDS[name], layer, out.ctypes.data, n))
        return out

    def set_option(self, key, value):
        _check(lib().l2_set_option(self._h, key, int(value)))

    def get_option(self, key):
        v = C.c_int()
        _check(lib().l2_get_option(self._h, key, C.byref(v)))
        return v.value

    def dispatch_reason(self):
        """Why the library's own queue is not in use on this context ("" when it is): l2_dispatch_reason."""
        return lib().l2_dispatch_reason(self._h).decode("utf8", "replace")

    # -- measurement
    def timer_start(self):
        _check(lib().l2_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        _check(lib().l2_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def bench_gemv(self, kind, layer, iters):
        ms = C.c_float()
        _check(lib().l2_bench_gemv(self._h, kind, layer, iters, C.byref(ms)))
        return ms.value

    def bench_dominant_in_situ(self, first_token, pos0, steps):
        us, n = C.c_float(), C.c_int()
        _check(lib().l2_bench_dominant_in_situ(self._h, first_token, pos0, steps, C.byref(us), C.byref(n)))
        return us.value, n.value

    def tp_mode_id(self):
        return int(lib().l2_tp_mode(self._h))

    def tp_mode(self):
        """How the tensor-parallel step runs (l2_tp_mode): none / RCCL eager / peer-to-peer in a graph / loopback test group."""
        return {0: "single GPU", 1: "eager launches, 2L RCCL fp64 all-reduces + 1 all-gather per token",
                2: "one hipGraph per token with the RCCL collectives captured in it",
                3: "one hipGraph per token, one-shot peer-to-peer fp64 all-reduce inside the residual kernels",
                4: "loopback test group", 5: "shard timing only (one rank alone, exchange against its own inbox)"}.get(lib().l2_tp_mode(self._h), "?")

    def bench_decode(self, first_token, pos0, steps):
        ms = C.c_float()
        _check(lib().l2_bench_decode(self._h, first_token, pos0, steps, C.byref(ms)))
        return ms.value

    def bench_tokens(self, n):
        """The first n tokens the last device-resident run (bench_decode / decode_greedy / decode_sample) chose."""
        out = np.zeros(n, dtype=np.int32)
        _check(lib().l2_bench_tokens(self._h, out.ctypes.data, int(n)))
        return out


class TransformerWeights:
    """Handle to the device-resident TransformerWeights (llama2.ts:95-110)."""

    def __init__(self, ctx):
        self.ctx = ctx


class RunState:
    """RunState (llama2.ts:131-146): only `logits` is host-visible; the rest is read on demand."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.logits = ctx.logits_host()

    def __getattr__(self, name):
        if name in STATE_IDS:
            return self.ctx.read_state(name)
        raise AttributeError(name)


def readWeights(config, f, device=0, ctx=None):
    """readWeights (llama2.ts:112-129): stream each Float32Array of the checkpoint straight to HBM.

    `f` is a binary file object positioned after the 28-byte header.  Each tensor (each layer of a
    per-layer tensor) is read into host memory, uploaded and dropped, so host RAM never holds more
    than one tensor (hard part 5)."""
    ctx = ctx or Context(config, device)
    for kind, layers, count in tensor_shapes(config):
        for layer in range(max(layers, 1)):
            buf = np.frombuffer(f.read(count * 4), dtype="<f4")
            if buf.size != count:
                raise IOError("checkpoint truncated in %s" % TENSOR_NAMES[kind])
            ctx.upload(kind, layer if layers else -1, buf)
    return TransformerWeights(ctx)


def newRunState(config, weights):
    """newRunState (llama2.ts:147-163): the buffers already exist on the device; wrap them."""
    return RunState(weights.ctx)


def transformer(token, pos, config, state, weights):
    """transformer(token, pos, p, s, w) (llama2.ts:205-303): fills state.logits."""
    weights.ctx.forward(token, pos)


def argmax(arr):
    """argmax (llama2.ts:364-366): first maximum (strict '>'), NaNs never win."""
    best = 0
    bv = arr[0]
    a = np.asarray(arr)
    # numpy's argmax returns the first maximum too; NaN handling differs, so guard
    if not np.isnan(a).any():
        return int(np.argmax(a))
    for i in range(1, a.size):
        if a[i] > bv:
            best, bv = i, a[i]
    return best


def load_checkpoint_native(path, device=0):
    """Same as load_checkpoint but through l2_load_checkpoint (pinned double-buffered streaming)."""
    h = C.c_void_p()
    n = C.c_uint64()
    _check(lib().l2_load_checkpoint(path.encode(), device, 0, 1, None, C.byref(h), C.byref(n)))
    hdr = (C.c_int32 * 7)()
    _check(lib().l2_get_header(h, hdr))
    ctx = Context.__new__(Context)
    ctx.cfg = Config(tuple(hdr))
    ctx._h, ctx.tp_rank, ctx.tp_size, ctx._logits_view = h, 0, 1, None
    with open(path, "rb") as f:                     # a version-1 export is loaded with F_GQA | F_GENERATE_ROPE (include/llama2_hip.h)
        ctx.flags = (F_GQA | F_GENERATE_ROPE) if f.read(4) == b"24ka" else 0
    weights = TransformerWeights(ctx)
    return ctx.cfg, newRunState(ctx.cfg, weights), weights, n.value


def load_checkpoint(path, device=0):
    """main()'s load sequence (llama2.ts:427-436, 451) -> (config, state, weights)."""
    with open(path, "rb") as f:
        config = readConfig(f.read(28))
        weights = readWeights(config, f, device)
    return config, newRunState(config, weights), weights
