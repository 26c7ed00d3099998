// loader.hip.h -- native checkpoint reader, the synthetic-weight generator and tensor read-back
// Part of the one translation unit llama2_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// ---- native checkpoint loader -------------------------------------------------
extern "C" int l2_get_header(l2_ctx* c, int32_t cfg_out[7]) {
  if (!c || !cfg_out) return fail(L2_E_ARG, "null argument");
  memcpy(cfg_out, c->hdr, sizeof(c->hdr));
  return L2_OK;
}

extern "C" int l2_load_checkpoint(const char* path, int device, int tp_rank, int tp_size, const void* nccl_id,
                                  l2_ctx** out, uint64_t* bytes_read) {
  if (!path || !out) return fail(L2_E_ARG, "null argument");
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return fail(L2_E_ARG, "cannot open checkpoint %s", path);
  int32_t hdr[7];
  if (fread(hdr, 4, 7, f) != 7) { fclose(f); return fail(L2_E_ARG, "checkpoint %s: short header", path); }
  // llama2.c "version 1" export: magic "ak42", version, the 7 ints, one byte shared_classifier, padded to 256 bytes;
  // tensors in a different order (norms first) and no freq_cis.  Anything else is the v0 layout the reference reads.
  static const int order_v0[] = {L2_T_TOKEN_EMBEDDING, L2_T_RMS_ATT, L2_T_WQ, L2_T_WK, L2_T_WV, L2_T_WO, L2_T_RMS_FFN, L2_T_W1, L2_T_W2, L2_T_W3,
                                 L2_T_RMS_FINAL, L2_T_FREQ_REAL, L2_T_FREQ_IMAG, L2_T_WCLS};
  static const int order_v1[] = {L2_T_RMS_ATT, L2_T_RMS_FFN, L2_T_RMS_FINAL, L2_T_TOKEN_EMBEDDING, L2_T_WQ, L2_T_WK, L2_T_WV, L2_T_WO, L2_T_W1,
                                 L2_T_W2, L2_T_W3, L2_T_WCLS};
  const int* order = order_v0;
  int n_order = 14;
  unsigned flags = 0;
  uint64_t total = 28;
  if ((uint32_t)hdr[0] == 0x616b3432u) {
    if (hdr[1] != 1) { fclose(f); return fail(L2_E_CONFIG, "checkpoint %s: version %d export (only the fp32 version 1 is supported)", path, hdr[1]); }
    int32_t h1[7];
    unsigned char shared = 0;
    if (fseek(f, 8, SEEK_SET) || fread(h1, 4, 7, f) != 7 || fread(&shared, 1, 1, f) != 1 || fseek(f, 256, SEEK_SET)) { fclose(f); return fail(L2_E_ARG, "checkpoint %s: short header", path); }
    memcpy(hdr, h1, sizeof(hdr));
    hdr[5] = shared ? abs(hdr[5]) : -abs(hdr[5]);        // the v0 convention: sign of vocab_size = shared classifier (llama2.ts:90)
    order = order_v1; n_order = 12; flags = L2_F_GQA | L2_F_GENERATE_ROPE; total = 256;
  }
  l2_ctx* c = nullptr;
  int rc = (tp_size > 1) ? create_impl(hdr, device, tp_rank, tp_size, nccl_id, &c, flags) : create_impl(hdr, device, 0, 1, nullptr, &c, flags);
  if (rc) { fclose(f); return rc; }
  if (flags & L2_F_GENERATE_ROPE) { rc = generate_rope(c); if (rc) { fclose(f); l2_destroy(c); return rc; } }
  // two pinned staging buffers: fread into one while the other is in flight to the device
  const size_t CH = (size_t)64 << 20;
  float* stage[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  bool pending[2] = {false, false};
  auto cleanup = [&](int code) {
    for (int i = 0; i < 2; ++i) { if (stage[i]) hipHostFree(stage[i]); if (done[i]) hipEventDestroy(done[i]); }
    fclose(f);
    if (code) l2_destroy(c);
    return code;
  };
  for (int i = 0; i < 2; ++i) {
    if (hipHostMalloc(&stage[i], CH, hipHostMallocDefault) != hipSuccess || hipEventCreate(&done[i]) != hipSuccess)
      return cleanup(fail(L2_E_HIP, "cannot allocate pinned staging"));
  }
  int cur = 0;
  for (int oi = 0; oi < n_order; ++oi) {
    const int kind = order[oi];
    if (kind == L2_T_WCLS && c->shared) continue;
    const Slice sl = tensor_slice(c, kind);
    const size_t full_layer = sl.full_rows * sl.full_cols;
    for (int layer = 0; layer < c->layers_of[kind]; ++layer) {
      float* dst = c->w[kind] + c->layer_elems[kind] * (size_t)layer;
      // stream the layer in whole-row chunks; a rank keeps only its rows / columns
      const size_t rows_per_chunk = CH / (sl.full_cols * sizeof(float)) ? CH / (sl.full_cols * sizeof(float)) : 1;
      if (sl.full_cols * sizeof(float) > CH) return cleanup(fail(L2_E_CONFIG, "row of %zu floats exceeds the staging buffer", sl.full_cols));
      for (size_t r0 = 0; r0 < sl.full_rows; r0 += rows_per_chunk) {
        const size_t nr = (sl.full_rows - r0 < rows_per_chunk) ? sl.full_rows - r0 : rows_per_chunk;
        if (pending[cur]) { if (hipEventSynchronize(done[cur]) != hipSuccess) return cleanup(fail(L2_E_HIP, "staging sync failed")); pending[cur] = false; }
        if (fread(stage[cur], sizeof(float), nr * sl.full_cols, f) != nr * sl.full_cols)
          return cleanup(fail(L2_E_ARG, "checkpoint %s truncated in tensor kind %d", path, kind));
        total += nr * sl.full_cols * sizeof(float);
        // intersect [r0, r0+nr) with the rank's rows [row0, row0+rows)
        const size_t a = r0 > sl.row0 ? r0 : sl.row0;
        const size_t b = (r0 + nr < sl.row0 + sl.rows) ? r0 + nr : sl.row0 + sl.rows;
        if (a < b) {
          const float* src = stage[cur] + (a - r0) * sl.full_cols + sl.col0;
          float* d = dst + (a - sl.row0) * sl.cols;
          hipError_t e = hipMemcpy2DAsync(d, sl.cols * sizeof(float), src, sl.full_cols * sizeof(float), sl.cols * sizeof(float),
                                          b - a, hipMemcpyHostToDevice, c->stream);
          if (e != hipSuccess) return cleanup(fail(L2_E_HIP, "hipMemcpy2DAsync: %s", hipGetErrorString(e)));
          hipEventRecord(done[cur], c->stream);
          pending[cur] = true;
        }
        cur ^= 1;
      }
      (void)full_layer;
      c->uploaded[kind][layer] = 1;
      mark_dirty(c, kind, (int)layer);
    }
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) return cleanup(fail(L2_E_HIP, "upload sync failed"));
  if (bytes_read) *bytes_read = total;
  *out = c;
  return cleanup(L2_OK);
}

// deterministic exp / sincos from IEEE basic operations (same recipe as the oracle's generator)
static double det_exp(double x) {
  const double y = x / 1024.0;
  double t = 1.0, s = 1.0;
  for (int k = 1; k <= 14; ++k) { t = (t * y) / (double)k; s = s + t; }
  for (int i = 0; i < 10; ++i) s = s * s;
  return s;
}
static void det_sincos(double x, double* sn, double* cs) {
  const double x2 = x * x;
  double ts = x, tc = 1.0, ss = x, cc = 1.0;
  for (int k = 1; k <= 12; ++k) {
    tc = ((-tc) * x2) / (double)((2 * k - 1) * (2 * k));
    cc = cc + tc;
    ts = ((-ts) * x2) / (double)((2 * k) * (2 * k + 1));
    ss = ss + ts;
  }
  *sn = ss; *cs = cc;
}

static uint64_t full_count(const l2_ctx* c, int kind) {
  if (kind == L2_T_WCLS && c->shared) return 0;
  const size_t d = c->d, h = c->h, V = c->V, S = c->S, hs2 = c->hs / 2, L = c->L;
  switch (kind) {
    case L2_T_TOKEN_EMBEDDING: case L2_T_WCLS: return V * d;
    case L2_T_RMS_ATT: case L2_T_RMS_FFN: return L * d;
    case L2_T_WQ: case L2_T_WO: return L * d * d;
    case L2_T_WK: case L2_T_WV: return L * (size_t)c->kvd * d;
    case L2_T_W1: case L2_T_W2: case L2_T_W3: return L * h * d;
    case L2_T_RMS_FINAL: return d;
    case L2_T_FREQ_REAL: case L2_T_FREQ_IMAG: return S * hs2;
    default: return 0;
  }
}

extern "C" int l2_synth_fill(l2_ctx* c, uint32_t seed) {
  if (!c) return fail(L2_E_ARG, "null context");
  HIPCHK(hipSetDevice(c->device));
  { const int rc_ = ensure_rowmajor(c, false); if (rc_) return rc_; }      // every tensor is overwritten: nothing to unpack
  uint64_t off = 0;
  for (int kind = 0; kind < L2_T_COUNT; ++kind) {
    const uint64_t n = full_count(c, kind);
    if (!n) continue;
    if (kind == L2_T_FREQ_REAL || kind == L2_T_FREQ_IMAG) {
      if (kind == L2_T_FREQ_REAL) {
        const int hs2 = c->hs / 2;
        std::vector<float> re((size_t)c->S * hs2), im((size_t)c->S * hs2);
        for (int j = 0; j < hs2; ++j) {
          const double theta = det_exp(-(((2.0 * (double)j) / (double)c->hs) * 9.210340371976184));
          double st, ct;
          det_sincos(theta, &st, &ct);
          double cr = 1.0, ci = 0.0;
          for (int t = 0; t < c->S; ++t) {
            re[(size_t)t * hs2 + j] = (float)cr;
            im[(size_t)t * hs2 + j] = (float)ci;
            const double nr = cr * ct - ci * st, ni = cr * st + ci * ct;
            cr = nr; ci = ni;
          }
        }
        HIPCHK(hipMemcpy(c->w[L2_T_FREQ_REAL], re.data(), re.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->w[L2_T_FREQ_IMAG], im.data(), im.size() * 4, hipMemcpyHostToDevice));
      }
    } else {
      double sigma = 0.0; float bias = 0.0f;
      switch (kind) {
        case L2_T_TOKEN_EMBEDDING: case L2_T_WCLS: sigma = 0.02; break;
        case L2_T_RMS_ATT: case L2_T_RMS_FFN: case L2_T_RMS_FINAL: sigma = 0.1; bias = 1.0f; break;
        case L2_T_W2: sigma = 1.0 / sqrt((double)c->h); break;
        default: sigma = 1.0 / sqrt((double)c->d); break;
      }
      const float scale = (float)(sigma / 37837.22723720648);
      // the rank's slice of every layer (whole tensor when not sharded); a shared classifier aliases the table
      const Slice sl = tensor_slice(c, kind);
      SynthSlice ss;
      ss.g0 = off; ss.full_layer = sl.full_rows * sl.full_cols; ss.rows = sl.rows; ss.cols = sl.cols;
      ss.full_cols = sl.full_cols; ss.row0 = sl.row0; ss.col0 = sl.col0;
      ss.n = sl.rows * sl.cols * (uint64_t)c->layers_of[kind];
      const uint64_t want = (ss.n + 256 * 8 - 1) / (256 * 8);
      const int blocks = (int)(want > 65535 ? 65535 : (want < 1 ? 1 : want));
      hipLaunchKernelGGL(synth_fill_kernel, dim3(blocks), dim3(256), 0, c->stream, c->w[kind], ss, seed, scale, bias);
      HIPCHK(hipGetLastError());
    }
    for (auto& u : c->uploaded[kind]) u = 1;
    mark_dirty(c, kind, -1);
    off += n;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return L2_OK;
}

extern "C" int l2_read_tensor(l2_ctx* c, int kind, int layer, size_t offset, float* out, size_t n_floats) {
  if (!c || !out) return fail(L2_E_ARG, "null argument");
  if (kind < 0 || kind >= L2_T_COUNT) return fail(L2_E_ARG, "tensor kind %d out of range", kind);
  int li = 0;
  if (is_layered(kind)) { if (layer < 0 || layer >= c->L) return fail(L2_E_ARG, "layer out of range"); li = layer; }
  const size_t n = c->layer_elems[kind] ? c->layer_elems[kind] : (size_t)c->V * c->d;
  if (offset + n_floats > n) return fail(L2_E_ARG, "read of %zu floats at %zu exceeds tensor (%zu)", n_floats, offset, n);
  HIPCHK(hipSetDevice(c->device));
  // out of the repacked copy; the next step (or the next l2_upload's repack) gives the tensors away again -- several reads in a row
  // (a parity check walks every layer) unpack once
  if (c->released[kind]) { const int rc_ = ensure_rowmajor(c, true); if (rc_) return rc_; c->rerelease = true; }
  HIPCHK(hipMemcpy(out, c->w[kind] + n * (size_t)li + offset, n_floats * 4, hipMemcpyDeviceToHost));
  return L2_OK;
}

// ------------------------------------------------------------------------------------------------
