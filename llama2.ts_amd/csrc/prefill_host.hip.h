// prefill_host.hip.h -- host side of batched prompt ingestion: l2_prefill and its launch sequence
// Part of the one translation unit llama2_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// ---- prefill -----------------------------------------------------------------
static bool can_prefill(const l2_ctx* c) {
  return !c->tp_path && c->kvd == c->d && (c->d % 16 == 0) && (c->h % 16 == 0) && (c->hs % 4 == 0) && attn_vec(c);
}

// One prefill GEMM.  `tt` = tiles of 16 tokens in the chunk (1, 2 or 4).  (The 16-row-tile kernel's LDS-tile variant -- short chunks over
// row-major tensors only -- went in round 5: 8 instances, one of them at 256 VGPRs + 134 AGPRs, for chunks of at most 32 tokens.)
// register-blocked form (prefill.hip.h: pf_gemm3_kernel): RT row tiles per wave, 4 waves split K, `chunks` 64-token chunks per launch
template <int MODE, int RT, bool F32 = false>
static void launch_pf3(const PfArgs& a, int chunks, hipStream_t st) {
  constexpr int NW = 4;
  const size_t lds = (size_t)4 * NW * 4 * 64 * (F32 ? 4 : 8);
  hipLaunchKernelGGL((pf_gemm3_kernel<MODE, NW, RT, 4, F32>), dim3(a.rows / (16 * RT), chunks), dim3(64 * NW), lds, st, a);
}

// Shapes the register-blocked GEMMs cover: whole batches of two 16-column blocks (n % 32) of both input widths (qkv's 3 d / 16 row
// tiles always come in threes).  stories15M (288 / 768) qualifies; the test shapes with hidden sizes like 176 keep the 16-row-tile kernels.
static bool pf3_ok(const l2_ctx* c) { return c->pf3 && c->d % 32 == 0 && c->h % 32 == 0; }

template <int MODE>
static void launch_pf_gemm(const l2_ctx* c, const PfArgs& a, int nw, int tt, int chunks, hipStream_t st) {
  if (pf3_ok(c) && tt == 4 && c->opt_pf_f32) {
    // L2_OPT_PREFILL_F32_MFMA (opt-in): the same blocking on v_mfma_f32_16x16x4_f32 -- fp32 accumulate, NOT the reference's arithmetic.  A
    // result tile is four registers, not eight, and an MFMA takes 32 cycles, not 64: with the fp64 form's row tiles per wave the operand
    // fragments (re-read from L2 by every wave) would need ~24 B / clock / CU, so with FOUR chunks in the launch (enough workgroups either
    // way) a wave takes more row tiles: q / k / v four, w1 / w3 two pairs (7B, 256 tokens: 6 510 -> 6 940 tok/s); with fewer chunks the
    // fp64 form's counts (more tiles per wave at 128 tokens left CUs idle: 5 520 -> 4 910)
    const int tiles = a.rows / 16;
    if constexpr (MODE == MODE_QKV) { if (chunks == 4 && tiles % 4 == 0) launch_pf3<MODE, 4, true>(a, chunks, st); else launch_pf3<MODE, 3, true>(a, chunks, st); return; }
    else if constexpr (MODE == MODE_W13) { if (chunks == 4 && tiles % 2 == 0) launch_pf3<MODE, 2, true>(a, chunks, st); else launch_pf3<MODE, 1, true>(a, chunks, st); return; }
    else {
      if (chunks == 4 && tiles % 4 == 0) launch_pf3<MODE, 4, true>(a, chunks, st);
      else if (chunks == 2 && tiles % 2 == 0) launch_pf3<MODE, 2, true>(a, chunks, st);
      else launch_pf3<MODE, 1, true>(a, chunks, st);
      return;
    }
  }
  if (pf3_ok(c) && tt == 4) {
    // row tiles per wave: conversions per MFMA are 16 (R + 64) / (64 R) for R rows per workgroup, so as many as still leave >= 256
    // workgroups: qkv 3 (3 d / 16 tiles), w1 / w3 one pair (688 pairs at 7B), wo / w2 (d / 16 tiles) 1, 2 or 4 with the chunk count
    if constexpr (MODE == MODE_QKV) { launch_pf3<MODE, 3>(a, chunks, st); return; }
    else if constexpr (MODE == MODE_W13) { launch_pf3<MODE, 1>(a, chunks, st); return; }
    else {
      const int tiles = a.rows / 16;
      if (chunks == 4 && tiles % 4 == 0) launch_pf3<MODE, 4>(a, chunks, st);
      else if (chunks == 2 && tiles % 2 == 0) launch_pf3<MODE, 2>(a, chunks, st);
      else launch_pf3<MODE, 1>(a, chunks, st);
      return;
    }
  }
  const dim3 grid(a.rows / 16);
  // four waves split K (the eight-wave instances spilled and were never launched: removed)
  (void)nw;
  if (tt == 4) hipLaunchKernelGGL((pf_gemm_kernel<MODE, 4, 4>), grid, dim3(256), 0, st, a);
  else if (tt == 2) hipLaunchKernelGGL((pf_gemm_kernel<MODE, 4, 2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((pf_gemm_kernel<MODE, 4, 1>), grid, dim3(256), 0, st, a);
}

// The weights of one prompt GEMM: the row-major tensors of layer l, or -- once they have been given back (one copy of the weights:
// ensure_packed) -- the decode step's repacked copy of this phase with the geometry it was packed for.
template <int MODE>
static void pf_weights(const l2_ctx* c, int l, PfArgs& a, int k0, int k1, int k2) {
  a.w0 = a.w1 = a.w2 = nullptr; a.wp = nullptr; a.pk_wstride = 0; a.pk_groups = 0;
  if (c->released[k0]) {
    const l2_ctx::Packed& p = c->packed[MODE];
    a.wp = p.buf + p.layer_elems * (size_t)l;
    a.pk_wstride = p.grid * p.nwaves;
    a.pk_groups = (int)(p.layer_elems / (2 * (size_t)(MODE == MODE_W2 ? c->h : c->d)));      // layer_elems = groups * 2 rows * n
    return;
  }
  a.w0 = c->w[k0] + c->layer_elems[k0] * l;
  if (k1 >= 0) a.w1 = c->w[k1] + c->layer_elems[k1] * l;
  if (k2 >= 0) a.w2 = c->w[k2] + c->layer_elems[k2] * l;
}

// One launch sequence for up to PF_S chunks of PF_T prompt positions (n tokens at pos0 ...): every GEMM sees all of them.
static int prefill_chunk(l2_ctx* c, const int32_t* tokens, int n, int pos0) {
  hipStream_t st = c->stream;
  const size_t d = c->d, h = c->h;
  constexpr size_t ROWS = (size_t)PF_S * PF_T;
  if (!c->pf_x) {
    HIPCHK(hipMalloc(&c->pf_x, ROWS * d * 4)); HIPCHK(hipMalloc(&c->pf_xn, ROWS * (d > h ? d : h) * 4));
    HIPCHK(hipMalloc(&c->pf_q, ROWS * d * 4)); HIPCHK(hipMalloc(&c->pf_xb, ROWS * d * 4)); HIPCHK(hipMalloc(&c->pf_hb, ROWS * h * 4));
    HIPCHK(hipMalloc(&c->pf_tok, ROWS * sizeof(int)));
    HIPCHK(hipMemset(c->pf_xb, 0, ROWS * d * 4)); HIPCHK(hipMemset(c->pf_q, 0, ROWS * d * 4));
  }
  const int chunks = (n + PF_T - 1) / PF_T;                              // > 1 only on the register-blocked path (l2_prefill)
  const int tt = (n > 32) ? 4 : (n > 16) ? 2 : 1, nt = (chunks > 1) ? chunks * PF_T : 16 * tt;   // token rows the kernels see (whole 16-token MFMA tiles)
  int32_t tk[ROWS] = {0};
  for (int i = 0; i < n; ++i) tk[i] = tokens[i];
  HIPCHK(hipMemcpyAsync(c->pf_tok, tk, sizeof(tk), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));   // tk is on the stack
  hipLaunchKernelGGL(pf_embed_kernel, dim3(nt), dim3(256), 0, st, c->pf_x, c->w[L2_T_TOKEN_EMBEDDING], c->pf_tok, c->d, n);
  LCHK(hipGetLastError());
  for (int l = 0; l < c->L; ++l) {
    const size_t loff = (size_t)l * c->S * c->d;
    PfArgs a;
    memset(&a, 0, sizeof(a));
    a.fr = c->w[L2_T_FREQ_REAL]; a.fi = c->w[L2_T_FREQ_IMAG]; a.head_size = c->hs; a.dim = c->d; a.pos0 = pos0; a.nvalid = n;
    a.x = c->pf_x;
    // rmsnorm + q,k,v + RoPE + cache rows (llama2.ts:216-240)
    hipLaunchKernelGGL(pf_norm_kernel, dim3(nt), dim3(256), 0, st, c->pf_xn, c->pf_x, c->w[L2_T_RMS_ATT] + d * l, c->d);
    pf_weights<MODE_QKV>(c, l, a, L2_T_WQ, L2_T_WK, L2_T_WV);
    a.xin = c->pf_xn; a.out = c->pf_q; a.kc = c->kc + loff; a.vc = c->vc + loff; a.n = c->d; a.rows = 3 * c->d;
    launch_pf_gemm<MODE_QKV>(c, a, 4, tt, chunks, st);
    LCHK(hipGetLastError());
    // attention, one workgroup per (head, query) (llama2.ts:244-267)
    const size_t alds = pf_attn_lds(pos0 + ((n + 15) & ~15) - 1);
    if (c->pf_attn && !c->opt_exact && (c->hs == 64 || c->hs == 128) && alds <= 150 * 1024) {
      // 16 queries per workgroup on the fp64 MFMA (prefill.hip.h: pf_attn_mfma_kernel)
      PfAttnArgs pa;
      pa.q = c->pf_q; pa.kc = c->kc + loff; pa.vc = c->vc + loff; pa.xb = c->pf_xb;
      pa.dim = c->d; pa.head_size = c->hs; pa.seq_len = c->S; pa.pos0 = pos0; pa.nvalid = n;
      pa.inv_sqrt_hs = 1.0 / sqrt((double)c->hs);
      const dim3 grid(c->H, (n + 15) / 16);
      if (c->hs == 128) {
        LCHK(lds_opt_in(&pf_attn_mfma_kernel<128>, alds));
        hipLaunchKernelGGL((pf_attn_mfma_kernel<128>), grid, dim3(256), alds, st, pa);
      } else {
        LCHK(lds_opt_in(&pf_attn_mfma_kernel<64>, alds));
        hipLaunchKernelGGL((pf_attn_mfma_kernel<64>), grid, dim3(256), alds, st, pa);
      }
      LCHK(hipGetLastError());
    } else {   // one workgroup per (head, query): the decode kernel (other head sizes, the exact accumulate, very long contexts)
      AttnArgs aa;
      c->cur_splits = 1; c->cur_fused = false;
      fill_attn_args(c, l, aa);
      aa.q = c->pf_q; aa.xb = c->pf_xb; aa.att = nullptr; aa.pos_plus1 = 1;
      LCHK(launch_attn_tile(c, aa, n, pos0, st));
    }
    // wo + residual (llama2.ts:270-273)
    pf_weights<MODE_WO>(c, l, a, L2_T_WO, -1, -1); a.xin = c->pf_xb; a.n = c->d; a.rows = c->d;
    launch_pf_gemm<MODE_WO>(c, a, 4, tt, chunks, st);
    // rmsnorm + w1,w3 + SwiGLU (llama2.ts:276-289)
    hipLaunchKernelGGL(pf_norm_kernel, dim3(nt), dim3(256), 0, st, c->pf_xn, c->pf_x, c->w[L2_T_RMS_FFN] + d * l, c->d);
    pf_weights<MODE_W13>(c, l, a, L2_T_W1, L2_T_W3, -1);
    a.xin = c->pf_xn; a.out = c->pf_hb; a.n = c->d; a.rows = c->h;
    launch_pf_gemm<MODE_W13>(c, a, 4, tt, chunks, st);
    // w2 + residual (llama2.ts:292-295)
    pf_weights<MODE_W2>(c, l, a, L2_T_W2, -1, -1); a.xin = c->pf_hb; a.n = c->h; a.rows = c->d;
    launch_pf_gemm<MODE_W2>(c, a, 4, tt, chunks, st);
    LCHK(hipGetLastError());
  }
  return L2_OK;
}

extern "C" int l2_prefill(l2_ctx* c, const int32_t* tokens, int n_tokens, int pos0, float* logits_out) {
  if (!c || !tokens) return fail(L2_E_ARG, "null argument");
  if (n_tokens <= 0 || pos0 < 0 || pos0 + n_tokens > c->S) return fail(L2_E_ARG, "positions %d..%d outside [0, seq_len=%d)", pos0, pos0 + n_tokens - 1, c->S);
  for (int i = 0; i < n_tokens; ++i) if (tokens[i] < 0 || tokens[i] >= c->V) return fail(L2_E_ARG, "token %d outside [0, vocab_size=%d)", tokens[i], c->V);
  int rc = ensure_ready(c);
  if (rc) return rc;
  // a single token: the decode step streams the weights once at full rate; the 16-token tile pass does not (7.7 against 4.3 ms at 7B)
  if (n_tokens == 1 || !can_prefill(c)) {   // and shapes the 16x16 tiles do not cover: the reference's own one-token-per-call loop
    for (int i = 0; i < n_tokens; ++i) { rc = l2_forward(c, tokens[i], pos0 + i, (i == n_tokens - 1) ? logits_out : nullptr); if (rc) return rc; }
    return L2_OK;
  }
  HIPCHK(hipSetDevice(c->device));
  if (c->opt_pos_check && pos0 != 0 && pos0 > c->next_pos)
    return fail(L2_E_STATE, "L2_CHECK_POS: pos %d skips ahead of the sequence (cache rows 0 .. %d have been written)", pos0, c->next_pos - 1);
  if (pos0 + n_tokens > c->next_pos || pos0 == 0) c->next_pos = pos0 + n_tokens;
  const int step = pf3_ok(c) ? PF_S * PF_T : PF_T;      // positions per launch sequence: several 64-token chunks where the register-blocked GEMMs apply
  int done = 0;
  while (done < n_tokens) {
    const int n = (n_tokens - done < step) ? n_tokens - done : step;
    rc = prefill_chunk(c, tokens + done, n, pos0 + done);
    if (rc) return rc;
    done += n;
  }
  // logits of the last position only (llama2.ts:299-302): the decode classifier on the last row of the chunk
  const int last = (n_tokens - 1) % step;
  c->h_tokpos[0] = tokens[n_tokens - 1]; c->h_tokpos[1] = pos0 + n_tokens - 1; c->h_tokpos[2] = 0; c->h_tokpos[3] = 0;
  HIPCHK(hipMemcpyAsync(c->tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  PhaseArgs a = cls_args(c, true);
  a.in = c->pf_x + (size_t)last * c->d;
  LCHK(launch_phase<MODE_CLS>(c, a, c->stream));
  if (!(c->opt_zero_copy && !c->tp_path))
    HIPCHK(hipMemcpyAsync(c->h_logits, c->logits, (size_t)c->V * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->ran_forward = true;
  if (logits_out) memcpy(logits_out, c->h_logits, (size_t)c->V * 4);
  return L2_OK;
}
