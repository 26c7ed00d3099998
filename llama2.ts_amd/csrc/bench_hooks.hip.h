// bench_hooks.hip.h -- measurement entry points (bench.py, tools/): timed decode, in-situ probe of the dominant kernel, stream timer, one GEMV phase back to back
// Part of the one translation unit llama2_hip.hip (included there, in order); not a stand-alone header.
#pragma once

extern "C" int l2_bench_decode(l2_ctx* c, int first_token, int pos0, int steps, float* total_ms) {
  if (!total_ms) return fail(L2_E_ARG, "null total_ms");
  return run_greedy(c, first_token, pos0, steps, true, total_ms);
}

// The tokens the last device-resident run chose (l2_bench_decode keeps them on the device like l2_decode_greedy does): bench.py
// checks the run it TIMED against the reference's golden tokens, not a second run.
extern "C" int l2_bench_tokens(l2_ctx* c, int32_t* tokens_out, int n) {
  if (!c || (!tokens_out && n > 0)) return fail(L2_E_ARG, "null argument");
  if (n < 0 || n > c->S) return fail(L2_E_ARG, "%d tokens asked for, a run holds at most seq_len = %d", n, c->S);
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (n > 0) HIPCHK(hipMemcpy(tokens_out, c->d_tokens, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  return L2_OK;
}

// The dominant kernel (rmsnorm + w1/w3 GEMV + SwiGLU) timed IN SITU: `steps` greedy decode steps launched eagerly
// with a HIP event pair around every one of its launches on the library's stream; mean duration in microseconds.
extern "C" int l2_bench_dominant_in_situ(l2_ctx* c, int first_token, int pos0, int steps, float* avg_us, int* launches) {
  if (!c || !avg_us) return fail(L2_E_ARG, "null argument");
  if (steps <= 0 || pos0 < 0 || pos0 + steps > c->S) return fail(L2_E_ARG, "bad step range");
  HIPCHK(hipSetDevice(c->device));
  const size_t need = (size_t)2 * c->L * steps;
  while (c->probe.size() < need) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->probe.push_back(e); }
  const int saved_graph = c->opt_graph;
  c->opt_graph = 0;
  c->probe_used = 0; c->probe_on = true;
  int rc = run_greedy(c, first_token, pos0, steps, false, nullptr);
  c->probe_on = false; c->opt_graph = saved_graph;
  if (rc) return rc;
  double total = 0.0;
  size_t n = 0;
  for (size_t i = 0; i + 1 < c->probe_used; i += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->probe[i], c->probe[i + 1]));
    total += ms; ++n;
  }
  if (!n) return fail(L2_E_STATE, "no launches were probed");
  *avg_us = (float)(1e3 * total / (double)n);
  if (launches) *launches = (int)n;
  return L2_OK;
}

extern "C" int l2_timer_start(l2_ctx* c) {
  if (!c) return fail(L2_E_ARG, "null context");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventRecord(c->ev0, c->stream));
  return L2_OK;
}

extern "C" int l2_timer_stop(l2_ctx* c, float* ms) {
  if (!c || !ms) return fail(L2_E_ARG, "null argument");
  HIPCHK(hipEventRecord(c->ev1, c->stream));
  HIPCHK(hipEventSynchronize(c->ev1));
  HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return L2_OK;
}

// The dominant kernel alone: one weight-streaming GEMV phase, launched `iters` times back to back.
extern "C" int l2_bench_gemv(l2_ctx* c, int kind, int layer, int iters, float* avg_ms) {
  if (!c || !avg_ms || iters <= 0) return fail(L2_E_ARG, "bad argument");
  if (layer < 0 || layer >= c->L) layer = 0;
  int rc = ensure_ready(c);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  // the phases read {token, pos} from device memory: whatever a previous decode left there may be pos == seq_len
  memset(c->h_tokpos, 0, 4 * sizeof(int));
  HIPCHK(hipMemcpyAsync(c->tokpos, c->h_tokpos, 4 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  PhaseArgs a;
  memset(&a, 0, sizeof(a));
  a.tokpos = c->tokpos; a.fr = c->w[L2_T_FREQ_REAL]; a.fi = c->w[L2_T_FREQ_IMAG]; a.head_size = c->hs; a.dim = c->d;
  a.inv_n = 1.0 / (double)c->d;
  const size_t loff = (size_t)layer * c->S * c->kvd_loc;
  int mode;
  switch (kind) {
    case L2_T_WQ: case L2_T_WK: case L2_T_WV:
      mode = MODE_QKV;
      a.w0 = wptr(c, L2_T_WQ, layer); a.w1 = wptr(c, L2_T_WK, layer);
      a.w2 = wptr(c, L2_T_WV, layer);
      a.in = c->xn; a.rmsw = c->w[L2_T_RMS_ATT] + (size_t)c->d * layer; a.out = c->q; a.out_k = c->kc + loff; a.out_v = c->vc + loff;
      a.n = c->d; a.rows = c->d_loc + 2 * c->kvd_loc; a.dim = c->d_loc; a.kv_dim = c->kvd_loc; break;
    case L2_T_WO:
      mode = MODE_WO; a.w0 = wptr(c, L2_T_WO, layer); a.in = c->xb; a.res = c->xn; a.out = c->xb2;
      a.n = c->d_loc; a.rows = c->d; break;
    case L2_T_W1: case L2_T_W3:
      mode = MODE_W13; a.w0 = wptr(c, L2_T_W1, layer); a.w1 = wptr(c, L2_T_W3, layer);
      a.in = c->xn; a.rmsw = c->w[L2_T_RMS_FFN] + (size_t)c->d * layer; a.out = c->hb; a.n = c->d; a.rows = c->h_loc; break;
    case L2_T_W2:
      mode = MODE_W2; a.w0 = wptr(c, L2_T_W2, layer); a.in = c->hb; a.res = c->xn; a.out = c->xb2;
      a.n = c->h_loc; a.rows = c->d; break;
    case L2_T_WCLS: case L2_T_TOKEN_EMBEDDING:
      mode = MODE_CLS; a.w0 = c->w[L2_T_WCLS]; a.in = c->xn; a.rmsw = c->w[L2_T_RMS_FINAL]; a.out = c->logits_loc; a.aux = c->xb2;
      a.n = c->d; a.rows = c->V_loc; break;
    default: return fail(L2_E_ARG, "tensor kind %d is not a GEMV matrix", kind);
  }
  a.wp = packed_of(c, mode, mode == MODE_CLS ? 0 : layer);
  for (int it = -2; it < iters; ++it) {
    if (it == 0) HIPCHK(hipEventRecord(c->ev0, c->stream));
    hipError_t e;
    switch (mode) {
      case MODE_QKV: e = launch_phase<MODE_QKV>(c, a, c->stream); break;
      case MODE_WO: e = launch_phase<MODE_WO>(c, a, c->stream); break;
      case MODE_W13: e = launch_phase<MODE_W13>(c, a, c->stream); break;
      case MODE_W2: e = launch_phase<MODE_W2>(c, a, c->stream); break;
      default: e = launch_phase<MODE_CLS>(c, a, c->stream); break;
    }
    if (e != hipSuccess) return fail(L2_E_HIP, "gemv launch: %s", hipGetErrorString(e));
  }
  HIPCHK(hipEventRecord(c->ev1, c->stream));
  HIPCHK(hipEventSynchronize(c->ev1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *avg_ms = ms / (float)iters;
  return L2_OK;
}
