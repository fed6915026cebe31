// libgantts_hip.so -- stand-alone operators of the C ABI (gt_op_*, gt_compute_distortions)
#include "engine_internal.hip.h"

using namespace gt;
// ------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------
extern "C" int gt_op_sequence_mask(const int64_t* lengths, int B, int T, float* mask, void* stream) {
  if (!lengths || !mask || B < 1 || T < 1) return fail(GT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(sequence_mask_kernel, dim3(cdiv((long)B * T, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)lengths, B, T, mask);
  LAUNCH_CHECK();
  return GT_OK;
}

extern "C" int gt_op_masked_mse(const float* input, const float* target, const float* mask, int B, int T, int D, float* loss_out,
                                float* grad_input, void* stream) {
  if (!input || !target) return fail(GT_ERR_INVALID, "null tensor");
  if (!mask) return fail(GT_ERR_INVALID, "Should provide either lengths or mask");  // seqloss.py:33-34
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  static thread_local Scratch tls_ws;     // grow-only, no per-call hipMalloc/hipFree (both synchronise the device)
  CHK(tls_ws.ensure(1024 + 1024 * sizeof(double)));
  void* ws = tls_ws.p;
  StepScalars* sc = (StepScalars*)ws;
  double* part = (double*)((char*)ws + 1024);
  hipLaunchKernelGGL(mask_sum_kernel, dim3(1), dim3(1024), 0, s, mask, (int)N, -1.f, (const double*)nullptr, sc);
  const int nblk = (int)std::min<long>(1000, cdiv(N * D, RED_THREADS * 4));
  hipLaunchKernelGGL(masked_sqerr_kernel, dim3(nblk), dim3(RED_THREADS), 0, s, input, D, target, D, mask, N, D, part, grad_input, D,
                     1.f, sc);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, part, nblk, &sc->s_mse);
  StepScalars h;
  hipError_t err = hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  if (err != hipSuccess) return fail(GT_ERR_HIP, "masked_mse: %s", hipGetErrorString(err));
  if (loss_out) *loss_out = (float)h.s_mse / h.tv;
  return GT_OK;
}

extern "C" int gt_compute_distortions(const float* y_static, const float* y_hat_static, int Ds, const void* stat_mean,
                                      const void* stat_std, int stats_f64, const int32_t* col_stat_host,
                                      const int32_t* col_role_host, int vuv_col, const int64_t* lengths_host, int B, int T,
                                      gt_distortion_sums* out, void* stream) {
  if (!y_static || !y_hat_static || !stat_mean || !stat_std || !col_stat_host || !col_role_host || !out)
    return fail(GT_ERR_INVALID, "null argument");
  if (Ds < 1 || B < 1 || T < 1 || vuv_col >= Ds) return fail(GT_ERR_DIM, "bad sizes: Ds=%d B=%d T=%d vuv_col=%d", Ds, B, T, vuv_col);
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  const int nblk = (int)std::min<long>(1024, cdiv(N, 4));
  std::vector<int> host(2 * Ds + B);
  for (int c = 0; c < Ds; ++c) {
    if (col_stat_host[c] < 0) return fail(GT_ERR_INVALID, "negative statistics index");
    host[c] = col_stat_host[c];
    host[Ds + c] = col_role_host[c];
  }
  for (int b = 0; b < B; ++b) {
    const int64_t n = lengths_host ? lengths_host[b] : T;
    if (n < 0 || n > T) return fail(GT_ERR_INVALID, "length %lld outside [0, T=%d]", (long long)n, T);
    host[2 * Ds + b] = (int)n;
  }
  // grow-only workspace shared by all calls of this thread: the function runs once per training step
  // (train.py:588-595) -- a hipMalloc/hipFree pair per call would synchronise the device every step
  static thread_local Scratch tls_ws;
  const size_t off_part = ((host.size() * sizeof(int) + 255) / 256) * 256;
  const size_t off_out = off_part + (size_t)nblk * DIST_NSUM * sizeof(double);
  CHK(tls_ws.ensure(off_out + DIST_NSUM * sizeof(double)));
  void* ws = tls_ws.p;
  int* d_int = (int*)ws;
  double* part = (double*)((char*)ws + off_part);
  double* d_out = (double*)((char*)ws + off_out);
  hipError_t err = hipMemcpyAsync(d_int, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s);
  if (err == hipSuccess) {
    if (stats_f64)
      hipLaunchKernelGGL(distortion_kernel<double>, dim3(nblk), dim3(256), 0, s, y_static, y_hat_static, Ds, (const double*)stat_mean,
                         (const double*)stat_std, d_int, d_int + Ds, vuv_col, d_int + 2 * Ds, B, T, part);
    else
      hipLaunchKernelGGL(distortion_kernel<float>, dim3(nblk), dim3(256), 0, s, y_static, y_hat_static, Ds, (const float*)stat_mean,
                         (const float*)stat_std, d_int, d_int + Ds, vuv_col, d_int + 2 * Ds, B, T, part);
    hipLaunchKernelGGL(distortion_finalize_kernel, dim3(1), dim3(64 * DIST_NSUM), 0, s, part, nblk, d_out);
    err = hipGetLastError();
  }
  double h[DIST_NSUM];
  if (err == hipSuccess) err = hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);   // also keeps `host` alive until the H2D is done
  if (err != hipSuccess) return fail(GT_ERR_HIP, "compute_distortions: %s", hipGetErrorString(err));
  out->s_mcd = h[0]; out->s_bap = h[1]; out->s_f0 = h[2]; out->n_voiced = h[3];
  out->n_vuv_err = h[4]; out->s_mse = h[5]; out->n_frames = h[6];
  return GT_OK;
}

extern "C" int gt_op_pad_sequences(const float* ragged, int D, const int64_t* start, const int64_t* len, int B, int T, float* out, int ld_out,
                                   void* stream) {
  if (!ragged || !start || !len || !out || D < 1 || B < 1 || T < 1 || ld_out < D) return fail(GT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(pad_sequences_kernel, dim3(cdiv((long)B * T * ld_out, 256)), dim3(256), 0, (hipStream_t)stream, ragged, D,
                     (const long*)start, (const long*)len, B, T, out, ld_out);
  LAUNCH_CHECK();
  return GT_OK;
}
extern "C" int gt_op_gather_cols(const float* in, int ld_in, const int32_t* idx, int n_idx, float* out, int ld_out,
                                 int out_col_offset, int64_t rows, void* stream) {
  if (!in || !out || n_idx < 0 || rows < 0) return fail(GT_ERR_INVALID, "bad argument");
  if (rows == 0 || n_idx == 0) return GT_OK;
  hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(rows * n_idx, 256)), dim3(256), 0, (hipStream_t)stream, in, ld_in, 0, idx, out,
                     ld_out, out_col_offset, (int)rows, n_idx);
  LAUNCH_CHECK();
  return GT_OK;
}

extern "C" int gt_op_mlpg_forward(gt_engine* e, const float* y, const float* R, int B, int T, float* y_static, void* stream) {
  CHK(check_common(e, B, T));
  if (!y || !R || !y_static) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  CHK(ensure_band(e, R, T, s));
  return mlpg_forward(e, y, e->Dout_cfg, e->d_scol, e->d_sstride, e->Ds, y_static, e->Ds, B, T, s);
}
extern "C" int gt_op_mlpg_backward(gt_engine* e, const float* g_static, const float* R, int B, int T, float* g_y, void* stream) {
  CHK(check_common(e, B, T));
  if (!g_static || !R || !g_y) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  CHK(ensure_band(e, R, T, s));
  return mlpg_backward(e, g_static, e->Ds, e->d_scol, e->d_sstride, e->Ds, g_y, e->Dout_cfg, B, T, 0.f, nullptr, nullptr, 0, nullptr, s);
}

static DropoutSpec buffer_spec(const float* keep_mask, float p, int ld) {
  DropoutSpec d = no_drop();
  if (keep_mask && p > 0.f) { d.mode = DROP_BUFFER; d.mask = keep_mask; d.ld_mask = ld; d.p = p; d.scale = 1.f / (1.f - p); }
  return d;
}

extern "C" int gt_op_linear_forward(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int64_t rows,
                                    int in_dim, int out_dim, int act, const float* keep_mask, float p, void* stream) {
  if (!X || !W || !Y || rows < 1 || in_dim < 1 || out_dim < 1) return fail(GT_ERR_INVALID, "bad argument");
  if (act < 0 || act > 2) return fail(GT_ERR_INVALID, "unknown activation");
  tl_gemm_prec = PREC_F32;
  return linear_forward(X, ldx, W, in_dim, bias, Y, ldy, rows, in_dim, out_dim, act, buffer_spec(keep_mask, p, out_dim), (hipStream_t)stream);
}

extern "C" int gt_op_linear_backward(const float* dY, int lddy, const float* X, int ldx, const float* W, int64_t rows, int in_dim,
                                     int out_dim, float* dX, int lddx, const float* H_prev, int act_prev,
                                     const float* keep_mask_prev, float p_prev, float* dW, float* db, void* stream) {
  if (!dY || rows < 1 || in_dim < 1 || out_dim < 1) return fail(GT_ERR_INVALID, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  tl_gemm_prec = PREC_F32;
  if (dX) {
    if (!W) return fail(GT_ERR_INVALID, "dX requested without W");
    if (act_prev != ACT_NONE && !H_prev) return fail(GT_ERR_INVALID, "activation derivative requested without H_prev");
    CHK(linear_backward_data(dY, lddy, W, in_dim, 0, dX, lddx, rows, out_dim, in_dim, act_prev, H_prev, in_dim,
                             buffer_spec(keep_mask_prev, p_prev, in_dim), s));
  }
  if (dW || db) {
    if (dW && !X) return fail(GT_ERR_INVALID, "dW requested without X");
    Scratch slabs, colp;
    int r = linear_backward_weight(dY, lddy, X, ldx, rows, out_dim, in_dim, dW, db, false, slabs, colp, s);
    hipError_t err = hipStreamSynchronize(s);
    slabs.release(); colp.release();
    if (r) return r;
    if (err != hipSuccess) return fail(GT_ERR_HIP, "linear_backward: %s", hipGetErrorString(err));
  }
  return GT_OK;
}

// nn.Linear forward / backward through the bf16-STORAGE products (gemm_bf16s.hip.h): operands are cast to bfloat16 images
// (both orientations) exactly as the engine keeps them with GT_OPT_MATMUL_BF16, results come back as float32.  Parity
// hook: against float64 arithmetic on the bf16-rounded operands the results agree to float32 accumulation error.
extern "C" int gt_op_linear_bf16(const float* X, const float* W, const float* bias, int64_t rows, int in_dim, int out_dim, int act,
                                 const float* keep_mask, float p, float* Y, const float* dY, const float* H_prev, int act_prev,
                                 const float* keep_mask_prev, float p_prev, float* dX, float* dW, float* db,
                                 float* Y_image, float* YT_image, void* stream) {
  if (!X || !W || rows < 1 || in_dim < 1 || out_dim < 1) return fail(GT_ERR_INVALID, "bad argument");
  if (act < 0 || act > 2 || act_prev < 0 || act_prev > 2) return fail(GT_ERR_INVALID, "unknown activation");
  hipStream_t s = (hipStream_t)stream;
  const int in8 = pad8(in_dim), out8 = pad8(out_dim);
  const long rows8 = pad8(rows);
  Scratch xb, xbt, wb, wbt, yb, ybt, dyb, dybt, hb, slabs, colp;
  int r = GT_OK;
  auto body = [&]() -> int {
    CHK(xb.ensure((size_t)rows * in8 * 2)); CHK(xbt.ensure((size_t)in_dim * rows8 * 2));
    CHK(wb.ensure((size_t)out_dim * in8 * 2)); CHK(wbt.ensure((size_t)in_dim * out8 * 2));
    CHK(cast_transpose(X, in_dim, rows, in_dim, xb.as<__bf16>(), in8, xbt.as<__bf16>(), rows8, nullptr, false, &colp, s));
    CHK(cast_transpose(W, in_dim, out_dim, in_dim, wb.as<__bf16>(), in8, wbt.as<__bf16>(), out8, nullptr, false, &colp, s));
    if (Y) {
      CHK(yb.ensure((size_t)rows * out8 * 2)); CHK(ybt.ensure((size_t)out_dim * rows8 * 2));
      GemmB16Args g = b16_args();
      g.A = xb.as<__bf16>(); g.lda = in8; g.B = wb.as<__bf16>(); g.ldb = in8; g.M = (int)rows; g.N = out_dim; g.K = in_dim;
      g.C = Y; g.ldc = out_dim; g.Cb = yb.as<__bf16>(); g.ldcb = out8; g.CbT = ybt.as<__bf16>(); g.ldcbt = (int)rows8;
      g.bias = bias; g.epi = B16_FWD; g.act = act; g.drop = buffer_spec(keep_mask, p, out_dim);
      CHK(launch_gemm_b16(g, 1, s));
      // the two bf16 images the same epilogue wrote ([frame][out] and [out][frame]), widened to float32 for inspection
      if (Y_image) {
        hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(cdiv(rows * out_dim, 256)), dim3(256), 0, s, (const __bf16*)yb.as<__bf16>(), (long)out8, rows, out_dim, Y_image);
        LAUNCH_CHECK();
      }
      if (YT_image) {
        hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(cdiv(rows * out_dim, 256)), dim3(256), 0, s, (const __bf16*)ybt.as<__bf16>(), rows8, (long)out_dim, (int)rows, YT_image);
        LAUNCH_CHECK();
      }
    }
    if (dY) {
      CHK(dyb.ensure((size_t)rows * out8 * 2)); CHK(dybt.ensure((size_t)out_dim * rows8 * 2));
      CHK(cast_transpose(dY, out_dim, rows, out_dim, dyb.as<__bf16>(), out8, dybt.as<__bf16>(), rows8, nullptr, false, &colp, s));
      if (dX) {
        GemmB16Args g = b16_args();
        g.A = dyb.as<__bf16>(); g.lda = out8; g.B = wbt.as<__bf16>(); g.ldb = out8; g.M = (int)rows; g.N = in_dim; g.K = out_dim;
        g.C = dX; g.ldc = in_dim; g.epi = B16_BWD_DATA; g.act = ACT_NONE;
        if (H_prev && act_prev != ACT_NONE) {
          CHK(hb.ensure((size_t)rows * in8 * 2));
          CHK(cast_transpose(H_prev, in_dim, rows, in_dim, hb.as<__bf16>(), in8, nullptr, 0, nullptr, false, &colp, s));
          g.act = act_prev; g.H = hb.as<__bf16>(); g.ldh = in8; g.drop = buffer_spec(keep_mask_prev, p_prev, in_dim);
        }
        CHK(launch_gemm_b16(g, 1, s));
      }
      if (dW) CHK(weight_grad_b16(dybt.as<__bf16>(), rows8, xbt.as<__bf16>(), rows8, rows, out_dim, in_dim, dW, db, false, slabs, s));
    }
    return GT_OK;
  };
  r = body();
  const hipError_t err = hipStreamSynchronize(s);
  for (Scratch* q : {&xb, &xbt, &wb, &wbt, &yb, &ybt, &dyb, &dybt, &hb, &slabs, &colp}) q->release();
  if (r) return r;
  if (err != hipSuccess) return fail(GT_ERR_HIP, "linear_bf16: %s", hipGetErrorString(err));
  return GT_OK;
}

