// Operator-level C-ABI entry points (include/mapnet_hip.h, "Operator-level entry points").
#include "../../include/mapnet_hip.h"

#include <string>

#include "common.h"
#include "criterion.h"
#include "elementwise.h"
#include "head.h"
#include "dense.h"
#include "igemm.h"
#include "dgrad.h"
#include "halo_pp.h"
#include "optim.h"
#include "pgo.h"
#include "pool.h"
#include "rehearsal.h"
#include "stem.h"
#include "stem_bwd.h"
#include "wgrad.h"
#include "util.h"

using namespace mn;

namespace mn {
thread_local std::string g_last_error;
}

extern "C" const char* mn_last_error(void) { return mn::g_last_error.c_str(); }

#ifndef MN_BACKEND_NAME
#define MN_BACKEND_NAME "hip"
#endif
extern "C" const char* mn_backend(void) { return MN_BACKEND_NAME; }

static GatherGeom to_geom(const mn_gather_geom* g) {
  GatherGeom r;
  r.B = g->B; r.Hi = g->Hi; r.Wi = g->Wi; r.C = g->C; r.P = g->P; r.Q = g->Q; r.R = g->R; r.S = g->S;
  r.mul_p = g->mul_p; r.mul_q = g->mul_q; r.rsign = g->rsign; r.ssign = g->ssign;
  r.off_h = g->off_h; r.off_w = g->off_w; r.div = g->div; r.M = g->M; r.N = g->N; r.K = g->K;
  return r;
}

static int check_geom(const GatherGeom& g, int dtype) {
  int vec = dtype == MN_F16 ? 8 : 4;
  if ((dtype == MN_DTYPE_F16X2 || dtype == MN_DTYPE_F16X2Q) && (g.C % 32 != 0 || g.N % 4 != 0))
    return fail("igemm: h2 operands need C % 32 == 0");
  if (g.C % vec != 0) return fail("igemm: C must be a multiple of the 16-byte piece");
  if (g.K != g.R * g.S * g.C) return fail("igemm: K != R*S*C");
  if (g.K % (4 * vec) != 0) return fail("igemm: K must be a multiple of the 64-byte K-step");
  if (g.M != g.B * g.P * g.Q) return fail("igemm: M != B*P*Q");
  if (g.div != 1 && g.div != 2) return fail("igemm: div must be 1 or 2");
  if (g.div == 2 && (g.rsign != -1 || g.ssign != -1)) return fail("igemm: div = 2 requires rsign = ssign = -1");
  if (g.R * g.S > 32) return fail("igemm: at most 32 taps");
  if ((long)g.B * g.Hi * g.Wi * g.C >= (1L << 31)) return fail("igemm: A tensor exceeds 2^31 elements");
  return 0;
}

extern "C" int mn_op_igemm_grid_m(int M) { return igemm_grid_m(M); }

extern "C" int mn_op_igemm(int dtype, const mn_gather_geom* gg, const void* A, const void* Bw, void* out, int ldc,
                           float* stats, const float* bias, int relu, const void* res, const void* res_gate, float alpha,
                           const void* zero_page, void* stream) {
  begin_call();
  GatherGeom g = to_geom(gg);
  if (int e = check_geom(g, dtype)) return e;
  if (!zero_page) return fail("igemm: zero_page (>= 16 zero bytes of device memory) is required");
  {
    const long es = dtype == MN_DTYPE_F16 ? 2 : 4;
    if ((long)gg->B * gg->Hi * gg->Wi * gg->C * es >= 0xfffffff0l || (long)gg->N * gg->K * es >= 0xfffffff0l)
      return fail("igemm: operands must be smaller than 4 GiB (32-bit buffer offsets)");
  }
  Epilogue ep;
  ep.out = out; ep.ldc = ldc; ep.stats = stats; ep.bias = bias; ep.relu = relu; ep.res = res; ep.res_gate = res_gate;
  ep.alpha = alpha;
  if (dtype == MN_DTYPE_F32X3) g.mma = MMA_F16X3;  // fp32 tensors, f16 matrix pipe with split operands
  if (dtype == MN_DTYPE_F16X2 || dtype == MN_DTYPE_F16X2Q) {  // h2 / h2q operands (A, Bw, res_gate), fp32 out / res
    launch_igemm_h2(g, (const half*)A, (const half*)Bw, ep, (hipStream_t)stream, (const half*)zero_page, dtype == MN_DTYPE_F16X2Q);
    return check_launch("igemm");
  }
  if (dtype == MN_F16)
    launch_igemm<half>(g, (const half*)A, (const half*)Bw, ep, (hipStream_t)stream, (const half*)zero_page);
  else
    launch_igemm<float>(g, (const float*)A, (const float*)Bw, ep, (hipStream_t)stream, (const float*)zero_page);
  return check_launch("igemm");
}

extern "C" int mn_op_conv_halo_pp(const mn_gather_geom* gg, const void* A, const void* Bw, void* out, int ldc, double* stats_accum,
                                 int stats_rows, int relu, const void* res, const void* res_gate, const void* out_gate,
                                 float alpha, int wgs, void* stream) {
  begin_call();
  GatherGeom g = to_geom(gg);
  if (int e = check_geom(g, MN_F16)) return e;
  Epilogue ep;
  ep.out = out; ep.ldc = ldc; ep.stats = nullptr; ep.bias = nullptr; ep.relu = relu; ep.res = res; ep.res_gate = res_gate;
  ep.out_gate = out_gate; ep.alpha = alpha; ep.stats_accum = stats_accum; ep.stats_rows = stats_rows;
  if (!conv_halo_pp_applies(g, ep))
    return fail("conv_halo_pp: fp16 3x3 stride-1 same-size convolutions of 64 -> 64 channels only (stats_rows > 0 with stats_accum)");
  launch_conv_halo_pp(g, (const half*)A, (const half*)Bw, ep, (hipStream_t)stream, wgs);
  return check_launch("conv_halo_pp");
}

static int op_wgrad(int dtype, const mn_gather_geom* gg, const void* dY, int ldy, const void* X, float* dW, int ldw,
                    const int32_t* colmap, float alpha, int target_blocks, const void* zero_page, float* ws, long ws_floats,
                    void* stream);
extern "C" int mn_op_conv_halo_h2(const mn_gather_geom* gg, const void* A, const void* Bw, float* out, int ldc, double* stats_accum,
                                  int stats_rows, void* stream) {
  begin_call();
  GatherGeom g = to_geom(gg);
  if (int e = check_geom(g, MN_DTYPE_F16X2)) return e;
  Epilogue ep;
  ep.out = out; ep.ldc = ldc; ep.stats = nullptr; ep.bias = nullptr; ep.relu = 0; ep.res = nullptr; ep.res_gate = nullptr;
  ep.out_gate = nullptr; ep.alpha = 1.f; ep.stats_accum = stats_accum; ep.stats_rows = stats_rows;
  if (!conv_halo_h2_applies(g, ep))
    return fail("conv_halo_h2: 3x3 stride-1 same-size forward convolutions of 64 -> 64 channel h2 tensors only (stats_rows > 0 with stats_accum)");
  launch_conv_halo_h2(g, (const half*)A, (const half*)Bw, ep, (hipStream_t)stream);
  return check_launch("conv_halo_h2");
}

extern "C" int mn_op_wgrad(int dtype, const mn_gather_geom* gg, const void* dY, int ldy, const void* X, float* dW, int ldw,
                           const int32_t* colmap, float alpha, int target_blocks, const void* zero_page, void* stream) {
  return op_wgrad(dtype, gg, dY, ldy, X, dW, ldw, colmap, alpha, target_blocks, zero_page, nullptr, 0, stream);
}
extern "C" int64_t mn_op_wgrad_ws_floats(void) { return wgrad_fused_ws_floats(WGF_BLOCKS); }
extern "C" int mn_op_wgrad_ws(int dtype, const mn_gather_geom* gg, const void* dY, int ldy, const void* X, float* dW, int ldw,
                              float alpha, float* ws, int64_t ws_floats, const void* zero_page, void* stream) {
  if (!ws || ws_floats <= 0) return fail("wgrad_ws: workspace required (mn_op_wgrad_ws_floats() floats)");
  return op_wgrad(dtype, gg, dY, ldy, X, dW, ldw, nullptr, alpha, 512, zero_page, ws, (long)ws_floats, stream);
}
static int op_wgrad(int dtype, const mn_gather_geom* gg, const void* dY, int ldy, const void* X, float* dW, int ldw,
                    const int32_t* colmap, float alpha, int target_blocks, const void* zero_page, float* ws, long ws_floats,
                    void* stream) {
  begin_call();
  WgradArgs a;
  a.ws = ws;
  a.ws_floats = ws_floats;
  a.g = to_geom(gg);
  int vec = dtype == MN_F16 ? 8 : 4;
  if (a.g.C % vec != 0 || a.g.N % vec != 0) return fail("wgrad: channel counts must be multiples of the piece");
  if (dtype == MN_DTYPE_F16X2 && (a.g.C % 32 != 0 || ldy % 32 != 0)) return fail("wgrad: h2 operands need channel counts % 32 == 0");
  a.dY = dY; a.ldy = ldy; a.X = X; a.dW = dW; a.ldw = ldw; a.colmap = colmap; a.alpha = alpha; a.rows_per_split = 0;
  if (dtype == MN_DTYPE_F32X3) a.g.mma = MMA_BF16X3;  // fp32 tensors, bf16 matrix pipe with split operands
  if (dtype == MN_DTYPE_F16X2) a.g.mma = MMA_H2;      // h2 tensors (dY, X), fp32 dW
  if (dtype == MN_F16 || dtype == MN_DTYPE_F16X2)
    launch_wgrad<half>(a, target_blocks, (hipStream_t)stream, zero_page);
  else
    launch_wgrad<float>(a, target_blocks, (hipStream_t)stream);
  return check_launch("wgrad");
}

extern "C" int mn_op_stem_conv(const void* xpad, const void* wf, void* y, double* stats_accum, int stats_rows, int B, int H, int W,
                               int Wp, void* stream) {
  begin_call();
  if (Wp % 2 != 0 || Wp < W + 7) return fail("stem_conv: Wp must be even and >= W + 7");
  if ((long)B * (H + 6) * Wp * 8 >= 0xfffffff0l) return fail("stem_conv: padded input exceeds 4 GiB");
  launch_stem_conv((const half*)xpad, (const half*)wf, (half*)y, stats_accum, stats_rows, B, H, W, Wp, (hipStream_t)stream);
  return check_launch("stem_conv");
}

extern "C" int mn_op_stem_conv_x3(const float* xpad, const float* wf, float* y, double* stats_accum, int stats_rows, int B, int H,
                                  int W, int Wp, void* stream) {
  begin_call();
  if (Wp % 2 != 0 || Wp < W + 7) return fail("stem_conv_x3: Wp must be even and >= W + 7");
  launch_stem_conv_x3(xpad, wf, y, stats_accum, stats_rows, B, H, W, Wp, (hipStream_t)stream);
  return check_launch("stem_conv_x3");
}

extern "C" int mn_op_dense(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int relu, void* stream) {
  begin_call();
  DenseArgs a;
  a.A = A; a.W = W; a.bias = bias; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N; a.relu = relu;
  if (M <= 0 || N <= 0 || !dense_nt_applies(a)) return fail("dense: K must be a multiple of 128, A and W 16-byte aligned");
  launch_dense_nt(a, (hipStream_t)stream);
  return check_launch("dense");
}

extern "C" int mn_op_dense_wgrad(const float* dY, const float* X, float* dW, float* db, int B, int F, int Cin, float alpha,
                                 void* stream) {
  begin_call();
  if (B <= 0 || F <= 0 || Cin <= 0) return fail("dense_wgrad: empty problem");
  DenseWgradArgs a;
  a.dY = dY; a.X = X; a.dW = dW; a.db = db; a.B = B; a.F = F; a.Cin = Cin; a.ldy = F; a.ldx = Cin; a.ldw = Cin; a.alpha = alpha;
  launch_dense_wgrad(a, (hipStream_t)stream);
  return check_launch("dense_wgrad");
}

extern "C" int mn_op_head_wgrad(const float* dposes, const float* feat, float* dWx, float* dbx, float* dWq, float* dbq, int B, int K,
                                float scale, int filter_nans, void* stream) {
  begin_call();
  if (B <= 0 || K <= 0) return fail("head_wgrad: empty problem");
  hipLaunchKernelGGL(head_bwd_weight_kernel, dim3(cdiv(K + 1, 64)), dim3(256), 0, (hipStream_t)stream, dposes, feat, dWx, dbx, dWq,
                     dbq, B, K, scale, filter_nans);
  return check_launch("head_wgrad");
}

extern "C" int mn_op_stem_bwd(const void* y, const unsigned char* idx, const void* gp, const float* gamma, const float* beta,
                              const float* mean, const float* invstd, const void* xpad, float* dW, int ldw, const int32_t* colmap,
                              float* dgamma, float* dbeta, float* coef_scratch, double* accum_scratch, int B, int H, int W, int Wp,
                              float alpha, void* stream) {
  begin_call();
  if (Wp % 2 != 0 || Wp < W + 7) return fail("stem_bwd: Wp must be even and >= W + 7");
  hipStream_t s = (hipStream_t)stream;
  const int H0 = (H - 1) / 2 + 1, W0 = (W - 1) / 2 + 1;
  StemBwdArgs a;
  a.y = (const half*)y; a.idx = idx; a.gp = (const half*)gp; a.gamma = gamma; a.beta = beta; a.coef = coef_scratch; a.mean = mean;
  a.invstd = invstd; a.accum = accum_scratch; a.accum_rows = 1; a.xpad = (const half*)xpad; a.dW = dW; a.colmap = colmap;
  a.ldw = ldw; a.alpha = alpha;
  hipMemsetAsync(accum_scratch, 0, 2 * 64 * sizeof(double), s);
  launch_stem_bn_reduce(a, B, H, W, Wp, s);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(64 / kBnFinalizeChannels), dim3(256), 0, s, (const double*)accum_scratch, (double)((long)B * H0 * W0), gamma,
                     mean, invstd, dgamma, dbeta, alpha, beta, coef_scratch, 64, 1);
  launch_stem_wgrad(a, B, H, W, Wp, s);
  return check_launch("stem_bwd");
}

extern "C" int mn_op_oihw_to_ohwi(const float* src, float* dst, int O, int I, int H, int W, int to_ohwi, void* stream) {
  begin_call();
  hipLaunchKernelGGL(oihw_ohwi_kernel, dim3(ew_grid((long)O * I * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, O,
                     I, H, W, to_ohwi);
  return check_launch("oihw_ohwi");
}

extern "C" int mn_op_criterion(int mode, int N, int T, const float* pred, const float* targ, const float* s, float* loss,
                               float* dpred, float* ds, float* vos_out, float grad_scale, void* stream) {
  begin_call();
  if (T < 1 || T > kMaxT) return fail("criterion: T out of range");
  if (mode < 0 || mode > 3) return fail("criterion: bad mode");
  CriterionArgs a;
  a.mode = mode; a.N = N; a.T = T; a.pred = pred; a.targ = targ; a.s = s; a.loss = loss; a.dpred = dpred; a.ds = ds;
  a.vos_out = vos_out; a.grad_scale = grad_scale;
  hipLaunchKernelGGL(criterion_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("criterion");
}

extern "C" int mn_op_calc_vos(const float* poses, int N, int T, float* vos, const float* cot, float* dposes,
                              void* stream) {
  begin_call();
  if (T < 2 || T > kMaxT) return fail("calc_vos: T out of range");
  hipLaunchKernelGGL(calc_vos_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, poses, N, T, vos, cot,
                     dposes);
  return check_launch("calc_vos");
}

extern "C" int mn_op_conv_dgrad(int dtype, int B, int Hin, int Win, int Cin, int Cout, int k, int stride, int pad,
                               const void* gy, const void* wd, void* gx, const void* res, const void* res_gate,
                               const void* out_gate, int parity, const void* zero_page, void* stream) {
  begin_call();
  if (stride != 1 && stride != 2) return fail("conv_dgrad: stride must be 1 or 2");
  if (k < 1 || k > 5 || pad < 0 || pad >= k) return fail("conv_dgrad: kernel / padding out of range");
  if (!zero_page) return fail("conv_dgrad: zero_page (>= 16 zero bytes of device memory) is required");
  const int Hout = (Hin + 2 * pad - k) / stride + 1, Wout = (Win + 2 * pad - k) / stride + 1;
  const int vec = dtype == MN_F16 ? 8 : 4;
  if (dtype == MN_DTYPE_F16X2 && (Cin % 32 != 0 || Cout % 32 != 0)) return fail("conv_dgrad: h2 operands need channel counts % 32 == 0");
  DgradGeom d = make_dgrad_geom(B, Hin, Win, Cin, Cout, k, stride, pad, Hout, Wout, vec);
  if (int e = check_geom(d.full, dtype)) return e;
  if (Cin % vec != 0) return fail("conv_dgrad: Cin must be a multiple of the 16-byte piece");
  Epilogue ep;
  ep.out = gx; ep.ldc = Cin; ep.stats = nullptr; ep.bias = nullptr; ep.relu = 0; ep.res = res; ep.res_gate = res_gate;
  ep.out_gate = out_gate; ep.alpha = 1.f;
  if (dtype == MN_DTYPE_F32X3) {
    d.full.mma = MMA_BF16X3;
    for (int i = 0; i < d.n_pc; ++i) d.pc[i].g.mma = MMA_BF16X3;
  }
  if (dtype == MN_DTYPE_F16X2)  // gy, wd, res_gate, out_gate h2; gx, res fp32
    launch_conv_dgrad<half>(d, (const half*)gy, (const half*)wd, ep, (hipStream_t)stream, (const half*)zero_page, parity != 0, true);
  else if (dtype == MN_F16)
    launch_conv_dgrad<half>(d, (const half*)gy, (const half*)wd, ep, (hipStream_t)stream, (const half*)zero_page, parity != 0);
  else
    launch_conv_dgrad<float>(d, (const float*)gy, (const float*)wd, ep, (hipStream_t)stream, (const float*)zero_page,
                             parity != 0);
  return check_launch("conv_dgrad");
}

extern "C" int mn_pgo_optimize(const double* poses, const double* vos, double* out, int32_t* status, int W, int N,
                               int fc_vos, double sax, double saq, double srx, double srq, int n_iters, void* stream) {
  begin_call();
  if (W <= 0) return fail("pgo: no windows");
  if (N < 2 || N > kPgoMaxN) return fail("pgo: 2 <= poses per window <= 12");
  if (!(sax > 0.0 && saq > 0.0 && srx > 0.0 && srq > 0.0)) return fail("pgo: covariances must be positive");
  if (n_iters < 0) return fail("pgo: n_iters < 0");
  if (!poses || !vos || !out || !status) return fail("pgo: null pointer");
  PgoArgs a;
  a.poses = poses; a.vos = vos; a.out = out; a.status = status; a.W = W; a.N = N; a.fc = fc_vos ? 1 : 0; a.n_iters = n_iters;
  a.w_ax = sqrt(1.0 / sax); a.w_aq = sqrt(1.0 / saq); a.w_rx = sqrt(1.0 / srx); a.w_rq = sqrt(1.0 / srq);
  hipLaunchKernelGGL(pgo_kernel, dim3(W), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("pgo");
}

extern "C" int mn_op_optim(int method, int nesterov, float* p, const float* g, float* m, float* v, int64_t n, int64_t n_clip,
                           float lr, float wd, float beta1, float beta2, float eps, int64_t step, float grad_mul, float max_norm,
                           double* sqnorm_scratch, int eps_mode, void* stream);
extern "C" int mn_op_adam(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_clip, float lr, float wd,
                          float beta1, float beta2, float eps, int64_t step, float grad_mul, float max_norm,
                          double* sqnorm_scratch, int eps_mode, void* stream) {
  return mn_op_optim(0, 0, p, g, m, v, n, n_clip, lr, wd, beta1, beta2, eps, step, grad_mul, max_norm, sqnorm_scratch, eps_mode,
                     stream);
}
extern "C" int mn_op_optim(int method, int nesterov, float* p, const float* g, float* m, float* v, int64_t n, int64_t n_clip,
                           float lr, float wd, float beta1, float beta2, float eps, int64_t step, float grad_mul, float max_norm,
                           double* sqnorm_scratch, int eps_mode, void* stream) {
  begin_call();
  if (method < 0 || method > 2) return fail("optim: method must be 0 (adam), 1 (sgd) or 2 (rmsprop)");
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.f) {
    if (!sqnorm_scratch) return fail("adam: clipping needs a scratch double");
    hipMemsetAsync(sqnorm_scratch, 0, sizeof(double), s);
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(sqnorm_grid(n_clip)), dim3(256), 0, s, g, (long)n_clip, sqnorm_scratch, (double*)nullptr);
  }
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.n_clip = n_clip; a.lr = lr; a.wd = wd; a.beta1 = beta1; a.beta2 = beta2;
  a.eps = eps; a.bc1 = (float)(1.0 - pow((double)beta1, (double)step)); a.bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  a.grad_mul = grad_mul; a.max_norm = max_norm; a.sqnorm = sqnorm_scratch; a.frozen = nullptr; a.eps_mode = eps_mode;
  a.bc_dev = nullptr;
  a.method = method; a.nesterov = nesterov ? 1 : 0; a.first_step = step <= 1 ? 1 : 0;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, s, a);
  return check_launch("optim");
}

template <typename T>
static int bn_train_fwd_t(const void* y, int64_t M, int C, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float* mean, float* invstd, const void* res, int relu, void* out, float eps,
                          float momentum, double* accum, hipStream_t s) {
  // standalone operator: statistics by a direct reduction (the network path takes them from the
  // conv epilogue instead).  accum: [2][C] doubles + [2][C] floats of scale/shift after it.
  hipMemsetAsync(accum, 0, 2 * C * sizeof(double), s);
  int rows_per_block = 256;
  hipLaunchKernelGGL((bn_fwd_stats_kernel<T>), dim3(cdiv(M, rows_per_block)), dim3(256), 0, s, (const T*)y, (long)M, C,
                     accum, rows_per_block);
  BnParams p;
  p.gamma = gamma; p.beta = beta; p.running_mean = running_mean; p.running_var = running_var;
  p.num_batches_tracked = nullptr; p.mean = mean; p.invstd = invstd; p.eps = eps; p.momentum = momentum;
  long np = M * C / ElemTraits<T>::VEC;
  float* coef = reinterpret_cast<float*>(accum + 2 * C);  // [2][C] floats behind the accumulators
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, kBnFinalizeChannels)), dim3(256), 0, s, (const double*)accum, (double)M, p, 1, coef, C, 1);
  hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(ew_grid(np)), dim3(256), 0, s, (const T*)y, (const float*)coef, (const T*)res,
                     (T*)out, np, C, relu);
  return check_launch("bn_train_fwd");
}

// the apply kernels keep a thread's per-channel coefficients in registers: its channel piece must be the same in every
// grid-stride iteration, i.e. C / VEC pieces per row must be a power of two that divides the 256-thread workgroup
static int check_bn_channels(int dtype, int C) {
  const int vec = dtype == MN_F16 ? 8 : 4;
  const int cpr = C / vec;
  if (C <= 0 || C % vec != 0 || C > 512 || cpr > 256 || (cpr & (cpr - 1)) != 0)
    return fail("bn: C must be a power of two times the 16-byte piece (8 halves / 4 floats), at most 512");
  return 0;
}

extern "C" int mn_op_bn_train_fwd(int dtype, const void* y, int64_t M, int C, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float* mean, float* invstd, const void* res,
                                  int relu, void* out, float eps, float momentum, double* accum_scratch, void* stream) {
  begin_call();
  if (int e = check_bn_channels(dtype, C)) return e;
  if (dtype == MN_F16)
    return bn_train_fwd_t<half>(y, M, C, gamma, beta, running_mean, running_var, mean, invstd, res, relu, out, eps,
                                momentum, accum_scratch, (hipStream_t)stream);
  return bn_train_fwd_t<float>(y, M, C, gamma, beta, running_mean, running_var, mean, invstd, res, relu, out, eps,
                               momentum, accum_scratch, (hipStream_t)stream);
}

template <typename T>
static int bn_bwd_t(const void* g, const void* gate, const void* y, int64_t M, int C, const float* gamma, const float* mean,
                    const float* invstd, float* dgamma, float* dbeta, void* gy, float* coef, double* accum,
                    float grad_unscale, hipStream_t s) {
  if (!coef) return fail("bn_bwd: coef_scratch (3*C floats) is required");
  launch_bn_bwd<T>((const T*)g, (const T*)gate, (const T*)y, (long)M, C, gamma, mean, invstd, dgamma, dbeta, (T*)gy, accum,
                   coef, grad_unscale, s);
  hipMemsetAsync(accum, 0, 2 * C * sizeof(double), s);  // hand the scratch back zeroed
  return check_launch("bn_bwd");
}

extern "C" int mn_op_bn_bwd(int dtype, const void* g, const void* gate, const void* y, int64_t M, int C, const float* gamma,
                            const float* mean, const float* invstd, float* dgamma, float* dbeta, void* gy,
                            float* coef_scratch, double* accum_scratch, float grad_unscale, void* stream) {
  begin_call();
  if (int e = check_bn_channels(dtype, C)) return e;
  hipMemsetAsync(accum_scratch, 0, 2 * C * sizeof(double), (hipStream_t)stream);
  if (dtype == MN_F16)
    return bn_bwd_t<half>(g, gate, y, M, C, gamma, mean, invstd, dgamma, dbeta, gy, coef_scratch, accum_scratch,
                          grad_unscale, (hipStream_t)stream);
  return bn_bwd_t<float>(g, gate, y, M, C, gamma, mean, invstd, dgamma, dbeta, gy, coef_scratch, accum_scratch,
                         grad_unscale, (hipStream_t)stream);
}

extern "C" int mn_op_maxpool_fwd(int dtype, const void* in, void* out, unsigned char* idx, int B, int H, int W, int C,
                                 void* stream) {
  begin_call();
  int Po = (H + 2 - 3) / 2 + 1, Qo = (W + 2 - 3) / 2 + 1;
  if (dtype == MN_F16)
    hipLaunchKernelGGL((maxpool_fwd_kernel<half>), dim3(ew_grid((long)B * Po * Qo * C / 8)), dim3(256), 0,
                       (hipStream_t)stream, (const half*)in, (half*)out, idx, B, H, W, C, Po, Qo);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(ew_grid((long)B * Po * Qo * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)in, (float*)out, idx, B, H, W, C, Po, Qo);
  return check_launch("maxpool_fwd");
}

extern "C" int mn_op_maxpool_bwd(int dtype, const unsigned char* idx, const void* gout, void* gin, int B, int H, int W,
                                 int C, void* stream) {
  begin_call();
  int Po = (H + 2 - 3) / 2 + 1, Qo = (W + 2 - 3) / 2 + 1;
  if (dtype == MN_F16)
    hipLaunchKernelGGL((maxpool_bwd_kernel<half>), dim3(ew_grid((long)B * H * W * C / 8)), dim3(256), 0,
                       (hipStream_t)stream, idx, (const half*)gout, (half*)gin, B, H, W, C, Po, Qo);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(ew_grid((long)B * H * W * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, idx, (const float*)gout, (float*)gin, B, H, W, C, Po, Qo);
  return check_launch("maxpool_bwd");
}

extern "C" int mn_op_occupy(int workgroups, int threads, float microseconds, const void* src, void* dst, int64_t bytes, int lds_kb,
                            void* stream) {
  begin_call();
  if (workgroups < 1 || workgroups > 4096 || threads < 64 || threads > 1024 || threads % 64 != 0)
    return fail("mn_op_occupy: 1..4096 workgroups of 64..1024 threads (a multiple of 64)");
  if (!(microseconds >= 0.f) || microseconds > 1e6f) return fail("mn_op_occupy: 0 <= microseconds <= 1e6");
  if (bytes < 0 || (bytes > 0 && (!src || !dst))) return fail("mn_op_occupy: bytes > 0 needs src and dst");
  if (lds_kb != 0 && lds_kb != 32 && lds_kb != 64) return fail("mn_op_occupy: lds_kb must be 0, 32 or 64");
  launch_occupy(workgroups, threads, microseconds, src, dst, (long)bytes, lds_kb, (hipStream_t)stream);
  return check_launch("occupy");
}
