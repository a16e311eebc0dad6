// Geometry-aware pose criteria, forward + backward fused (fp32).
//
// Implements the losses of /root/reference/common/criterion.py as one kernel launch each:
//   mode 0  PoseNetCriterion        (:42-52)    pred/targ [N][6]
//   mode 1  MapNetCriterion         (:76-109)   pred/targ [N][T][6], VO = calc_vos_simple
//   mode 2  MapNetOnlineCriterion   (:137-184)  pred [N][2T][6], targ [N][2T-1][6], VO = calc_vos
//   mode 3  MapNetOnlineCriterion, gps_mode     pred/targ [N][2T][6], VO = translation[:2] only
// together with the analytic gradient w.r.t. pred and the four log-weights (sax, saq, srx, srq).
// The relative-pose chain of mode 2 (pose_utils.py: qexp_t :73-84, qinv :64-71, rotate_vec_by_q
// :120-132, qmult+normalize :44-62, qlog_t :86-96, calc_vo_logq :167-179) is evaluated per
// consecutive pair with the reference's operation order, and differentiated by hand with the
// same sub-gradient conventions autograd uses (clamp passes gradient on the closed interval,
// norm has zero gradient at 0, acos' -1/sqrt(1-x^2) is allowed to produce inf/NaN), so the
// NaN hazard of coincident rotations is reproduced, not hidden.
//
// One workgroup of 256 threads; thread n-strides over windows; block reduction of the four
// L1 sums in fp64.  d(pred) is written multiplied by `grad_scale` (the fp16 loss scale).
#pragma once
#include "common.h"

namespace mn {

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) {
  V3 r = {x, y, z};
  return r;
}
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float norm3(V3 a) { return sqrtf(dot(a, a)); }

struct Q4 {
  float s;
  V3 v;
};

// ---- forward pieces ---------------------------------------------------------------------------
__device__ __forceinline__ Q4 qexp_fwd(V3 w, float& n_raw, float& n) {
  n_raw = norm3(w);
  n = fmaxf(n_raw, 1e-8f);
  Q4 q;
  q.s = cosf(n);
  q.v = (w * sinf(n)) / n;
  return q;
}
__device__ __forceinline__ V3 rot_fwd(V3 t, Q4 q) {
  V3 b = cross(q.v, t);
  V3 c = cross(q.v, b) * 2.f;
  V3 b2 = (b * q.s) * 2.f;
  return (t + b2) + c;
}
// cotangent `o` on rot(t, q): accumulates into tb, qb
__device__ __forceinline__ void rot_bwd(V3 t, Q4 q, V3 o, V3& tb, Q4& qb) {
  V3 b = cross(q.v, t);
  tb = tb + o;
  qb.s += 2.f * dot(o, b);
  V3 bb = o * (2.f * q.s);
  // c = 2 (qv x b)
  qb.v = qb.v + cross(b, o) * 2.f;
  bb = bb + cross(o, q.v) * 2.f;
  // b = qv x t
  qb.v = qb.v + cross(t, bb);
  tb = tb + cross(bb, q.v);
}
__device__ __forceinline__ void qexp_bwd(V3 w, float n_raw, float n, Q4 qb, V3& wb) {
  float sn = sinf(n), cs = cosf(n);
  V3 u = w * sn;
  V3 ub = qb.v / n;
  float nb = -dot(qb.v, u) / (n * n);
  wb = wb + ub * sn;
  float snb = dot(ub, w);
  nb += snb * cs;
  nb += qb.s * (-sn);
  float nrb = (n_raw >= 1e-8f) ? nb : 0.f;
  if (n_raw > 0.f) wb = wb + w * (nrb / n_raw);
}

// relative pose of p1 in the frame of p0 (log-quaternion in/out) and its VJP.
// p0, p1: [6] = (t, w).  u: [6] out.  With cotangent ub[6]: accumulates into g0[6], g1[6].
__device__ __forceinline__ void vo_logq(const float* p0, const float* p1, float* u, const float* ub, float* g0,
                                        float* g1, bool want_grad) {
  V3 t0 = v3(p0[0], p0[1], p0[2]), w0 = v3(p0[3], p0[4], p0[5]);
  V3 t1 = v3(p1[0], p1[1], p1[2]), w1 = v3(p1[3], p1[4], p1[5]);
  float n0r, n0, n1r, n1;
  Q4 q0 = qexp_fwd(w0, n0r, n0), q1 = qexp_fwd(w1, n1r, n1);
  Q4 qi;
  qi.s = q0.s;
  qi.v = neg(q0.v);
  V3 tinv = neg(rot_fwd(t0, qi));
  // qmult(qi, q1)
  Q4 q;
  q.s = qi.s * q1.s - dot(qi.v, q1.v);
  q.v = (qi.v * q1.s + q1.v * qi.s) + cross(qi.v, q1.v);
  float nq = sqrtf((q.s * q.s + q.v.x * q.v.x) + (q.v.y * q.v.y + q.v.z * q.v.z));
  Q4 qn;
  qn.s = q.s / nq;
  qn.v = q.v / nq;
  V3 t = tinv + rot_fwd(t1, qi);
  // qlog
  float nv = norm3(qn.v);
  float n2 = fmaxf(nv, 1e-8f);
  float xc = fminf(fmaxf(qn.s, -1.f), 1.f);
  float ac = acosf(xc);
  V3 lq = (qn.v * ac) / n2;
  u[0] = t.x;
  u[1] = t.y;
  u[2] = t.z;
  u[3] = lq.x;
  u[4] = lq.y;
  u[5] = lq.z;
  if (!want_grad) return;

  V3 tb_out = v3(ub[0], ub[1], ub[2]), lb = v3(ub[3], ub[4], ub[5]);
  // qlog backward
  Q4 qnb;
  qnb.v = lb * (ac / n2);
  float acb = dot(lb, qn.v) / n2;
  float n2b = -dot(lb, qn.v * ac) / (n2 * n2);
  float nvb = (nv >= 1e-8f) ? n2b : 0.f;
  if (nv > 0.f) qnb.v = qnb.v + qn.v * (nvb / nv);
  float gx = acb * (-1.f / sqrtf(1.f - xc * xc));
  float mask = (qn.s >= -1.f && qn.s <= 1.f) ? 1.f : 0.f;
  qnb.s = gx * mask;
  // normalise backward
  Q4 qb;
  qb.s = qnb.s / nq;
  qb.v = qnb.v / nq;
  float nqb = -((qnb.s * q.s + qnb.v.x * q.v.x) + (qnb.v.y * q.v.y + qnb.v.z * q.v.z)) / (nq * nq);
  if (nq > 0.f) {
    qb.s += nqb * q.s / nq;
    qb.v = qb.v + q.v * (nqb / nq);
  }
  // qmult backward (a = qi, b = q1)
  Q4 qib, q1b;
  qib.s = qb.s * q1.s + dot(qb.v, q1.v);
  q1b.s = qb.s * qi.s + dot(qb.v, qi.v);
  qib.v = (qb.v * q1.s - q1.v * qb.s) + cross(q1.v, qb.v);
  q1b.v = (qb.v * qi.s - qi.v * qb.s) + cross(qb.v, qi.v);
  // t = tinv + rot(t1, qi);  tinv = -rot(t0, qi)
  V3 t1b = v3(0, 0, 0), t0b = v3(0, 0, 0);
  rot_bwd(t1, qi, tb_out, t1b, qib);
  rot_bwd(t0, qi, neg(tb_out), t0b, qib);
  // qi = conj(q0)
  Q4 q0b;
  q0b.s = qib.s;
  q0b.v = neg(qib.v);
  V3 w0b = v3(0, 0, 0), w1b = v3(0, 0, 0);
  qexp_bwd(w0, n0r, n0, q0b, w0b);
  qexp_bwd(w1, n1r, n1, q1b, w1b);
  g0[0] += t0b.x;
  g0[1] += t0b.y;
  g0[2] += t0b.z;
  g0[3] += w0b.x;
  g0[4] += w0b.y;
  g0[5] += w0b.z;
  g1[0] += t1b.x;
  g1[1] += t1b.y;
  g1[2] += t1b.z;
  g1[3] += w1b.x;
  g1[4] += w1b.y;
  g1[5] += w1b.z;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

struct CriterionArgs {
  int mode;            // 0 posenet, 1 mapnet, 2 online, 3 online-gps
  int N, T;            // windows, frames per (absolute) window
  const float* pred;   // mode 0: [N][6]; 1: [N][T][6]; 2,3: [N][2T][6]
  const float* targ;   // mode 0: [N][6]; 1: [N][T][6]; 2: [N][2T-1][6]; 3: [N][2T][6]
  const float* s;      // [4] sax, saq, srx, srq
  float* loss;         // [1]
  float* dpred;        // like pred, or null (forward only)
  float* ds;           // [4] accumulated (+=) d loss / d s, or null
  float* vos_out;      // optional: predicted relative poses [N][T-1][6] (modes 1,2), for tests
  float grad_scale;
};

constexpr int kMaxT = 8;

static __global__ void __launch_bounds__(256) criterion_kernel(CriterionArgs a) {
  __shared__ double red[4][4];
  const int T = a.T, N = a.N;
  const int Tp = (a.mode >= 2) ? 2 * T : (a.mode == 0 ? 1 : T);       // pred rows per window
  const int Tg = (a.mode == 2) ? 2 * T - 1 : Tp;                      // targ rows per window
  const bool want_grad = a.dpred != nullptr;
  const float sax = a.s[0], saq = a.s[1], srx = a.s[2], srq = a.s[3];
  const float n_abs = 3.f * (float)N * (float)((a.mode == 0) ? 1 : T);
  float n_vo_t, n_vo_q;
  if (a.mode == 3) {
    n_vo_t = 2.f * (float)N * (float)T;
    n_vo_q = 1.f;
  } else {
    n_vo_t = n_vo_q = 3.f * (float)N * (float)(T - 1);
  }
  const float w_at = expf(-sax) / n_abs, w_aq = expf(-saq) / n_abs;
  const float w_vt = expf(-srx) / n_vo_t, w_vq = expf(-srq) / n_vo_q;
  double sum[4] = {0, 0, 0, 0};  // At, Aq, Vt, Vq

  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const float* p = a.pred + (long)n * Tp * 6;
    const float* g = a.targ + (long)n * Tg * 6;
    float grad[2 * kMaxT][6];
    for (int i = 0; i < Tp; ++i)
      for (int c = 0; c < 6; ++c) grad[i][c] = 0.f;
    const int Tabs = (a.mode == 0) ? 1 : T;
    for (int i = 0; i < Tabs; ++i)
      for (int c = 0; c < 6; ++c) {
        float d = p[i * 6 + c] - g[i * 6 + c];
        sum[c < 3 ? 0 : 1] += fabsf(d);
        grad[i][c] += (c < 3 ? w_at : w_aq) * sgn(d);
      }
    if (a.mode == 1) {
      for (int i = 0; i + 1 < T; ++i)
        for (int c = 0; c < 6; ++c) {
          float pv = p[(i + 1) * 6 + c] - p[i * 6 + c];
          float gv = g[(i + 1) * 6 + c] - g[i * 6 + c];
          float d = pv - gv;
          if (a.vos_out) a.vos_out[((long)n * (T - 1) + i) * 6 + c] = pv;
          sum[c < 3 ? 2 : 3] += fabsf(d);
          float w = (c < 3 ? w_vt : w_vq) * sgn(d);
          grad[i + 1][c] += w;
          grad[i][c] -= w;
        }
    } else if (a.mode == 2) {
      for (int i = 0; i + 1 < T; ++i) {
        float u[6], ub[6];
        vo_logq(p + (T + i) * 6, p + (T + i + 1) * 6, u, nullptr, nullptr, nullptr, false);
        for (int c = 0; c < 6; ++c) {
          float d = u[c] - g[(T + i) * 6 + c];
          if (a.vos_out) a.vos_out[((long)n * (T - 1) + i) * 6 + c] = u[c];
          sum[c < 3 ? 2 : 3] += fabsf(d);
          ub[c] = (c < 3 ? w_vt : w_vq) * sgn(d);
        }
        if (want_grad) vo_logq(p + (T + i) * 6, p + (T + i + 1) * 6, u, ub, grad[T + i], grad[T + i + 1], true);
      }
    } else if (a.mode == 3) {
      for (int i = 0; i < T; ++i)
        for (int c = 0; c < 2; ++c) {
          float d = p[(T + i) * 6 + c] - g[(T + i) * 6 + c];
          sum[2] += fabsf(d);
          grad[T + i][c] += w_vt * sgn(d);
        }
    }
    if (want_grad)
      for (int i = 0; i < Tp; ++i)
        for (int c = 0; c < 6; ++c) a.dpred[((long)n * Tp + i) * 6 + c] = grad[i][c] * a.grad_scale;
  }
  // block reduction
  for (int k = 0; k < 4; ++k) sum[k] = wave_sum_d(sum[k]);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 4; ++k) red[threadIdx.x >> 6][k] = sum[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[4];
    const int nw = (blockDim.x + 63) / 64;
    for (int k = 0; k < 4; ++k) {
      tot[k] = 0;
      for (int w = 0; w < nw; ++w) tot[k] += red[w][k];
    }
    float At = (float)(tot[0] / n_abs), Aq = (float)(tot[1] / n_abs);
    float Vt = (float)(tot[2] / n_vo_t), Vq = (float)(tot[3] / n_vo_q);
    float ea = expf(-sax), eq = expf(-saq), ex = expf(-srx), er = expf(-srq);
    float loss = (ea * At + sax) + (eq * Aq + saq);
    if (a.mode == 1 || a.mode == 2) loss = loss + ((ex * Vt + srx) + (er * Vq + srq));
    if (a.mode == 3) loss = loss + (ex * Vt + srx);
    a.loss[0] = loss;
    if (a.ds) {
      a.ds[0] += 1.f - ea * At;
      a.ds[1] += 1.f - eq * Aq;
      if (a.mode >= 1) a.ds[2] += 1.f - ex * Vt;
      if (a.mode == 1 || a.mode == 2) a.ds[3] += 1.f - er * Vq;
    }
  }
}

// standalone relative-pose op: vos[n][i] = calc_vo_logq(poses[n][i], poses[n][i+1]) (+ VJP)
static __global__ void __launch_bounds__(256) calc_vos_kernel(const float* poses, int N, int T, float* vos, const float* cot,
                                                        float* dposes) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float grad[kMaxT][6];
  for (int i = 0; i < T; ++i)
    for (int c = 0; c < 6; ++c) grad[i][c] = 0.f;
  const float* p = poses + (long)n * T * 6;
  for (int i = 0; i + 1 < T; ++i) {
    float u[6];
    vo_logq(p + i * 6, p + (i + 1) * 6, u, cot ? cot + ((long)n * (T - 1) + i) * 6 : nullptr, grad[i], grad[i + 1],
            cot != nullptr);
    for (int c = 0; c < 6; ++c) vos[((long)n * (T - 1) + i) * 6 + c] = u[c];
  }
  if (dposes)
    for (int i = 0; i < T; ++i)
      for (int c = 0; c < 6; ++c) dposes[((long)n * T + i) * 6 + c] = grad[i][c];
}

}  // namespace mn
