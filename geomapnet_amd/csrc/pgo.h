// Pose-graph optimisation of sliding windows, batched: one wavefront per window, fp64, everything in LDS.
//
// Replaces PoseGraph.optimize / PoseGraphFC.optimize (common/pose_utils.py:458-771), which scripts/eval.py:177-182
// calls once per window on the host (numpy + scipy.linalg).  Here every window of the evaluation set is one
// workgroup of a single launch.
//
// Per iteration (n_iters = 10 in the reference):
//   H = J'J and b = J'r are assembled block-wise without materialising J: constraints only couple 3-column blocks
//     (translation block 2i / rotation block 2i+1 of pose i), so a thread owns whole 6x6 pose-pair blocks of H
//     (and, on the diagonal, the 6 entries of b): no atomics, no races.
//       unary translation i  (:474-477, :527-529)   J = w_ax I3                 r = w_ax (t_i - t_i^0)
//       unary rotation    i  (:480-482)             J = w_aq M(q_i)  (4x3)      r = w_aq (q_i - q_i^0)
//       pair translation i,j (:487-496, :533-541)   J_i = -w_rx Dt(q_i), J_j = +w_rx Dt(q_i);   r = w_rx (rot(t_j - t_i, q_i^-1) - vo_t)
//       pair rotation    i,j (:498-506, :543-549)   J_i = w_rq dpsq_p(q_j) M(q_i), J_j = w_rq dpsq_q(q_i) M(q_j);  r = w_rq (q_i^-1 q_j - vo_q)
//     with M(q) = columns 1..3 of dpq_q(q) (:445-456), Dt = dqstq_t (:435-443).  The information factors are
//     cholesky(I / s).T = I / sqrt(s) (:596-599): scalars.
//   R = chol(H) (upper, H = R'R) in place; then the reference's two solves EXACTLY as written (:605-607):
//     y = solve_triangular(R.T, -b) with scipy's default lower=False reads only the upper triangle of R.T, i.e. its
//     diagonal, so y = -b / diag(R); x = solve_triangular(R, y) is the back substitution.  (Not the Gauss-Newton
//     step; it is what the reference computes and what its results come from.)
//   update on the manifold (:555-577): t += x_t; q <- q (x) [cos n, sinc(n/pi) x_q], no renormalisation.
// A non-positive pivot (scipy raises LinAlgError) sets status[window] = 1 and leaves the window's output as the
// state reached so far.
#pragma once
#include "common.h"

namespace mn {

constexpr int kPgoMaxN = 12;                               // poses per window (reference: steps = 7)
constexpr int kPgoMaxD = 6 * kPgoMaxN;                     // 72 manifold dimensions
constexpr int kPgoMaxP = kPgoMaxN * (kPgoMaxN - 1) / 2;    // fully connected pairs

struct PgoArgs {
  const double* poses;  // [W][N][7]  (t, q = [w x y z])
  const double* vos;    // [W][P][7]  P = N-1 (chain) or N(N-1)/2 (fully connected, pairs i<j in lexicographic order)
  double* out;          // [W][N][7]
  int* status;          // [W]
  int W, N, fc, n_iters;
  double w_ax, w_aq, w_rx, w_rq;  // 1/sqrt(sigma)
};

// transforms3d.quaternions.qmult (Hamilton product), operation order as published
__device__ __forceinline__ void pgo_qmult(const double* a, const double* b, double* o) {
  const double w1 = a[0], x1 = a[1], y1 = a[2], z1 = a[3];
  const double w2 = b[0], x2 = b[1], y2 = b[2], z2 = b[3];
  o[0] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
  o[1] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
  o[2] = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2;
  o[3] = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2;
}
// qinverse = conjugate / dot(q, q)
__device__ __forceinline__ void pgo_qinverse(const double* q, double* o) {
  const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  o[0] = q[0] / d;
  o[1] = -q[1] / d;
  o[2] = -q[2] / d;
  o[3] = -q[3] / d;
}
// M(q) = dpq_q(q)[:, 1:]   4x3 row-major
__device__ __forceinline__ void pgo_mrot(const double* p, double* M) {
  M[0] = -p[1]; M[1] = -p[2]; M[2] = -p[3];
  // rows 1..3: p0 I + skew(pv)
  M[3] = p[0];  M[4] = -p[3]; M[5] = p[2];
  M[6] = p[3];  M[7] = p[0];  M[8] = -p[1];
  M[9] = -p[2]; M[10] = p[1]; M[11] = p[0];
}
// dqstq_t(q): (q0^2 - qv.qv) I + 2 qv qv' - 2 q0 skew(qv)   3x3 row-major
__device__ __forceinline__ void pgo_dt(const double* q, double* D) {
  const double s = q[0] * q[0] - (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double* v = q + 1;
  const double sk[9] = {0.0, -v[2], v[1], v[2], 0.0, -v[0], -v[1], v[0], 0.0};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) D[a * 3 + c] = (a == c ? s : 0.0) + 2.0 * (v[a] * v[c]) - 2.0 * q[0] * sk[a * 3 + c];
}
// A = dpsq_p(qj) M(qi), B = dpsq_q(qi) M(qj)     4x3 row-major each
__device__ __forceinline__ void pgo_pair_q_jac(const double* qi, const double* qj, double* A, double* B) {
  double Mi[12], Mj[12];
  pgo_mrot(qi, Mi);
  pgo_mrot(qj, Mj);
  // dpsq_p(q): [[q0, qv'], [qv, -q0 I + skew(qv)]]
  const double P[16] = {qj[0], qj[1],  qj[2],  qj[3],
                        qj[1], -qj[0], -qj[3], qj[2],
                        qj[2], qj[3],  -qj[0], -qj[1],
                        qj[3], -qj[2], qj[1],  -qj[0]};
  // dpsq_q(p): [[p0, -pv'], [-pv, p0 I - skew(pv)]]
  const double Q[16] = {qi[0],  -qi[1], -qi[2], -qi[3],
                        -qi[1], qi[0],  qi[3],  -qi[2],
                        -qi[2], -qi[3], qi[0],  qi[1],
                        -qi[3], qi[2],  -qi[1], qi[0]};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a += P[r * 4 + k] * Mi[k * 3 + c];
        b += Q[r * 4 + k] * Mj[k * 3 + c];
      }
      A[r * 3 + c] = a;
      B[r * 3 + c] = b;
    }
}
// residuals of pair (i, j) against vo (7 values), unweighted
__device__ __forceinline__ void pgo_pair_res(const double* zi, const double* zj, const double* vo, double* rt, double* rq) {
  double qin[4], qc[4], varr[4], tmp[4], rot[4];
  pgo_qinverse(zi + 3, qin);
  qc[0] = qin[0]; qc[1] = -qin[1]; qc[2] = -qin[2]; qc[3] = -qin[3];
  varr[0] = 0.0; varr[1] = zj[0] - zi[0]; varr[2] = zj[1] - zi[1]; varr[3] = zj[2] - zi[2];
  pgo_qmult(varr, qc, tmp);
  pgo_qmult(qin, tmp, rot);  // rotate_vector(v, q) = (q (0, v) q*)[1:]
  rt[0] = rot[1] - vo[0]; rt[1] = rot[2] - vo[1]; rt[2] = rot[3] - vo[2];
  pgo_qmult(qin, zj + 3, tmp);
  rq[0] = tmp[0] - vo[3]; rq[1] = tmp[1] - vo[4]; rq[2] = tmp[2] - vo[5]; rq[3] = tmp[3] - vo[6];
}
__device__ __forceinline__ int pgo_pair_index(int i, int j, int N, int fc) {  // i < j
  return fc ? i * N - (i * (i + 1)) / 2 + (j - i - 1) : i;
}

// C[3x3] += s * X' Y for X, Y with `rows` rows of 3
__device__ __forceinline__ void pgo_xty(const double* X, const double* Y, int rows, double s, double* C) {
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double acc = 0.0;
      for (int r = 0; r < rows; ++r) acc += X[r * 3 + a] * Y[r * 3 + c];
      C[a * 3 + c] += s * acc;
    }
}

static __global__ void __launch_bounds__(64) pgo_kernel(PgoArgs a) {
  __shared__ double z[7 * kPgoMaxN], z0[7 * kPgoMaxN], vo[7 * kPgoMaxP];
  __shared__ double H[kPgoMaxD * kPgoMaxD], bv[kPgoMaxD], xv[kPgoMaxD];
  __shared__ int bad;
  const int w = blockIdx.x, t = threadIdx.x;
  const int N = a.N, D = 6 * N, fc = a.fc;
  const int P = fc ? N * (N - 1) / 2 : N - 1;
  for (int i = t; i < 7 * N; i += 64) z[i] = z0[i] = a.poses[(long)w * 7 * N + i];
  for (int i = t; i < 7 * P; i += 64) vo[i] = a.vos[(long)w * 7 * P + i];
  if (t == 0) bad = 0;
  __syncthreads();
  const double ax2 = a.w_ax * a.w_ax, aq2 = a.w_aq * a.w_aq, rx2 = a.w_rx * a.w_rx, rq2 = a.w_rq * a.w_rq;

  for (int it = 0; it < a.n_iters; ++it) {
    // ---- H = J'J, b = J'r: one 6x6 pose-pair block per thread-iteration ----------------------------
    for (int blk = t; blk < N * N; blk += 64) {
      const int i = blk / N, j = blk - i * N;
      double Ct[9], Cq[9];  // translation-translation and rotation-rotation 3x3 sub-blocks (t-q coupling is zero)
#pragma unroll
      for (int e = 0; e < 9; ++e) Ct[e] = Cq[e] = 0.0;
      if (i == j) {
        double gt[3] = {0.0, 0.0, 0.0}, gq[3] = {0.0, 0.0, 0.0};  // b entries of pose i
        const double* zi = z + 7 * i;
        // unary
        Ct[0] += ax2; Ct[4] += ax2; Ct[8] += ax2;
#pragma unroll
        for (int c = 0; c < 3; ++c) gt[c] += ax2 * (zi[c] - z0[7 * i + c]);
        double Mi[12];
        pgo_mrot(zi + 3, Mi);
        pgo_xty(Mi, Mi, 4, aq2, Cq);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc += Mi[r * 3 + c] * (zi[3 + r] - z0[7 * i + 3 + r]);
          gq[c] += aq2 * acc;
        }
        // pairs in which pose i takes part
        for (int o = 0; o < N; ++o) {
          if (o == i || (!fc && (o - i > 1 || i - o > 1))) continue;
          const int lo = o < i ? o : i, hi = o < i ? i : o;
          const double *zl = z + 7 * lo, *zh = z + 7 * hi;
          double Dt[9], A[12], B[12], rt[3], rq[4];
          pgo_dt(zl + 3, Dt);
          pgo_pair_q_jac(zl + 3, zh + 3, A, B);
          pgo_pair_res(zl, zh, vo + 7 * pgo_pair_index(lo, hi, N, fc), rt, rq);
          pgo_xty(Dt, Dt, 3, rx2, Ct);
          const double sgn = (i == lo) ? -1.0 : 1.0;  // J_lo = -w Dt, J_hi = +w Dt
          const double* Jq = (i == lo) ? A : B;
          pgo_xty(Jq, Jq, 4, rq2, Cq);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            double acc = 0.0, accq = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) acc += Dt[r * 3 + c] * rt[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) accq += Jq[r * 3 + c] * rq[r];
            gt[c] += sgn * rx2 * acc;
            gq[c] += rq2 * accq;
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          bv[6 * i + c] = gt[c];
          bv[6 * i + 3 + c] = gq[c];
        }
      } else if (fc || j - i == 1 || i - j == 1) {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const double *zl = z + 7 * lo, *zh = z + 7 * hi;
        double Dt[9], A[12], B[12];
        pgo_dt(zl + 3, Dt);
        pgo_pair_q_jac(zl + 3, zh + 3, A, B);
        pgo_xty(Dt, Dt, 3, -rx2, Ct);  // (-w Dt)'(+w Dt): symmetric in (lo, hi)
        if (i == lo)
          pgo_xty(A, B, 4, rq2, Cq);  // rows of pose lo, columns of pose hi: A'B
        else
          pgo_xty(B, A, 4, rq2, Cq);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          H[(6 * i + r) * D + 6 * j + c] = Ct[r * 3 + c];
          H[(6 * i + 3 + r) * D + 6 * j + 3 + c] = Cq[r * 3 + c];
          H[(6 * i + r) * D + 6 * j + 3 + c] = 0.0;
          H[(6 * i + 3 + r) * D + 6 * j + c] = 0.0;
        }
    }
    __syncthreads();

    // ---- Cholesky, lower factor L = R' in place (only the lower triangle is read or written) ----------
    for (int k = 0; k < D; ++k) {
      if (t == 0) {
        const double d = H[k * D + k];
        if (!(d > 0.0)) bad = 1;
        H[k * D + k] = sqrt(d);
      }
      __syncthreads();
      if (bad) break;
      const double piv = H[k * D + k];
      for (int i = k + 1 + t; i < D; i += 64) H[i * D + k] /= piv;
      __syncthreads();
      for (int i = k + 1 + t; i < D; i += 64) {
        const double lik = H[i * D + k];
        for (int j = k + 1; j <= i; ++j) H[i * D + j] -= lik * H[j * D + k];
      }
      __syncthreads();
    }
    if (bad) break;
    // ---- y = -b / diag(R)  (see the header), then R x = y by back substitution; R[i][j] = L[j][i] --------
    for (int i = t; i < D; i += 64) xv[i] = -bv[i] / H[i * D + i];
    __syncthreads();
    for (int i = D - 1; i >= 0; --i) {
      if (t == 0) xv[i] /= H[i * D + i];
      __syncthreads();
      const double xi = xv[i];
      for (int j = t; j < i; j += 64) xv[j] -= H[i * D + j] * xi;
      __syncthreads();
    }
    // ---- update on the manifold -----------------------------------------------------------------------
    for (int i = t; i < N; i += 64) {
      double* zi = z + 7 * i;
      const double* x = xv + 6 * i;
      zi[0] += x[0]; zi[1] += x[1]; zi[2] += x[2];
      const double n = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
      const double xs = n / 3.141592653589793;
      const double y = 3.141592653589793 * (xs == 0.0 ? 1.0e-20 : xs);  // numpy.sinc
      const double sc = sin(y) / y;
      const double dq[4] = {cos(n), sc * x[3], sc * x[4], sc * x[5]};
      double q[4];
      pgo_qmult(zi + 3, dq, q);
      zi[3] = q[0]; zi[4] = q[1]; zi[5] = q[2]; zi[6] = q[3];
    }
    __syncthreads();
  }
  for (int i = t; i < 7 * N; i += 64) a.out[(long)w * 7 * N + i] = z[i];
  if (t == 0) a.status[w] = bad;
}

}  // namespace mn
