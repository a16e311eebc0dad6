// Data gradient of a convolution as implicit GEMMs (shared by the training plan and the op-level entry point).
//
//   gx[b, h, w, ci] = sum_{r, s, co} gy[b, (h + pad - r) / stride, (w + pad - s) / stride, co] * W[co, r, s, ci]
//
// stride 1: one launch of the transposed gather (rsign = ssign = -1).  stride 2: the generic form (`div = 2`: a tap is
// valid only where both indices are even) wastes three quarters of its MFMA work on structural zeros, so the gradient
// is decomposed by the parity (a, b) of the input pixel: class (a, b) only receives the taps r = r0 + 2r', s = s0 + 2s'
// with r0 = (a + pad) % 2, s0 = (b + pad) % 2 -- a dense stride-1 gather of 1, 2, 2, 4 taps (3x3, pad 1) whose rows are
// stored at the pixels (2p + a, 2q + b) (Epilogue output map) and whose weight walk skips the other taps
// (GatherGeom::bt_*).  A 1x1 stride-2 projection has a single class (the even pixels).
#pragma once
#include "igemm.h"

namespace mn {

struct DgradParityClass {
  GatherGeom g;
  int a, b, P, Q;
};

struct DgradGeom {
  GatherGeom full;         // generic form (any stride)
  DgradParityClass pc[4];  // parity classes (stride 2 only)
  int n_pc = 0;
  int Hin = 0, Win = 0;
};

// vec = elements per 16-byte piece of the operand type
inline DgradGeom make_dgrad_geom(int B, int Hin, int Win, int cin, int cout, int k, int stride, int pad, int Hout, int Wout,
                                 int vec) {
  DgradGeom d;
  d.Hin = Hin;
  d.Win = Win;
  GatherGeom& g = d.full;
  g.B = B; g.Hi = Hout; g.Wi = Wout; g.C = cout; g.P = Hin; g.Q = Win; g.R = k; g.S = k;
  g.mul_p = 1; g.mul_q = 1; g.rsign = -1; g.ssign = -1; g.off_h = pad; g.off_w = pad; g.div = stride;
  g.M = B * Hin * Win; g.N = cin; g.K = k * k * cout;
  if (stride == 2 && (cout / vec) % 4 == 0) {  // the tap-subset weight walk needs the uniform tap walk
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const int r0 = (a + pad) % 2, s0 = (b + pad) % 2;
        const int R = r0 < k ? (k - r0 + 1) / 2 : 0, S = s0 < k ? (k - s0 + 1) / 2 : 0;
        const int P = (Hin - a + 1) / 2, Q = (Win - b + 1) / 2;
        if (R == 0 || S == 0 || P <= 0 || Q <= 0) continue;  // no tap reaches this class: its gradient is zero
        DgradParityClass& c = d.pc[d.n_pc++];
        c.a = a; c.b = b; c.P = P; c.Q = Q;
        GatherGeom& q = c.g;
        q.B = B; q.Hi = Hout; q.Wi = Wout; q.C = cout; q.P = P; q.Q = Q; q.R = R; q.S = S;
        q.mul_p = 1; q.mul_q = 1; q.rsign = -1; q.ssign = -1;
        q.off_h = (a + pad - r0) / 2; q.off_w = (b + pad - s0) / 2; q.div = 1;
        q.M = B * P * Q; q.N = cin; q.K = R * S * cout;
        q.ldb = k * k * cout; q.bt_on = 1; q.bt_r0 = r0; q.bt_s0 = s0; q.bt_S = k;
      }
  }
  return d;
}

// `wd`: weights in the data-gradient layout [Cin][R][S][Cout].  `parity`: use the decomposition when it applies.
// Classes no tap reaches (1x1 stride 2: three of four) have a zero gradient: they are only skipped when the residual
// is accumulated in place (ep.res == ep.out), where they are already right; otherwise the generic form runs.
// `h2` (T = half only): gy / wd are h2 tensors (common.h), the gradient and the residual fp32, the gates h2; the geometry
// counts real channels (make_dgrad_geom with vec = 4).
template <typename T>
inline void launch_conv_dgrad(const DgradGeom& d, const T* gy, const T* wd, const Epilogue& ep, hipStream_t s, const T* zero_page,
                              bool parity, bool h2 = false) {
  if constexpr (std::is_same<T, half>::value) {
    if (h2) {
      if (parity && d.n_pc > 0 && (d.n_pc == 4 || ep.res == ep.out)) {
        for (int i = 0; i < d.n_pc; ++i) {
          const DgradParityClass& c = d.pc[i];
          Epilogue e2 = ep;
          e2.om_on = 1; e2.om_P = c.P; e2.om_Q = c.Q; e2.om_H = d.Hin; e2.om_W = d.Win; e2.om_a = c.a; e2.om_b = c.b;
          e2.om_dq = make_fastdiv(c.Q);
          e2.om_dp = make_fastdiv(c.P);
          launch_igemm_h2(c.g, gy, wd, e2, s, zero_page);
        }
        return;
      }
      launch_igemm_h2(d.full, gy, wd, ep, s, zero_page);
      return;
    }
  }
  if (parity && d.n_pc > 0 && (d.n_pc == 4 || ep.res == ep.out)) {
    for (int i = 0; i < d.n_pc; ++i) {
      const DgradParityClass& c = d.pc[i];
      Epilogue e2 = ep;
      e2.om_on = 1; e2.om_P = c.P; e2.om_Q = c.Q; e2.om_H = d.Hin; e2.om_W = d.Win; e2.om_a = c.a; e2.om_b = c.b;
      e2.om_dq = make_fastdiv(c.Q);
      e2.om_dp = make_fastdiv(c.P);
      launch_igemm<T>(c.g, gy, wd, e2, s, zero_page);
    }
    return;
  }
  launch_igemm<T>(d.full, gy, wd, ep, s, zero_page);
}

}  // namespace mn
