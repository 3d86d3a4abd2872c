// Linearisation of a user-compiled model (user_model.h) by forward-mode duals: device bodies shared by linearize.hip (one
// launch per call) and solve_user.hip (the device-resident loop).
#pragma once
#include "models_device.h"

namespace {

// Lane j of an item owns direction j of z = (x, u).  Column j of [A | B] is the derivative of the WHOLE integrator step along
// e_j (Dual<float> pushed through Euler / the four RK4 stages) — what the reference approximates by a central difference of f
// (quattro_ilqr_tf.py:182-204); row j of the cost Hessian and entry j of its gradient come from n + m evaluations of L on
// Dual<Dual<float>> (:217-275 evaluates L ~4 (n+m)^2 times).  px, pu: the item's (x_t, u_t); r: its record (layout L).
template <class L, bool RK4>
__device__ __forceinline__ void user_linearize_item(const quattro_model_params& p, const float* __restrict__ px,
                                                    const float* __restrict__ pu, float* __restrict__ r, const int j) {
  constexpr int NX = QT_USER_NX, NU = QT_USER_NU, NZ = NX + NU;
  using D = qtad::Dual<float>;
  using DD = qtad::Dual<D>;
  float xs[NX], us[NU];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = px[i];
#pragma unroll
  for (int a = 0; a < NU; ++a) us[a] = pu[a];
  {
    D xd[NX], ud[NU], xn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xd[i] = D(xs[i], i == j ? 1.0f : 0.0f);
#pragma unroll
    for (int a = 0; a < NU; ++a) ud[a] = D(us[a], NX + a == j ? 1.0f : 0.0f);
    qt_user::step<D, RK4>(p, xd, ud, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (j < NX)
        r[L::a(i, j)] = xn[i].d;
      else
        r[L::b(i, j - NX)] = xn[i].d;
    }
  }
  float gj = 0.0f;
#pragma unroll 1
  for (int c = 0; c < NZ; ++c) {
    DD xz[NX], uz[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) xz[i] = DD(D(xs[i], i == j ? 1.0f : 0.0f), D(i == c ? 1.0f : 0.0f, 0.0f));
#pragma unroll
    for (int a = 0; a < NU; ++a) uz[a] = DD(D(us[a], NX + a == j ? 1.0f : 0.0f), D(NX + a == c ? 1.0f : 0.0f, 0.0f));
    const DD l = qt_user::stage_cost<DD>(p, xz, uz);
    gj = l.v.d;
    const float h = l.d.d;                       // d2 L / dz_j dz_c
    if (j < NX) {
      if (c < NX) r[L::lxx(j, c)] = h;           // (the x-u block is written once, by the control lanes, as l_ux)
    } else if (c < NX) {
      r[L::lux(j - NX, c)] = h;
    } else {
      r[L::luu(j - NX, c - NX)] = h;
    }
  }
  if (j < NX)
    r[L::lx(j)] = gj;
  else
    r[L::lu(j - NX)] = gj;
}

// V_x(N)[i] = dLf/dx_i and row i of V_xx(N) = d2Lf/dx2 at x_N (reference: _finite_diff_gradient_final :149,
// _finite_diff_hessian_final :163)
__device__ __forceinline__ void user_terminal_row(const quattro_model_params& p, const float* __restrict__ xN, const int i,
                                                  float* __restrict__ Vx, float* __restrict__ Vxx) {
  constexpr int NX = QT_USER_NX;
  using D = qtad::Dual<float>;
  using DD = qtad::Dual<D>;
  float xs[NX];
#pragma unroll
  for (int q = 0; q < NX; ++q) xs[q] = xN[q];
  float gi = 0.0f;
#pragma unroll 1
  for (int c = 0; c < NX; ++c) {
    DD xz[NX];
#pragma unroll
    for (int q = 0; q < NX; ++q) xz[q] = DD(D(xs[q], q == i ? 1.0f : 0.0f), D(q == c ? 1.0f : 0.0f, 0.0f));
    const DD l = qt_user::final_cost<DD>(p, xz);
    gi = l.v.d;
    Vxx[(size_t)i * NX + c] = l.d.d;
  }
  Vx[i] = gi;
}

}  // namespace
