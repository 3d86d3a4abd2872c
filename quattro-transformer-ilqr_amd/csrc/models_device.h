// Built-in device models (fp32): plant, cost and their exact derivatives.
//
// The reference passes Python callables f, L, Lf into iLQR_TF (quattro_ilqr_tf.py:82-84) and differentiates them
// by finite differences (:149-275).  These are the device statements of the two shipped problems:
//   cart-pole : examples/cartpole/cartpole_dynamics.py:48-71,   cartpole_mpc.py:187-189,244-269
//   quadrotor : examples/quadrotor/quadrotor_dynamics.py:63-164, quadrotor_mpc.py:40-46,74-100
// Derivatives are analytic (fp32 cannot take eps=1e-5 second differences); they are checked against the
// reference's own finite differences through the oracle (tests/test_oracle_golden.py, tests/test_linearize_gpu.py).
#pragma once
#include "quattro_device.h"

template <int MODEL>
struct ModelDims;
template <>
struct ModelDims<QUATTRO_MODEL_CARTPOLE> {
  static constexpr int NX = 4, NU = 1;
};
template <>
struct ModelDims<QUATTRO_MODEL_QUADROTOR> {
  static constexpr int NX = 12, NU = 4;
};

// softplus_beta(z) = log(1 + exp(beta z)) / beta, overflow-safe; and the logistic function.
// exp/log go to the hardware v_exp_f32 / v_log_f32 (about 1e-7..2e-6 relative): the argument of the log is in
// (1, 2], where the absolute error 6e-8 of forming 1 + e bounds the error of the whole barrier term far below
// fp32 round-off of the cost it is added to.
__device__ __forceinline__ float qt_softplus(float z, float beta) {
  const float bz = beta * z;
  // (1 / beta is loop-invariant and hoisted; a true division is ~10 instructions in the serial rollout chain)
  return (fmaxf(bz, 0.0f) + __logf(1.0f + __expf(-fabsf(bz)))) * (1.0f / beta);
}
__device__ __forceinline__ float qt_sigmoid(float z) { return 1.0f / (1.0f + __expf(-z)); }

// sin and cos: quadrant reduction + minimax polynomials on [-pi/4, pi/4] (Cephes sinf/cosf kernels, <= 1 ulp there).
// |x| <= 2048 (every angle a trajectory can reasonably reach, including a diverging line-search candidate that tumbles
// through many turns) reduces branch-free in fp32 by Cody-Waite with a three-part pi/2 (exact products for |k| < 2^13).
// Larger arguments reduce in fp64 (pi/2 as a double-double; good to |x| ~ 1e15) — a short rarely-taken branch instead
// of the library's Payne-Hanek path, whose ~800 inlined instructions per call made whole solves several times slower
// whenever one lane of a wave took it.  Beyond 1e15 (only a numerically exploded candidate, whose cost is rejected
// whatever the angle) the result is sin = 0, cos = 1; inf / NaN give NaN.
// the two polynomial kernels on the reduced argument r in [-pi/4, pi/4]
__device__ __forceinline__ void qt_sincos_kernels(float r, float* sr, float* cr) {
  const float z = r * r;
  const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  *sr = fmaf(ps * z, r, r);
  const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  *cr = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
}
__device__ __forceinline__ void qt_sincos(float x, float* s, float* c) {
  // fast reduction unconditionally; the rare large-argument fix-up sits behind ONE wave-uniform branch (a per-lane
  // if/else costs ~8 exec-mask instructions per call even when no lane takes it)
  const float kf = rintf(x * 0.63661977236758134f);
  float r = fmaf(kf, -1.5703125f, x);                   // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.549789948768648e-8
  r = fmaf(kf, -4.837512969970703125e-4f, r);
  r = fmaf(kf, -7.549789948768648e-8f, r);
  int q = (int)kf;
  if (__builtin_expect(__any(!(fabsf(x) <= 2048.0f)), 0)) {
    if (!(fabsf(x) <= 2048.0f)) {
      if (fabsf(x) <= 1.0e15f) {
        const double xd = (double)x;
        const double kd = rint(xd * 0.63661977236758134308);
        const double rd = fma(kd, -6.123233995736766036e-17, fma(kd, -1.5707963267948965580, xd));
        r = (float)rd;
        q = (int)(kd - 4.0 * floor(kd * 0.25));         // k mod 4, exact: |k| < 2^53
      } else {
        r = x * 0.0f;                                   // 0, or NaN for inf / NaN
        q = 0;
      }
    }
  }
  float sr, cr;
  qt_sincos_kernels(r, &sr, &cr);
  const float sa = (q & 1) ? cr : sr, ca = (q & 1) ? sr : cr;
  *s = (q & 2) ? -sa : sa;
  *c = ((q + 1) & 2) ? -ca : ca;
}

// qt_sincos for the serial rollout chains, with a wave-uniform short cut: when every lane's |x| <= 0.78 (< pi/4) the quadrant
// index is 0 and the reduced argument is x itself (fmaf(+-0, c, x) == x exactly), so the reduction and the quadrant selects — 14
// of the ~25 instructions — drop out and the result is the same bit for bit.  A flying vehicle's Euler angles and a balanced
// pole live there; anything else (one lane suffices) takes the general path.
__device__ __forceinline__ void qt_sincos_chain(float x, float* s, float* c) {
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(fabsf(x) <= 0.78f)) != 0ull, 0)) {
    qt_sincos(x, s, c);
    return;
  }
  qt_sincos_kernels(x, s, c);
}

// ------------------------------------------------------------------------------------------------ cart-pole
struct CartpoleTerms {
  float xdd, thdd;                                  // accelerations
  float dxdd_th, dxdd_thd, dxdd_F, dthdd_th, dthdd_thd, dthdd_F;  // their partials
};

template <bool WITH_JAC>
__device__ __forceinline__ CartpoleTerms cartpole_terms(const quattro_model_params& p, float th, float thd, float F) {
  const float M = p.phys[0], mp = p.phys[1], l = p.phys[2], g = p.phys[3];
  float s, c;
  if constexpr (WITH_JAC) qt_sincos(th, &s, &c);
  else qt_sincos_chain(th, &s, &c);              // (the rollouts; same bits)
  const float mt = M + mp;
  const float imt = 1.0f / mt;
  const float tmp = (F + mp * l * thd * thd * s) * imt;
  const float den = l * (4.0f / 3.0f - mp * c * c * imt);
  const float iden = 1.0f / den;
  const float num = -g * s + c * tmp;
  CartpoleTerms o;
  o.thdd = num * iden;
  const float kk = mp * l * imt;
  o.xdd = tmp - kk * o.thdd * c;
  if (WITH_JAC) {
    const float dtmp_th = mp * l * thd * thd * c * imt;
    const float dtmp_thd = 2.0f * mp * l * thd * s * imt;
    const float dden_th = l * (2.0f * mp * c * s * imt);
    const float dnum_th = -g * c - s * tmp + c * dtmp_th;
    o.dthdd_th = (dnum_th * den - num * dden_th) * iden * iden;
    o.dthdd_thd = c * dtmp_thd * iden;
    o.dthdd_F = c * imt * iden;
    o.dxdd_th = dtmp_th - kk * (o.dthdd_th * c - o.thdd * s);
    o.dxdd_thd = dtmp_thd - kk * c * o.dthdd_thd;
    o.dxdd_F = imt - kk * c * o.dthdd_F;
  }
  return o;
}

// ------------------------------------------------------------------------------------------------ quadrotor
struct QuadTrig {
  float sph, cph, sth, cth, sps, cps, tth, sec;
};
__device__ __forceinline__ QuadTrig quad_trig(float phi, float th, float psi) {
  QuadTrig t;
  qt_sincos(phi, &t.sph, &t.cph);
  qt_sincos(th, &t.sth, &t.cth);
  qt_sincos(psi, &t.sps, &t.cps);
  {  // 1 / cos(theta): hardware reciprocal + one Newton step (<= 1 ulp) instead of the ~10-instruction division
    const float y = __builtin_amdgcn_rcpf(t.cth);
    t.sec = fmaf(fmaf(-t.cth, y, 1.0f), y, y);
  }
  t.tth = t.sth * t.sec;
  return t;
}

// ------------------------------------------------------------------------------------------------ generic API
template <int MODEL>
__device__ __forceinline__ void qt_rate(const quattro_model_params& p, const float* x, const float* u, float* xd);

template <>
__device__ __forceinline__ void qt_rate<QUATTRO_MODEL_CARTPOLE>(const quattro_model_params& p, const float* x,
                                                                const float* u, float* xd) {
  const CartpoleTerms t = cartpole_terms<false>(p, x[2], x[3], u[0]);
  xd[0] = x[1];
  xd[1] = t.xdd;
  xd[2] = x[3];
  xd[3] = t.thdd;
}

template <>
__device__ __forceinline__ void qt_rate<QUATTRO_MODEL_QUADROTOR>(const quattro_model_params& p, const float* x,
                                                                 const float* u, float* xd) {
  const float mass = p.phys[0], Ix = p.phys[1], Iy = p.phys[2], Iz = p.phys[3], arm = p.phys[4], grav = p.phys[5],
              kyaw = p.phys[6];
  // parameter-only quotients are loop-invariant (hoisted out of the rollout loop); the state-dependent ones are
  // multiplications by those reciprocals: a true fp32 division is ~10 instructions on the serial chain of a rollout
  const float inv_mass = 1.0f / mass, inv_Ix = 1.0f / Ix, inv_Iy = 1.0f / Iy, inv_Iz = 1.0f / Iz;
  const QuadTrig t = quad_trig(x[6], x[7], x[8]);
  const float wp = x[9], wq = x[10], wr = x[11];
  const float tm = (u[0] + u[1] + u[2] + u[3]) * inv_mass;
  xd[0] = x[3];
  xd[1] = x[4];
  xd[2] = x[5];
  xd[3] = tm * (t.sps * t.sph + t.cps * t.sth * t.cph);
  xd[4] = tm * (t.cps * t.sph - t.sps * t.sth * t.cph);
  xd[5] = -grav + tm * (t.cth * t.cph);
  const float mix = wq * t.sph + wr * t.cph;
  xd[6] = wp + mix * t.tth;
  xd[7] = wq * t.cph - wr * t.sph;
  xd[8] = mix * t.sec;
  const float tau_phi = arm * ((u[1] + u[2]) - (u[0] + u[3]));
  const float tau_th = arm * ((u[0] + u[1]) - (u[2] + u[3]));
  const float tau_psi = kyaw * (u[0] - u[1] + u[2] - u[3]);
  xd[9] = ((Iy - Iz) * inv_Ix) * (wq * wr) + tau_phi * inv_Ix;
  xd[10] = ((Iz - Ix) * inv_Iy) * (wp * wr) + tau_th * inv_Iy;
  xd[11] = ((Ix - Iy) * inv_Iz) * (wp * wq) + tau_psi * inv_Iz;
}

// x_next = f(x, u): explicit Euler or classic RK4 with zero-order-hold u.  The integrator is a compile-time choice
// (the launchers dispatch on p.integrator): with both in one kernel the Euler rollouts carried the RK4 code and its
// registers along.
template <int MODEL, bool RK4>
__device__ __forceinline__ void qt_step(const quattro_model_params& p, const float* x, const float* u, float* xn) {
  constexpr int NX = ModelDims<MODEL>::NX;
  const float dt = p.dt;
  float k1[NX];
  qt_rate<MODEL>(p, x, u, k1);
  if constexpr (!RK4) {
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = fmaf(dt, k1[i], x[i]);
    return;
  }
  float k2[NX], k3[NX], k4[NX], xs[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = fmaf(0.5f * dt, k1[i], x[i]);
  qt_rate<MODEL>(p, xs, u, k2);
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = fmaf(0.5f * dt, k2[i], x[i]);
  qt_rate<MODEL>(p, xs, u, k3);
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = fmaf(dt, k3[i], x[i]);
  qt_rate<MODEL>(p, xs, u, k4);
#pragma unroll
  for (int i = 0; i < NX; ++i) xn[i] = x[i] + (dt / 6.0f) * (k1[i] + 2.0f * k2[i] + 2.0f * k3[i] + k4[i]);
}

// running cost L(x,u) and terminal cost Lf(x)
template <int MODEL>
__device__ __forceinline__ float qt_stage_cost(const quattro_model_params& p, const float* x, const float* u) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  float c = 0.0f;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const float d = x[i] - p.x_ref[i];
    c = fmaf(p.q[i] * d, d, c);
  }
#pragma unroll
  for (int a = 0; a < NU; ++a) c = fmaf(p.r[a] * u[a], u[a], c);
  if (p.barrier_alpha != 0.0f) {
    // Exact shortcut: with every beta*u_a > 20, softplus_beta(-u_a) <= exp(-20)/beta, so the whole barrier term is below
    // NU * alpha * 4.25e-18 / beta^2.  When that is under half an ulp of c (c * 2^-25) the fmaf below returns c
    // unchanged, bit for bit, and the NU exp/log pairs can be skipped.  Taken only when every active lane of the
    // wave agrees (wave-uniform branch); a NaN control fails the comparison and takes the full path.
    float umin = u[0];
#pragma unroll
    for (int a = 1; a < NU; ++a) umin = fminf(umin, u[a]);
    const float ib = 1.0f / p.barrier_beta;
    const bool negligible = (p.barrier_beta * umin > 20.0f) &&
                            (fabsf(p.barrier_alpha) * (float)NU * 4.25e-18f * ib * ib < c * 2.9e-8f);
    if (!__all(negligible)) {
      float bar = 0.0f;
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        const float sp = qt_softplus(-u[a], p.barrier_beta);
        bar = fmaf(sp, sp, bar);
      }
      c = fmaf(p.barrier_alpha, bar, c);
    }
  }
  return c;
}

template <int MODEL>
__device__ __forceinline__ float qt_final_cost(const quattro_model_params& p, const float* x) {
  constexpr int NX = ModelDims<MODEL>::NX;
  float c = 0.0f;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const float d = x[i] - p.x_ref[i];
    c = fmaf(p.qf[i] * d, d, c);
  }
  return c;
}

// ------------------------------------------------------------------------------------------------ linearisation
// Record fillers for the Euler discretisation (A = I + dt Jx, B = dt Ju).  `fill_const` writes every entry that
// does not depend on (x,u) into a zeroed record once; `fill_state` overwrites the state-dependent entries.
// L is the record layout (RowMajorRec<NX,NU> or Tile16Rec).
template <int MODEL, class L>
struct EulerRecord;

template <class L>
struct EulerRecord<QUATTRO_MODEL_CARTPOLE, L> {
  static __device__ __forceinline__ void fill_const(float* rec, const quattro_model_params& p) {
    for (int i = 0; i < 4; ++i) {
      rec[L::a(i, i)] = 1.0f;
      rec[L::lxx(i, i)] = 2.0f * p.q[i];
    }
    rec[L::a(0, 1)] = p.dt;
    rec[L::a(2, 3)] = p.dt;
    rec[L::luu(0, 0)] = 2.0f * p.r[0];
  }
  static __device__ __forceinline__ void fill_state(float* rec, const quattro_model_params& p, const float* x,
                                                    const float* u) {
    const CartpoleTerms t = cartpole_terms<true>(p, x[2], x[3], u[0]);
    const float dt = p.dt;
    rec[L::a(1, 2)] = dt * t.dxdd_th;
    rec[L::a(1, 3)] = dt * t.dxdd_thd;
    rec[L::a(3, 2)] = dt * t.dthdd_th;
    rec[L::a(3, 3)] = fmaf(dt, t.dthdd_thd, 1.0f);
    rec[L::b(1, 0)] = dt * t.dxdd_F;
    rec[L::b(3, 0)] = dt * t.dthdd_F;
#pragma unroll
    for (int i = 0; i < 4; ++i) rec[L::lx(i)] = 2.0f * p.q[i] * (x[i] - p.x_ref[i]);
    float lu = 2.0f * p.r[0] * u[0], luu = 2.0f * p.r[0];
    if (p.barrier_alpha != 0.0f) {
      const float sp = qt_softplus(-u[0], p.barrier_beta), sg = qt_sigmoid(-p.barrier_beta * u[0]);
      lu = fmaf(p.barrier_alpha, -2.0f * sp * sg, lu);
      luu = fmaf(p.barrier_alpha, 2.0f * sg * sg + 2.0f * sp * p.barrier_beta * sg * (1.0f - sg), luu);
    }
    rec[L::lu(0)] = lu;
    rec[L::luu(0, 0)] = luu;
  }
};

template <class L>
struct EulerRecord<QUATTRO_MODEL_QUADROTOR, L> {
  static __device__ __forceinline__ void fill_const(float* rec, const quattro_model_params& p) {
    const float Ix = p.phys[1], Iy = p.phys[2], Iz = p.phys[3], arm = p.phys[4], kyaw = p.phys[6];
    const float dt = p.dt;
    for (int i = 0; i < 12; ++i) {
      rec[L::a(i, i)] = 1.0f;
      rec[L::lxx(i, i)] = 2.0f * p.q[i];
    }
    for (int i = 0; i < 3; ++i) rec[L::a(i, 3 + i)] = dt;
    rec[L::a(6, 9)] = dt;
    const float sphi[4] = {-1.0f, 1.0f, 1.0f, -1.0f}, sth[4] = {1.0f, 1.0f, -1.0f, -1.0f}, sps[4] = {1.0f, -1.0f, 1.0f, -1.0f};
    for (int a = 0; a < 4; ++a) {
      rec[L::b(9, a)] = dt * sphi[a] * (arm / Ix);
      rec[L::b(10, a)] = dt * sth[a] * (arm / Iy);
      rec[L::b(11, a)] = dt * sps[a] * (kyaw / Iz);
      rec[L::luu(a, a)] = 2.0f * p.r[a];
    }
  }
  static __device__ __forceinline__ void fill_state(float* rec, const quattro_model_params& p, const float* x,
                                                    const float* u) {
    // no implicit fma contraction here: the same entries are produced through different offset maps (TILE16, TILE16C,
    // ROWMAJOR) in different kernels, and they must come out bit-identical (explicit fmaf calls stay fused)
#pragma clang fp contract(off)
    const float mass = p.phys[0], Ix = p.phys[1], Iy = p.phys[2], Iz = p.phys[3];
    const float dt = p.dt;
    const QuadTrig t = quad_trig(x[6], x[7], x[8]);
    const float wp = x[9], wq = x[10], wr = x[11];
    const float tm = (u[0] + u[1] + u[2] + u[3]) / mass;
    const float rx = t.sps * t.sph + t.cps * t.sth * t.cph;
    const float ry = t.cps * t.sph - t.sps * t.sth * t.cph;
    const float rz = t.cth * t.cph;
    const float dtm = dt * tm;
    rec[L::a(3, 6)] = dtm * (t.sps * t.cph - t.cps * t.sth * t.sph);
    rec[L::a(3, 7)] = dtm * (t.cps * t.cth * t.cph);
    rec[L::a(3, 8)] = dtm * ry;
    rec[L::a(4, 6)] = dtm * (t.cps * t.cph + t.sps * t.sth * t.sph);
    rec[L::a(4, 7)] = -dtm * (t.sps * t.cth * t.cph);
    rec[L::a(4, 8)] = -dtm * rx;
    rec[L::a(5, 6)] = -dtm * t.cth * t.sph;
    rec[L::a(5, 7)] = -dtm * t.sth * t.cph;
    const float dm = dt / mass;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      rec[L::b(3, a)] = dm * rx;
      rec[L::b(4, a)] = dm * ry;
      rec[L::b(5, a)] = dm * rz;
    }
    const float mix = wq * t.sph + wr * t.cph;
    const float dmix = wq * t.cph - wr * t.sph;
    const float sec2 = t.sec * t.sec;
    rec[L::a(6, 6)] = fmaf(dt, dmix * t.tth, 1.0f);
    rec[L::a(6, 7)] = dt * mix * sec2;
    rec[L::a(6, 10)] = dt * t.sph * t.tth;
    rec[L::a(6, 11)] = dt * t.cph * t.tth;
    rec[L::a(7, 6)] = -dt * mix;
    rec[L::a(7, 10)] = dt * t.cph;
    rec[L::a(7, 11)] = -dt * t.sph;
    rec[L::a(8, 6)] = dt * dmix * t.sec;
    rec[L::a(8, 7)] = dt * mix * t.sth * sec2;
    rec[L::a(8, 10)] = dt * t.sph * t.sec;
    rec[L::a(8, 11)] = dt * t.cph * t.sec;
    const float c1 = (Iy - Iz) / Ix, c2 = (Iz - Ix) / Iy, c3 = (Ix - Iy) / Iz;
    rec[L::a(9, 10)] = dt * c1 * wr;
    rec[L::a(9, 11)] = dt * c1 * wq;
    rec[L::a(10, 9)] = dt * c2 * wr;
    rec[L::a(10, 11)] = dt * c2 * wp;
    rec[L::a(11, 9)] = dt * c3 * wq;
    rec[L::a(11, 10)] = dt * c3 * wp;
#pragma unroll
    for (int i = 0; i < 12; ++i) rec[L::lx(i)] = 2.0f * p.q[i] * (x[i] - p.x_ref[i]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float lu = 2.0f * p.r[a] * u[a], luu = 2.0f * p.r[a];
      if (p.barrier_alpha != 0.0f) {
        const float sp = qt_softplus(-u[a], p.barrier_beta), sg = qt_sigmoid(-p.barrier_beta * u[a]);
        lu = fmaf(p.barrier_alpha, -2.0f * sp * sg, lu);
        luu = fmaf(p.barrier_alpha, 2.0f * sg * sg + 2.0f * sp * p.barrier_beta * sg * (1.0f - sg), luu);
      }
      rec[L::lu(a)] = lu;
      rec[L::luu(a, a)] = luu;
    }
  }
};

// ------------------------------------------------------------------------------------------------ forward-mode pieces
// Directional derivative of the continuous rate function: xd_dot = (d rate / d x) dx + (d rate / d u) du at (x, u).
// Used to push the 16 (5) unit directions of z = (x, u) through the RK4 stages (linearize_rk4_kernel).
template <int MODEL>
__device__ __forceinline__ void qt_rate_jvp(const quattro_model_params& p, const float* x, const float* u,
                                            const float* dx, const float* du, float* out);

template <>
__device__ __forceinline__ void qt_rate_jvp<QUATTRO_MODEL_CARTPOLE>(const quattro_model_params& p, const float* x,
                                                                    const float* u, const float* dx, const float* du,
                                                                    float* out) {
  const CartpoleTerms t = cartpole_terms<true>(p, x[2], x[3], u[0]);
  out[0] = dx[1];
  out[1] = t.dxdd_th * dx[2] + t.dxdd_thd * dx[3] + t.dxdd_F * du[0];
  out[2] = dx[3];
  out[3] = t.dthdd_th * dx[2] + t.dthdd_thd * dx[3] + t.dthdd_F * du[0];
}

template <>
__device__ __forceinline__ void qt_rate_jvp<QUATTRO_MODEL_QUADROTOR>(const quattro_model_params& p, const float* x,
                                                                     const float* u, const float* dx, const float* du,
                                                                     float* out) {
  const float mass = p.phys[0], Ix = p.phys[1], Iy = p.phys[2], Iz = p.phys[3], arm = p.phys[4], kyaw = p.phys[6];
  const QuadTrig t = quad_trig(x[6], x[7], x[8]);
  const float wp = x[9], wq = x[10], wr = x[11];
  const float tm = (u[0] + u[1] + u[2] + u[3]) / mass;
  const float dT = (du[0] + du[1] + du[2] + du[3]) / mass;
  const float rx = t.sps * t.sph + t.cps * t.sth * t.cph;
  const float ry = t.cps * t.sph - t.sps * t.sth * t.cph;
  const float rz = t.cth * t.cph;
  const float dphi = dx[6], dth = dx[7], dpsi = dx[8];
  out[0] = dx[3];
  out[1] = dx[4];
  out[2] = dx[5];
  out[3] = tm * ((t.sps * t.cph - t.cps * t.sth * t.sph) * dphi + (t.cps * t.cth * t.cph) * dth + ry * dpsi) + rx * dT;
  out[4] = tm * ((t.cps * t.cph + t.sps * t.sth * t.sph) * dphi - (t.sps * t.cth * t.cph) * dth - rx * dpsi) + ry * dT;
  out[5] = tm * (-t.cth * t.sph * dphi - t.sth * t.cph * dth) + rz * dT;
  const float mix = wq * t.sph + wr * t.cph;
  const float dmix = wq * t.cph - wr * t.sph;
  const float sec2 = t.sec * t.sec;
  out[6] = dmix * t.tth * dphi + mix * sec2 * dth + dx[9] + t.sph * t.tth * dx[10] + t.cph * t.tth * dx[11];
  out[7] = -mix * dphi + t.cph * dx[10] - t.sph * dx[11];
  out[8] = dmix * t.sec * dphi + mix * t.sth * sec2 * dth + t.sph * t.sec * dx[10] + t.cph * t.sec * dx[11];
  const float c1 = (Iy - Iz) / Ix, c2 = (Iz - Ix) / Iy, c3 = (Ix - Iy) / Iz;
  out[9] = c1 * (wr * dx[10] + wq * dx[11]) + (arm / Ix) * ((du[1] + du[2]) - (du[0] + du[3]));
  out[10] = c2 * (wr * dx[9] + wp * dx[11]) + (arm / Iy) * ((du[0] + du[1]) - (du[2] + du[3]));
  out[11] = c3 * (wq * dx[9] + wp * dx[10]) + (kyaw / Iz) * (du[0] - du[1] + du[2] - du[3]);
}

// The quadrotor's rate function and its directional derivative at one stage point, split into "everything that depends on
// the point" (QuadStage: trig of the three angles, the body rates, the thrust — computed ONCE) and its application to a
// direction.  linearize_rk4_quad_kernel pushes all 16 unit directions of z = (x, u) through the four RK4 stages of ONE
// lane's item with these; the per-direction work is then a few dozen multiply-adds (most of them folded away for a unit
// vector), where the generic qt_rate_jvp recomputes the trig for every direction.
struct QuadStage {
  QuadTrig t;
  float wp, wq, wr, tm, rx, ry, rz, mix, dmix, sec2;
};
__device__ __forceinline__ QuadStage quad_stage(const quattro_model_params& p, const float* x, const float* u) {
  QuadStage s;
  s.t = quad_trig(x[6], x[7], x[8]);
  s.wp = x[9];
  s.wq = x[10];
  s.wr = x[11];
  s.tm = (u[0] + u[1] + u[2] + u[3]) / p.phys[0];
  s.rx = s.t.sps * s.t.sph + s.t.cps * s.t.sth * s.t.cph;
  s.ry = s.t.cps * s.t.sph - s.t.sps * s.t.sth * s.t.cph;
  s.rz = s.t.cth * s.t.cph;
  s.mix = s.wq * s.t.sph + s.wr * s.t.cph;
  s.dmix = s.wq * s.t.cph - s.wr * s.t.sph;
  s.sec2 = s.t.sec * s.t.sec;
  return s;
}
// xd = rate(x, u) at the stage point (the expressions of qt_rate<QUADROTOR>)
__device__ __forceinline__ void quad_rate_at(const QuadStage& s, const quattro_model_params& p, const float* x,
                                             const float* u, float* xd) {
  const float Ix = p.phys[1], Iy = p.phys[2], Iz = p.phys[3], arm = p.phys[4], grav = p.phys[5], kyaw = p.phys[6];
  xd[0] = x[3];
  xd[1] = x[4];
  xd[2] = x[5];
  xd[3] = s.tm * s.rx;
  xd[4] = s.tm * s.ry;
  xd[5] = -grav + s.tm * s.rz;
  xd[6] = s.wp + s.mix * s.t.tth;
  xd[7] = s.dmix;
  xd[8] = s.mix * s.t.sec;
  xd[9] = ((Iy - Iz) / Ix) * (s.wq * s.wr) + (arm / Ix) * ((u[1] + u[2]) - (u[0] + u[3]));
  xd[10] = ((Iz - Ix) / Iy) * (s.wp * s.wr) + (arm / Iy) * ((u[0] + u[1]) - (u[2] + u[3]));
  xd[11] = ((Ix - Iy) / Iz) * (s.wp * s.wq) + (kyaw / Iz) * (u[0] - u[1] + u[2] - u[3]);
}
// out = (d rate / d x) dx + (d rate / d u) du at the stage point (the expressions of qt_rate_jvp<QUADROTOR>)
__device__ __forceinline__ void quad_jvp_at(const QuadStage& s, const quattro_model_params& p, const float* dx,
                                            const float* du, float* out) {
  const float mass = p.phys[0], Ix = p.phys[1], Iy = p.phys[2], Iz = p.phys[3], arm = p.phys[4], kyaw = p.phys[6];
  const QuadTrig& t = s.t;
  const float dT = (du[0] + du[1] + du[2] + du[3]) / mass;
  const float dphi = dx[6], dth = dx[7], dpsi = dx[8];
  out[0] = dx[3];
  out[1] = dx[4];
  out[2] = dx[5];
  out[3] = s.tm * ((t.sps * t.cph - t.cps * t.sth * t.sph) * dphi + (t.cps * t.cth * t.cph) * dth + s.ry * dpsi) + s.rx * dT;
  out[4] = s.tm * ((t.cps * t.cph + t.sps * t.sth * t.sph) * dphi - (t.sps * t.cth * t.cph) * dth - s.rx * dpsi) + s.ry * dT;
  out[5] = s.tm * (-t.cth * t.sph * dphi - t.sth * t.cph * dth) + s.rz * dT;
  out[6] = s.dmix * t.tth * dphi + s.mix * s.sec2 * dth + dx[9] + t.sph * t.tth * dx[10] + t.cph * t.tth * dx[11];
  out[7] = -s.mix * dphi + t.cph * dx[10] - t.sph * dx[11];
  out[8] = s.dmix * t.sec * dphi + s.mix * t.sth * s.sec2 * dth + t.sph * t.sec * dx[10] + t.cph * t.sec * dx[11];
  const float c1 = (Iy - Iz) / Ix, c2 = (Iz - Ix) / Iy, c3 = (Ix - Iy) / Iz;
  out[9] = c1 * (s.wr * dx[10] + s.wq * dx[11]) + (arm / Ix) * ((du[1] + du[2]) - (du[0] + du[3]));
  out[10] = c2 * (s.wr * dx[9] + s.wp * dx[11]) + (arm / Iy) * ((du[0] + du[1]) - (du[2] + du[3]));
  out[11] = c3 * (s.wq * dx[9] + s.wp * dx[10]) + (kyaw / Iz) * (du[0] - du[1] + du[2] - du[3]);
}

// cost derivative entries of a record (independent of the integrator): l_x, l_u, diag(l_xx), diag(l_uu); l_ux = 0
template <int MODEL, class L>
__device__ __forceinline__ void fill_cost_entries(float* rec, const quattro_model_params& p, const float* x,
                                                  const float* u) {
  // no implicit fma contraction (as in EulerRecord::fill_state): the RK4 kernels of different layouts call this from
  // different surroundings and their cost entries must come out bit-identical (explicit fmaf calls stay fused)
#pragma clang fp contract(off)
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    rec[L::lx(i)] = 2.0f * p.q[i] * (x[i] - p.x_ref[i]);
    rec[L::lxx(i, i)] = 2.0f * p.q[i];
  }
#pragma unroll
  for (int a = 0; a < NU; ++a) {
    float lu = 2.0f * p.r[a] * u[a], luu = 2.0f * p.r[a];
    if (p.barrier_alpha != 0.0f) {
      const float sp = qt_softplus(-u[a], p.barrier_beta), sg = qt_sigmoid(-p.barrier_beta * u[a]);
      lu = fmaf(p.barrier_alpha, -2.0f * sp * sg, lu);
      luu = fmaf(p.barrier_alpha, 2.0f * sg * sg + 2.0f * sp * p.barrier_beta * sg * (1.0f - sg), luu);
    }
    rec[L::lu(a)] = lu;
    rec[L::luu(a, a)] = luu;
  }
}

// ------------------------------------------------------------------------------------------------ user models
// a third model, compiled in from the caller's own source (user_model.h); absent from libquattro_hip.so itself
#ifdef QT_USER_MODEL_HEADER
#include "user_model.h"
#endif
