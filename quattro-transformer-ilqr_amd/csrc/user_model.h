// A user-supplied device model: the MI355X answer to the reference taking arbitrary Python callables f, L, Lf
// (quattro_ilqr_tf/quattro_ilqr_tf.py:66-84).  A kernel cannot call Python; instead the problem is written ONCE as three
// function templates over its scalar type (quattro_ilqr_amd.user_model.compile_model generates this header from three
// expression bodies) and compiled, with the generic kernels of this directory, into a library of its own that exports the
// same C ABI with model id QUATTRO_MODEL_USER:
//
//   template <class T> __device__ void rate(const quattro_model_params& p, const T* x, const T* u, T* xd);   // x' = rate(x, u)
//   template <class T> __device__ T stage_cost(const quattro_model_params& p, const T* x, const T* u);       // L(x, u)
//   template <class T> __device__ T final_cost(const quattro_model_params& p, const T* x);                   // Lf(x)
//
// T = float in the rollouts; T = Dual<float> / Dual<Dual<float>> (dual.h) where the reference takes finite differences
// (:149-275): columns of [A | B] through the whole integrator step, gradient and Hessian of L and Lf.  p.phys[0..7] are the
// model's free parameters, p.q / p.r / p.qf / p.x_ref / p.barrier_* are there for the cost to use (default_stage_cost /
// default_final_cost are the built-in diagonal forms).  Compiled with -DQT_USER_MODEL_HEADER="..." -DQT_USER_NX=n -DQT_USER_NU=m.
#pragma once
#include "dual.h"

#if QT_USER_NX < 1 || QT_USER_NX > QUATTRO_MAX_NX || QT_USER_NU < 1 || QT_USER_NU > QUATTRO_MAX_NU
#error "user model: 1 <= n <= QUATTRO_MAX_NX and 1 <= m <= QUATTRO_MAX_NU"
#endif

namespace qt_user {
// the elementary functions a model may call, for every scalar type (these names hide the global ones in here)
using qtad::abs; using qtad::atan; using qtad::cos; using qtad::exp; using qtad::fabs; using qtad::fmax; using qtad::fmin;
using qtad::log; using qtad::pow; using qtad::primal; using qtad::sin; using qtad::sincos; using qtad::softplus;
using qtad::sqrt; using qtad::square; using qtad::tan; using qtad::tanh;

constexpr int NX = QT_USER_NX, NU = QT_USER_NU;

// L = sum_i q_i (x_i - x_ref_i)^2 + sum_a r_a u_a^2 + barrier_alpha sum_a softplus_beta(-u_a)^2, Lf = sum_i qf_i (x_i - x_ref_i)^2:
// the cost form of both shipped problems (include/quattro_hip.h), for any scalar type
template <class T>
__device__ __forceinline__ T default_stage_cost(const quattro_model_params& p, const T* x, const T* u) {
  T c(0.0f);
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const T d = x[i] - p.x_ref[i];
    c = c + d * d * p.q[i];
  }
#pragma unroll
  for (int a = 0; a < NU; ++a) c = c + u[a] * u[a] * p.r[a];
  if (p.barrier_alpha != 0.0f) {
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      const T sp = softplus(-u[a], p.barrier_beta);
      c = c + sp * sp * p.barrier_alpha;
    }
  }
  return c;
}
template <class T>
__device__ __forceinline__ T default_final_cost(const quattro_model_params& p, const T* x) {
  T c(0.0f);
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const T d = x[i] - p.x_ref[i];
    c = c + d * d * p.qf[i];
  }
  return c;
}

#include QT_USER_MODEL_HEADER

// x_next = f(x, u) for any scalar type: explicit Euler or classic RK4 with zero-order-hold u (what qt_step does in fp32)
template <class T, bool RK4>
__device__ __forceinline__ void step(const quattro_model_params& p, const T* x, const T* u, T* xn) {
  const float dt = p.dt;
  T k1[NX];
  rate<T>(p, x, u, k1);
  if constexpr (!RK4) {
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = x[i] + k1[i] * dt;
    return;
  }
  T k2[NX], k3[NX], k4[NX], xs[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = x[i] + k1[i] * (0.5f * dt);
  rate<T>(p, xs, u, k2);
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = x[i] + k2[i] * (0.5f * dt);
  rate<T>(p, xs, u, k3);
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = x[i] + k3[i] * dt;
  rate<T>(p, xs, u, k4);
#pragma unroll
  for (int i = 0; i < NX; ++i) xn[i] = x[i] + (k1[i] + k2[i] * 2.0f + k3[i] * 2.0f + k4[i]) * (dt / 6.0f);
}
}  // namespace qt_user

template <>
struct ModelDims<QUATTRO_MODEL_USER> {
  static constexpr int NX = QT_USER_NX, NU = QT_USER_NU;
};

template <>
__device__ __forceinline__ void qt_rate<QUATTRO_MODEL_USER>(const quattro_model_params& p, const float* x, const float* u,
                                                            float* xd) {
  qt_user::rate<float>(p, x, u, xd);
}
template <>
__device__ __forceinline__ float qt_stage_cost<QUATTRO_MODEL_USER>(const quattro_model_params& p, const float* x,
                                                                   const float* u) {
  return qt_user::stage_cost<float>(p, x, u);
}
template <>
__device__ __forceinline__ float qt_final_cost<QUATTRO_MODEL_USER>(const quattro_model_params& p, const float* x) {
  return qt_user::final_cost<float>(p, x);
}
