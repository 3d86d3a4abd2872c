// Rollout kernels: open-loop simulate + cost, closed-loop forward passes for all line-search step sizes, and
// the fused line search (evaluate every alpha, accept the first that does not increase the cost, commit it).
//
// Arithmetic replaced (reference quattro_ilqr_tf/quattro_ilqr_tf.py): simulate :127-132, compute_total_cost
// :138-143, forward_pass :377-390, and the alpha loop + stop test of optimize :433-451, :472 (:546-563, :584).
//
// Serial in t by nature (x'_{t+1} depends on x'_t); parallel over (trajectory, alpha).  Eight lanes per
// trajectory (one per alpha, n_alpha <= 8) sit next to each other in a wave, so the loads of K_t, k_t, x_t, u_t
// they share are single broadcast requests, and "first accepted alpha" is one ballot.  The nominal data of step
// t+2 is requested while step t is computed (two register buffers, loop unrolled by two): with well under one wave
// per SIMD there is no other wave to hide the load latency behind.
// States and controls are fp32; the running sum of the cost is kept in fp64 because the accept test
// `cand_cost <= current_cost` (:444) and the stop test |dJ| < tol (:472) compare nearly equal totals.
#include "rollout_body.h"

namespace {

template <int MODEL, bool RK4>
__global__ __launch_bounds__(64) void simulate_kernel(const quattro_model_params p, const float* __restrict__ x0,
                                const float* __restrict__ u, int B, int N, float* __restrict__ x,
                                double* __restrict__ cost) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  simulate_body<MODEL, RK4>(p, x0, u, N, x, cost, b);
}

// sum_t L(x_t,u_t) + Lf(x_N) of given (x,u) sequences, which need not satisfy the dynamics
template <int MODEL>
__global__ __launch_bounds__(64) void total_cost_kernel(const quattro_model_params p, const float* __restrict__ x,
                                  const float* __restrict__ u, int B, int N, double* __restrict__ cost) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* xb = x + (size_t)b * (N + 1) * NX;
  const float* ub = u + (size_t)b * N * NU;
  double J = 0.0;
  float xt[NX], ut[NU];
  for (int t = 0; t < N; ++t) {
    load_vec<NX>(xb + (size_t)t * NX, xt);
    load_vec<NU>(ub + (size_t)t * NU, ut);
    J += (double)qt_stage_cost<MODEL>(p, xt, ut);
  }
  load_vec<NX>(xb + (size_t)N * NX, xt);
  J += (double)qt_final_cost<MODEL>(p, xt);
  cost[b] = J;
}

// costs (and optionally trajectories) of every (trajectory, alpha): thread = b * 8 + alpha slot
template <int MODEL, bool RK4>
__global__ __launch_bounds__(64) void rollout_kernel(const quattro_model_params p, const float* __restrict__ x_nom,
                               const float* __restrict__ u_nom, const float* __restrict__ K,
                               const float* __restrict__ k, AlphaList al, int n_alpha, int B, int N,
                               float* __restrict__ x_new, float* __restrict__ u_new, double* __restrict__ cost,
                               const int32_t* __restrict__ active) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3, ai = gid & 7;
  if (b >= B || ai >= n_alpha) return;
  if (active != nullptr && active[b] == 0) return;
  const float* xn = x_nom + (size_t)b * (N + 1) * NX;
  const float* un = u_nom + (size_t)b * N * NU;
  const float* Kb = K + (size_t)b * N * NU * NX;
  const float* kb = k + (size_t)b * N * NU;
  double J;
  if (x_new != nullptr && u_new != nullptr) {
    float* xo = x_new + ((size_t)ai * B + b) * (N + 1) * NX;
    float* uo = u_new + ((size_t)ai * B + b) * N * NU;
    float x0[NX];
    load_vec<NX>(xn, x0);
    store_vec<NX>(xo, x0);
    J = rollout_closed<MODEL, RK4>(p, xn, un, Kb, kb, al.a[ai], N, ArrayStore<MODEL>{xo, uo});
  } else {
    J = rollout_closed<MODEL, RK4>(p, xn, un, Kb, kb, al.a[ai], N, NoStore{});
  }
  cost[(size_t)ai * B + b] = J;
}

// Fused line search: 8 lanes per trajectory (linesearch_body)
template <int MODEL, bool RK4>
__global__ __launch_bounds__(64) void linesearch_kernel(const quattro_model_params p, float* x_nom, float* u_nom,
                                  const float* __restrict__ K, const float* __restrict__ k, AlphaList al, int n_alpha,
                                  int B, int N, double tol, double* cost, int32_t* __restrict__ alpha_idx,
                                  int32_t* active, int32_t* iters, float* __restrict__ scratch) {
  linesearch_body<MODEL, RK4, 8>(p, x_nom, u_nom, K, k, al, n_alpha, B, N, tol, cost, alpha_idx, active, iters, scratch,
                                 blockIdx.x * blockDim.x + threadIdx.x, false);
}

}  // namespace

#define QT_DISPATCH_INTEG(p, ...)                                 \
  if ((p).integrator == QUATTRO_INTEGRATOR_EULER) {               \
    constexpr bool RK4 = false;                                   \
    __VA_ARGS__;                                                  \
  } else if ((p).integrator == QUATTRO_INTEGRATOR_RK4) {          \
    constexpr bool RK4 = true;                                    \
    __VA_ARGS__;                                                  \
  } else {                                                        \
    return QUATTRO_ERR_UNSUPPORTED;                               \
  }
// the one-lane-per-candidate kernels of this file serve the cart-pole; quadrotor calls were routed to rollout_quad.hip
// by the launchers before they get here
#ifdef QT_USER_MODEL_HEADER
#define QT_DISPATCH_USER(p, ...)                           \
  else if ((p).model_id == QUATTRO_MODEL_USER) {           \
    constexpr int MODEL = QUATTRO_MODEL_USER;              \
    QT_DISPATCH_INTEG(p, __VA_ARGS__);                     \
  }
#else
#define QT_DISPATCH_USER(p, ...)
#endif
#define QT_DISPATCH_MODEL(p, ...)                          \
  if ((p).model_id == QUATTRO_MODEL_CARTPOLE) {            \
    constexpr int MODEL = QUATTRO_MODEL_CARTPOLE;          \
    QT_DISPATCH_INTEG(p, __VA_ARGS__);                     \
  }                                                        \
  QT_DISPATCH_USER(p, __VA_ARGS__)                         \
  else {                                                   \
    return QUATTRO_ERR_UNSUPPORTED;                        \
  }

// quadrotor: lane-cooperative kernels (rollout_quad.hip)
int quattro_launch_simulate_quad(const quattro_model_params&, const float*, const float*, int, int, float*, double*,
                                 hipStream_t);
int quattro_launch_rollout_quad(const quattro_model_params&, const float*, const float*, const float*, const float*,
                                const float*, int, int, int, float*, float*, double*, const int32_t*, hipStream_t);
int quattro_launch_linesearch_quad(const quattro_model_params&, float*, float*, const float*, const float*,
                                   const float*, int, int, int, double, double*, int32_t*, int32_t*, int32_t*, float*,
                                   hipStream_t);

size_t quattro_linesearch_scratch_bytes_impl(int n, int m, int B, int N) {
  const size_t cs = (size_t)((n + m + 3) / 4 * 4);
  return (size_t)B * 8 * (size_t)N * cs * sizeof(float);
}

int quattro_launch_simulate(const quattro_model_params& p, const float* x0, const float* u, int B, int N, float* x,
                            double* cost, hipStream_t stream) {
  if (p.model_id == QUATTRO_MODEL_QUADROTOR) return quattro_launch_simulate_quad(p, x0, u, B, N, x, cost, stream);
  const int threads = 64;
  QT_DISPATCH_MODEL(p, hipLaunchKernelGGL((simulate_kernel<MODEL, RK4>), dim3((B + threads - 1) / threads), dim3(threads),
                                          0, stream, p, x0, u, B, N, x, cost));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_total_cost(const quattro_model_params& p, const float* x, const float* u, int B, int N,
                              double* cost, hipStream_t stream) {
  const int threads = 64;
  const dim3 grid((B + threads - 1) / threads);
  if (p.model_id == QUATTRO_MODEL_CARTPOLE)
    hipLaunchKernelGGL((total_cost_kernel<QUATTRO_MODEL_CARTPOLE>), grid, dim3(threads), 0, stream, p, x, u, B, N, cost);
  else if (p.model_id == QUATTRO_MODEL_QUADROTOR)
    hipLaunchKernelGGL((total_cost_kernel<QUATTRO_MODEL_QUADROTOR>), grid, dim3(threads), 0, stream, p, x, u, B, N, cost);
#ifdef QT_USER_MODEL_HEADER
  else if (p.model_id == QUATTRO_MODEL_USER)
    hipLaunchKernelGGL((total_cost_kernel<QUATTRO_MODEL_USER>), grid, dim3(threads), 0, stream, p, x, u, B, N, cost);
#endif
  else
    return QUATTRO_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_rollout(const quattro_model_params& p, const float* x_nom, const float* u_nom, const float* K,
                           const float* k, const float* alphas, int n_alpha, int B, int N, float* x_new, float* u_new,
                           double* cost, const int32_t* active, hipStream_t stream) {
  if (p.model_id == QUATTRO_MODEL_QUADROTOR)
    return quattro_launch_rollout_quad(p, x_nom, u_nom, K, k, alphas, n_alpha, B, N, x_new, u_new, cost, active, stream);
  AlphaList al;
  for (int i = 0; i < QUATTRO_MAX_ALPHAS; ++i) al.a[i] = i < n_alpha ? alphas[i] : 0.0f;
  const int threads = 64;
  const long long tot = (long long)B * 8;
  QT_DISPATCH_MODEL(p, hipLaunchKernelGGL((rollout_kernel<MODEL, RK4>), dim3((unsigned)((tot + threads - 1) / threads)),
                                          dim3(threads), 0, stream, p, x_nom, u_nom, K, k, al, n_alpha, B, N, x_new,
                                          u_new, cost, active));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_linesearch(const quattro_model_params& p, float* x_nom, float* u_nom, const float* K,
                              const float* k, const float* alphas, int n_alpha, int B, int N, double tol, double* cost,
                              int32_t* alpha_idx, int32_t* active, int32_t* iters, float* scratch,
                              hipStream_t stream) {
  if (p.model_id == QUATTRO_MODEL_QUADROTOR)
    return quattro_launch_linesearch_quad(p, x_nom, u_nom, K, k, alphas, n_alpha, B, N, tol, cost, alpha_idx, active,
                                          iters, scratch, stream);
  AlphaList al;
  for (int i = 0; i < QUATTRO_MAX_ALPHAS; ++i) al.a[i] = i < n_alpha ? alphas[i] : 0.0f;
  const int threads = 64;
  const long long tot = (long long)B * 8;
  QT_DISPATCH_MODEL(p, hipLaunchKernelGGL((linesearch_kernel<MODEL, RK4>), dim3((unsigned)((tot + threads - 1) / threads)),
                                          dim3(threads), 0, stream, p, x_nom, u_nom, K, k, al, n_alpha, B, N, tol, cost,
                                          alpha_idx, active, iters, scratch));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
