// Rollout kernels: open-loop simulate + cost, closed-loop forward passes for all line-search step sizes, and
// the fused line search (evaluate every alpha, accept the first that does not increase the cost, commit it).
//
// Arithmetic replaced (reference quattro_ilqr_tf/quattro_ilqr_tf.py): simulate :127-132, compute_total_cost
// :138-143, forward_pass :377-390, and the alpha loop + stop test of optimize :433-451, :472 (:546-563, :584).
//
// Serial in t by nature (x'_{t+1} depends on x'_t); parallel over (trajectory, alpha).  Eight lanes per
// trajectory (one per alpha, n_alpha <= 8) sit next to each other in a wave, so the loads of K_t, k_t, x_t, u_t
// they share are single broadcast requests, and "first accepted alpha" is one ballot.
// States and controls are fp32; the running sum of the cost is kept in fp64 because the accept test
// `cand_cost <= current_cost` (:444) and the stop test |dJ| < tol (:472) compare nearly equal totals.
#include "models_device.h"

namespace {

struct AlphaList {
  float a[QUATTRO_MAX_ALPHAS];
};

// One closed-loop rollout.  xnom/unom/K/k point at this trajectory's data.  When WRITE, new states/controls go to
// xout/uout, which MAY alias xnom/unom (in-place commit): the nominal row t+1 is read before row t+1 is written.
template <int MODEL, bool WRITE>
__device__ __forceinline__ double rollout_closed(const quattro_model_params& p, const float* __restrict__ xnom,
                                                 const float* __restrict__ unom, const float* __restrict__ K,
                                                 const float* __restrict__ k, float alpha, int N, float* xout,
                                                 float* uout) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  float xh[NX], xn_t[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    xn_t[i] = xnom[i];
    xh[i] = xn_t[i];   // x'_0 = x_0
  }
  if (WRITE) {
#pragma unroll
    for (int i = 0; i < NX; ++i) xout[i] = xh[i];
  }
  double J = 0.0;
  for (int t = 0; t < N; ++t) {
    float un[NU], kt[NU], Kt[NU * NX], xn_next[NX];
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      un[a] = unom[t * NU + a];
      kt[a] = k[t * NU + a];
    }
#pragma unroll
    for (int e = 0; e < NU * NX; ++e) Kt[e] = K[(size_t)t * NU * NX + e];
#pragma unroll
    for (int i = 0; i < NX; ++i) xn_next[i] = xnom[(t + 1) * NX + i];
    float uh[NU];
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      float du = kt[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) du = fmaf(Kt[a * NX + i], xh[i] - xn_t[i], du);
      uh[a] = fmaf(alpha, du, un[a]);
    }
    J += (double)qt_stage_cost<MODEL>(p, xh, uh);
    float xnext[NX];
    qt_step<MODEL>(p, xh, uh, xnext);
    if (WRITE) {
      // the old nominal row t+1 must be in registers before it is overwritten
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int a = 0; a < NU; ++a) uout[t * NU + a] = uh[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) xout[(t + 1) * NX + i] = xnext[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      xh[i] = xnext[i];
      xn_t[i] = xn_next[i];
    }
  }
  J += (double)qt_final_cost<MODEL>(p, xh);
  return J;
}

template <int MODEL>
__global__ void simulate_kernel(const quattro_model_params p, const float* __restrict__ x0,
                                const float* __restrict__ u, int B, int N, float* __restrict__ x,
                                double* __restrict__ cost) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float xh[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xh[i] = x0[(size_t)b * NX + i];
  float* xo = x + (size_t)b * (N + 1) * NX;
  const float* ub = u + (size_t)b * N * NU;
#pragma unroll
  for (int i = 0; i < NX; ++i) xo[i] = xh[i];
  double J = 0.0;
  for (int t = 0; t < N; ++t) {
    float ut[NU], xn[NX];
#pragma unroll
    for (int a = 0; a < NU; ++a) ut[a] = ub[t * NU + a];
    J += (double)qt_stage_cost<MODEL>(p, xh, ut);
    qt_step<MODEL>(p, xh, ut, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      xh[i] = xn[i];
      xo[(t + 1) * NX + i] = xn[i];
    }
  }
  J += (double)qt_final_cost<MODEL>(p, xh);
  if (cost != nullptr) cost[b] = J;
}

// sum_t L(x_t,u_t) + Lf(x_N) of given (x,u) sequences, which need not satisfy the dynamics
template <int MODEL>
__global__ void total_cost_kernel(const quattro_model_params p, const float* __restrict__ x,
                                  const float* __restrict__ u, int B, int N, double* __restrict__ cost) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* xb = x + (size_t)b * (N + 1) * NX;
  const float* ub = u + (size_t)b * N * NU;
  double J = 0.0;
  float xt[NX], ut[NU];
  for (int t = 0; t < N; ++t) {
#pragma unroll
    for (int i = 0; i < NX; ++i) xt[i] = xb[t * NX + i];
#pragma unroll
    for (int a = 0; a < NU; ++a) ut[a] = ub[t * NU + a];
    J += (double)qt_stage_cost<MODEL>(p, xt, ut);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) xt[i] = xb[N * NX + i];
  J += (double)qt_final_cost<MODEL>(p, xt);
  cost[b] = J;
}

// costs (and optionally trajectories) of every (trajectory, alpha): thread = b * 8 + alpha slot
template <int MODEL>
__global__ void rollout_kernel(const quattro_model_params p, const float* __restrict__ x_nom,
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
    J = rollout_closed<MODEL, true>(p, xn, un, Kb, kb, al.a[ai], N, xo, uo);
  } else {
    J = rollout_closed<MODEL, false>(p, xn, un, Kb, kb, al.a[ai], N, nullptr, nullptr);
  }
  cost[(size_t)ai * B + b] = J;
}

// fused line search: 8 lanes per trajectory
template <int MODEL>
__global__ void linesearch_kernel(const quattro_model_params p, float* x_nom, float* u_nom,
                                  const float* __restrict__ K, const float* __restrict__ k, AlphaList al, int n_alpha,
                                  int B, int N, double tol, double* cost, int32_t* __restrict__ alpha_idx,
                                  int32_t* active, int32_t* iters) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3, ai = gid & 7;
  const bool live = (b < B) && (active == nullptr || active[b] != 0);
  const bool mine = live && ai < n_alpha;
  float* xn = x_nom + (size_t)(live ? b : 0) * (N + 1) * NX;
  float* un = u_nom + (size_t)(live ? b : 0) * N * NU;
  const float* Kb = K + (size_t)(live ? b : 0) * N * NU * NX;
  const float* kb = k + (size_t)(live ? b : 0) * N * NU;
  const double J0 = live ? cost[b] : 0.0;
  double J = 0.0;
  bool ok = false;
  if (mine) {
    J = rollout_closed<MODEL, false>(p, xn, un, Kb, kb, al.a[ai], N, nullptr, nullptr);
    ok = (J <= J0);   // false for NaN, like the reference's comparison
  }
  // first accepted alpha inside this trajectory's 8-lane group
  const unsigned long long bal = __ballot(ok);
  const int lane = threadIdx.x & 63;
  const unsigned grp = (unsigned)((bal >> (lane & ~7)) & 0xffull);
  const int first = grp ? (__ffs((int)grp) - 1) : -1;
  if (mine && ai == first) {
    rollout_closed<MODEL, true>(p, xn, un, Kb, kb, al.a[ai], N, xn, un);
    cost[b] = J;
    if (active != nullptr && fabs(J0 - J) < tol) active[b] = 0;   // converged
  }
  if (live && ai == 0) {
    if (alpha_idx != nullptr) alpha_idx[b] = first;
    if (iters != nullptr) iters[b] += 1;
    if (first < 0 && active != nullptr) active[b] = 0;            // no improving step
  }
}

}  // namespace

#define QT_DISPATCH_MODEL(p, CALL)                         \
  if ((p).model_id == QUATTRO_MODEL_CARTPOLE) {            \
    constexpr int MODEL = QUATTRO_MODEL_CARTPOLE;          \
    CALL;                                                  \
  } else if ((p).model_id == QUATTRO_MODEL_QUADROTOR) {    \
    constexpr int MODEL = QUATTRO_MODEL_QUADROTOR;         \
    CALL;                                                  \
  } else {                                                 \
    return QUATTRO_ERR_UNSUPPORTED;                        \
  }

int quattro_launch_simulate(const quattro_model_params& p, const float* x0, const float* u, int B, int N, float* x,
                            double* cost, hipStream_t stream) {
  const int threads = 64;
  QT_DISPATCH_MODEL(p, hipLaunchKernelGGL((simulate_kernel<MODEL>), dim3((B + threads - 1) / threads), dim3(threads),
                                          0, stream, p, x0, u, B, N, x, cost));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_total_cost(const quattro_model_params& p, const float* x, const float* u, int B, int N,
                              double* cost, hipStream_t stream) {
  const int threads = 64;
  QT_DISPATCH_MODEL(p, hipLaunchKernelGGL((total_cost_kernel<MODEL>), dim3((B + threads - 1) / threads),
                                          dim3(threads), 0, stream, p, x, u, B, N, cost));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_rollout(const quattro_model_params& p, const float* x_nom, const float* u_nom, const float* K,
                           const float* k, const float* alphas, int n_alpha, int B, int N, float* x_new, float* u_new,
                           double* cost, const int32_t* active, hipStream_t stream) {
  AlphaList al;
  for (int i = 0; i < QUATTRO_MAX_ALPHAS; ++i) al.a[i] = i < n_alpha ? alphas[i] : 0.0f;
  const int threads = 64;
  const long long tot = (long long)B * 8;
  QT_DISPATCH_MODEL(p, hipLaunchKernelGGL((rollout_kernel<MODEL>), dim3((unsigned)((tot + threads - 1) / threads)),
                                          dim3(threads), 0, stream, p, x_nom, u_nom, K, k, al, n_alpha, B, N, x_new,
                                          u_new, cost, active));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_linesearch(const quattro_model_params& p, float* x_nom, float* u_nom, const float* K,
                              const float* k, const float* alphas, int n_alpha, int B, int N, double tol, double* cost,
                              int32_t* alpha_idx, int32_t* active, int32_t* iters, hipStream_t stream) {
  AlphaList al;
  for (int i = 0; i < QUATTRO_MAX_ALPHAS; ++i) al.a[i] = i < n_alpha ? alphas[i] : 0.0f;
  const int threads = 64;
  const long long tot = (long long)B * 8;
  QT_DISPATCH_MODEL(p, hipLaunchKernelGGL((linesearch_kernel<MODEL>), dim3((unsigned)((tot + threads - 1) / threads)),
                                          dim3(threads), 0, stream, p, x_nom, u_nom, K, k, al, n_alpha, B, N, tol, cost,
                                          alpha_idx, active, iters));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
