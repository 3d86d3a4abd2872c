// Quadrotor rollouts, lane-cooperative: FOUR lanes (one DPP quad) advance one trajectory / line-search candidate.
//
// Same arithmetic as rollout.hip (reference quattro_ilqr_tf.py: simulate :127-132, compute_total_cost :138-143,
// forward_pass :377-390, alpha loop + stop test of optimize :433-451, :472), different mapping.  A rollout is a serial
// chain in t, and with one lane per candidate a B = 4096 line search is only 512 waves on 1024 SIMDs: every wave runs
// alone on its SIMD, issues one instruction (vector OR scalar) per 4-cycle slot, and a step of ~550 instructions costs
// 1.7 us.  Splitting a step over the 4 lanes of a quad cuts the per-wave instruction count ~4x and gives 4x the waves
// (2 per SIMD at B = 4096), so the chain gets shorter and the SIMDs fill up.
//
// Lane roles inside a quad (j = lane & 3):
//   axis lane a = j < 3 owns the four states of its axis: (p_a, v_a, angle_a, omega_a) = x[a], x[3+a], x[6+a], x[9+a];
//   lane 3 clones axis 0 for the state work (its cost weights are zero, it stores no state);
//   every lane j owns control j: u'_j = u_j + alpha (k_j + K_j,: dx) with the 12 dx read from the three axis lanes by
//   DPP quad_perm broadcasts, its R / barrier cost term, and the store of u'_j.
// Per step the quad exchanges: dx (inside the K dx FMAs), the 4 controls, sin/cos of the 3 angles (each computed once,
// by its axis lane) and the 3 body rates — all DPP quad_perm moves, no LDS.  The per-lane partial costs are accumulated
// in fp64 and summed over the quad once, at the end.
//
// Layouts: line search = 8 candidate quads per trajectory (32 lanes), 2 trajectories per wave; simulate = 16
// trajectories per wave.  Euler and RK4 (the rate function is the cooperative part; RK4 just calls it four times).
// Wave priority inside a rollout step (round 4): the state recurrence (dx, K dx, u, rate function, update) at priority 1, the stage
// cost, the record store and the next loads that hang off it at 0.  Two line-search waves share a SIMD, and vector issue is
// arbitrated by priority, then age: whichever wave is on its recurrence goes first.  Line search 36.7 -> 33.6 us, RK4 63.2 -> 54.7 us
// (A/B on one box; same instructions, same results).  The persistent kernel keeps its line-search wave at one constant priority
// (solve_quad.hip), where toggling inside the step bought nothing.
#define QT_ROLLOUT_PRIO 1
#include "rollout_quad_body.h"

// steps of nominal data a line-search quad keeps requested ahead (rollout_quad_body.h: quad_rollout_closed)
#ifndef QT_LS_PF
#define QT_LS_PF 4
#endif

namespace {

// ---------------------------------------------------------------------------------------------- kernels
// simulate: quad per trajectory, 16 trajectories per wave
template <bool RK4>
__global__ __launch_bounds__(64) void simulate_quad_kernel(const quattro_model_params p, const float* __restrict__ x0,
                                                           const float* __restrict__ u, int B, int N,
                                                           float* __restrict__ x, double* __restrict__ cost) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  simulate_quad_body<RK4>(p, x0, u, N, x, cost, gid, (gid >> 2) < B);
}

// costs (and optionally trajectories) of every (trajectory, alpha): 8 candidate quads per trajectory
template <bool RK4>
__global__ __launch_bounds__(64) void rollout_quad_kernel(const quattro_model_params p, const float* __restrict__ x_nom,
                                                          const float* __restrict__ u_nom, const float* __restrict__ K,
                                                          const float* __restrict__ k, AlphaList al, int n_alpha, int B,
                                                          int N, float* __restrict__ x_new, float* __restrict__ u_new,
                                                          double* __restrict__ cost, const int32_t* __restrict__ active) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 5, ai = (gid >> 2) & 7;
  const bool live = (b < B) && (active == nullptr || active[b < B ? b : 0] != 0);
  if (!__any(live)) return;
  const bool mine = live && ai < n_alpha;
  const size_t bb = live ? b : 0;
  const int aa = mine ? ai : 0;
  const LaneConst L = lane_const(p, gid & 3);
  const float* xn = x_nom + bb * (N + 1) * NX;
  const float* un = u_nom + bb * N * NU;
  const int wb = __builtin_amdgcn_readfirstlane(b);    // the wave's first trajectory (gid grows with the lane; wb < B: a lane is live)
  const NomSrc nom(L, x_nom, u_nom, K, k, N, wb, live ? b - wb : 0, 2);
  float alpha = al.a[0];
#pragma unroll
  for (int i = 1; i < QUATTRO_MAX_ALPHAS; ++i) alpha = (aa == i) ? al.a[i] : alpha;
  double J;
  if (x_new != nullptr && u_new != nullptr) {
    float* xo = x_new + ((size_t)aa * B + bb) * (N + 1) * NX;
    float* uo = u_new + ((size_t)aa * B + bb) * N * NU;
    if (mine && L.j < 3) {
#pragma unroll
      for (int g = 0; g < 4; ++g) xo[3 * g + L.a] = xn[3 * g + L.a];
    }
    J = quad_rollout_closed<RK4, 4>(p, L, nom, alpha, N, mine, ArrayStore(L, xo, uo, mine));
  } else {
    J = quad_rollout_closed<RK4, 4>(p, L, nom, alpha, N, mine, NoStore{});
  }
  J = quad_sum(J);
  if (mine && L.j == 0) cost[(size_t)ai * B + b] = J;
}

// Fused line search: 8 candidate quads = 32 lanes per trajectory, 2 trajectories per wave (linesearch_quad_body)
template <bool RK4>
__global__ __launch_bounds__(64) void linesearch_quad_kernel(const quattro_model_params p, float* x_nom, float* u_nom,
                                                             const float* __restrict__ K, const float* __restrict__ k,
                                                             AlphaList al, int n_alpha, int B, int N, double tol,
                                                             double* cost, int32_t* __restrict__ alpha_idx,
                                                             int32_t* active, int32_t* iters,
                                                             float* __restrict__ scratch) {
  linesearch_quad_body<RK4, QT_LS_PF>(p, x_nom, u_nom, K, k, al, n_alpha, B, N, tol, cost, alpha_idx, active, iters, scratch,
                               blockIdx.x * blockDim.x + threadIdx.x, false);
}

}  // namespace

#define QT_QUAD_DISPATCH(p, ...)                              \
  if ((p).integrator == QUATTRO_INTEGRATOR_EULER) {           \
    constexpr bool RK4 = false;                               \
    __VA_ARGS__;                                              \
  } else if ((p).integrator == QUATTRO_INTEGRATOR_RK4) {      \
    constexpr bool RK4 = true;                                \
    __VA_ARGS__;                                              \
  } else {                                                    \
    return QUATTRO_ERR_UNSUPPORTED;                           \
  }

int quattro_launch_simulate_quad(const quattro_model_params& p, const float* x0, const float* u, int B, int N, float* x,
                                 double* cost, hipStream_t stream) {
  const long long tot = (long long)B * 4;
  QT_QUAD_DISPATCH(p, hipLaunchKernelGGL((simulate_quad_kernel<RK4>), dim3((unsigned)((tot + 63) / 64)), dim3(64), 0,
                                         stream, p, x0, u, B, N, x, cost));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_rollout_quad(const quattro_model_params& p, const float* x_nom, const float* u_nom, const float* K,
                                const float* k, const float* alphas, int n_alpha, int B, int N, float* x_new,
                                float* u_new, double* cost, const int32_t* active, hipStream_t stream) {
  AlphaList al;
  for (int i = 0; i < QUATTRO_MAX_ALPHAS; ++i) al.a[i] = i < n_alpha ? alphas[i] : 0.0f;
  const long long tot = (long long)B * 32;
  QT_QUAD_DISPATCH(p, hipLaunchKernelGGL((rollout_quad_kernel<RK4>), dim3((unsigned)((tot + 63) / 64)), dim3(64), 0,
                                         stream, p, x_nom, u_nom, K, k, al, n_alpha, B, N, x_new, u_new, cost, active));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_linesearch_quad(const quattro_model_params& p, float* x_nom, float* u_nom, const float* K,
                                   const float* k, const float* alphas, int n_alpha, int B, int N, double tol,
                                   double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, float* scratch,
                                   hipStream_t stream) {
  AlphaList al;
  for (int i = 0; i < QUATTRO_MAX_ALPHAS; ++i) al.a[i] = i < n_alpha ? alphas[i] : 0.0f;
  const long long tot = (long long)B * 32;
  QT_QUAD_DISPATCH(p, hipLaunchKernelGGL((linesearch_quad_kernel<RK4>), dim3((unsigned)((tot + 63) / 64)), dim3(64), 0,
                                         stream, p, x_nom, u_nom, K, k, al, n_alpha, B, N, tol, cost, alpha_idx, active,
                                         iters, scratch));
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
