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
#include "models_device.h"

namespace {

struct AlphaList {
  float a[QUATTRO_MAX_ALPHAS];
};

template <int N>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float* dst) {
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const float4 v = reinterpret_cast<const float4*>(p)[i];
      dst[4 * i + 0] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) dst[i] = p[i];
  }
}

template <int N>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float* src) {
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i)
      reinterpret_cast<float4*>(p)[i] = make_float4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = src[i];
  }
}

// nominal data of one step: x_t, u_t and the gains K_t, k_t
template <int MODEL>
struct NomStep {
  static constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  float x[NX], u[NU], k[NU], K[NU * NX];
  __device__ __forceinline__ void load(const float* __restrict__ xnom, const float* __restrict__ unom,
                                       const float* __restrict__ Kb, const float* __restrict__ kb, int t) {
    load_vec<NX>(xnom + (size_t)t * NX, x);
    load_vec<NU>(unom + (size_t)t * NU, u);
    load_vec<NU>(kb + (size_t)t * NU, k);
    load_vec<NU * NX>(Kb + (size_t)t * NU * NX, K);
  }
};

// candidate record stride in the line-search scratch: x'_{t+1} (NX) then u'_t (NU), padded to 16 bytes
template <int MODEL>
constexpr int cand_stride() {
  return (ModelDims<MODEL>::NX + ModelDims<MODEL>::NU + 3) / 4 * 4;
}

struct NoStore {
  __device__ __forceinline__ void operator()(int, const float*, const float*) const {}
};
template <int MODEL>
struct ArrayStore {   // separate output arrays (they do not alias the nominal)
  float* xo;
  float* uo;
  __device__ __forceinline__ void operator()(int t, const float* uh, const float* xnext) const {
    store_vec<ModelDims<MODEL>::NU>(uo + (size_t)t * ModelDims<MODEL>::NU, uh);
    store_vec<ModelDims<MODEL>::NX>(xo + (size_t)(t + 1) * ModelDims<MODEL>::NX, xnext);
  }
};
template <int MODEL>
struct ScratchStore {  // packed candidate records [t][cand_stride]
  float* s;
  __device__ __forceinline__ void operator()(int t, const float* uh, const float* xnext) const {
    constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU, CS = cand_stride<MODEL>();
    float rec[CS];
#pragma unroll
    for (int i = 0; i < NX; ++i) rec[i] = xnext[i];
#pragma unroll
    for (int a = 0; a < NU; ++a) rec[NX + a] = uh[a];
#pragma unroll
    for (int i = NX + NU; i < CS; ++i) rec[i] = 0.0f;
    store_vec<CS>(s + (size_t)t * CS, rec);
  }
};

// One closed-loop rollout: u'_t = u_t + alpha (k_t + K_t (x'_t - x_t)), x'_{t+1} = f(x'_t, u'_t), x'_0 = x_0.
// Returns sum_t L(x'_t, u'_t) + Lf(x'_N).
template <int MODEL, bool RK4, class Store>
__device__ __forceinline__ double rollout_closed(const quattro_model_params& p, const float* __restrict__ xnom,
                                                 const float* __restrict__ unom, const float* __restrict__ Kb,
                                                 const float* __restrict__ kb, float alpha, int N, Store store) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  NomStep<MODEL> b0, b1;
  b0.load(xnom, unom, Kb, kb, 0);
  b1.load(xnom, unom, Kb, kb, N > 1 ? 1 : 0);
  float xh[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xh[i] = b0.x[i];
  double J = 0.0;
  auto step = [&](const NomStep<MODEL>& nb, int t) __attribute__((always_inline)) {
    float uh[NU];
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      float du = nb.k[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) du = fmaf(nb.K[a * NX + i], xh[i] - nb.x[i], du);
      uh[a] = fmaf(alpha, du, nb.u[a]);
    }
    J += (double)qt_stage_cost<MODEL>(p, xh, uh);
    float xnext[NX];
    qt_step<MODEL, RK4>(p, xh, uh, xnext);
    store(t, uh, xnext);
#pragma unroll
    for (int i = 0; i < NX; ++i) xh[i] = xnext[i];
  };
  int t = 0;
  for (; t + 1 < N; t += 2) {
    step(b0, t);
    b0.load(xnom, unom, Kb, kb, t + 2 < N ? t + 2 : N - 1);
    step(b1, t + 1);
    b1.load(xnom, unom, Kb, kb, t + 3 < N ? t + 3 : N - 1);
  }
  if (t < N) step(b0, t);
  J += (double)qt_final_cost<MODEL>(p, xh);
  return J;
}

template <int MODEL, bool RK4>
__global__ __launch_bounds__(64) void simulate_kernel(const quattro_model_params p, const float* __restrict__ x0,
                                const float* __restrict__ u, int B, int N, float* __restrict__ x,
                                double* __restrict__ cost) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float xh[NX];
  load_vec<NX>(x0 + (size_t)b * NX, xh);
  float* xo = x + (size_t)b * (N + 1) * NX;
  const float* ub = u + (size_t)b * N * NU;
  store_vec<NX>(xo, xh);
  double J = 0.0;
  float u0[NU], u1[NU];
  load_vec<NU>(ub, u0);
  load_vec<NU>(ub + (size_t)(N > 1 ? 1 : 0) * NU, u1);
  auto step = [&](const float* ut, int t) __attribute__((always_inline)) {
    float xn[NX];
    J += (double)qt_stage_cost<MODEL>(p, xh, ut);
    qt_step<MODEL, RK4>(p, xh, ut, xn);
    store_vec<NX>(xo + (size_t)(t + 1) * NX, xn);
#pragma unroll
    for (int i = 0; i < NX; ++i) xh[i] = xn[i];
  };
  int t = 0;
  for (; t + 1 < N; t += 2) {
    step(u0, t);
    load_vec<NU>(ub + (size_t)(t + 2 < N ? t + 2 : N - 1) * NU, u0);
    step(u1, t + 1);
    load_vec<NU>(ub + (size_t)(t + 3 < N ? t + 3 : N - 1) * NU, u1);
  }
  if (t < N) step(u0, t);
  J += (double)qt_final_cost<MODEL>(p, xh);
  if (cost != nullptr) cost[b] = J;
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

// Fused line search: 8 lanes per trajectory.  Every candidate rollout leaves its (x', u') in the scratch; after the
// ballot the group's 8 lanes copy the accepted candidate over the nominal.
template <int MODEL, bool RK4>
__global__ __launch_bounds__(64) void linesearch_kernel(const quattro_model_params p, float* x_nom, float* u_nom,
                                  const float* __restrict__ K, const float* __restrict__ k, AlphaList al, int n_alpha,
                                  int B, int N, double tol, double* cost, int32_t* __restrict__ alpha_idx,
                                  int32_t* active, int32_t* iters, float* __restrict__ scratch) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU, CS = cand_stride<MODEL>();
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 3, ai = gid & 7;
  const bool live = (b < B) && (active == nullptr || active[b] != 0);
  const bool mine = live && ai < n_alpha;
  const size_t bb = live ? b : 0;
  float* xn = x_nom + bb * (N + 1) * NX;
  float* un = u_nom + bb * N * NU;
  const float* Kb = K + bb * N * NU * NX;
  const float* kb = k + bb * N * NU;
  float* sc = scratch + (bb * 8) * (size_t)N * CS;     // this trajectory's 8 candidate slots
  const double J0 = live ? cost[b] : 0.0;
  double J = 0.0;
  bool ok = false;
  if (mine) {
    J = rollout_closed<MODEL, RK4>(p, xn, un, Kb, kb, al.a[ai], N, ScratchStore<MODEL>{sc + (size_t)ai * N * CS});
    ok = (J <= J0);   // false for NaN, like the reference's comparison
  }
  // first accepted alpha inside this trajectory's 8-lane group
  const unsigned long long bal = __ballot(ok);
  const int lane = threadIdx.x & 63;
  const unsigned grp = (unsigned)((bal >> (lane & ~7)) & 0xffull);
  const int first = grp ? (__ffs((int)grp) - 1) : -1;
  if (live && first >= 0) {
    // all 8 lanes of the group copy the accepted candidate.  Writer and readers are lanes of ONE wave (same CU, same
    // L1): a workgroup-scope release/acquire pair (the stores are waited for before the loads issue) is sufficient.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const float* src = sc + (size_t)first * N * CS;
    for (int t = ai; t < N; t += 8) {
      float rec[CS];
      load_vec<CS>(src + (size_t)t * CS, rec);
      store_vec<NX>(xn + (size_t)(t + 1) * NX, rec);
      store_vec<NU>(un + (size_t)t * NU, rec + NX);
    }
    if (ai == first) {
      cost[b] = J;
      if (active != nullptr && fabs(J0 - J) < tol) active[b] = 0;   // converged
    }
  }
  if (live && ai == 0) {
    if (alpha_idx != nullptr) alpha_idx[b] = first;
    if (iters != nullptr) iters[b] += 1;
    if (first < 0 && active != nullptr) active[b] = 0;            // no improving step
  }
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
#define QT_DISPATCH_MODEL(p, ...)                          \
  if ((p).model_id == QUATTRO_MODEL_CARTPOLE) {            \
    constexpr int MODEL = QUATTRO_MODEL_CARTPOLE;          \
    QT_DISPATCH_INTEG(p, __VA_ARGS__);                     \
  } else {                                                 \
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
