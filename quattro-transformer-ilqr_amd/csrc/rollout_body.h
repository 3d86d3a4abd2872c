// Device-side bodies of the one-lane-per-candidate rollouts (cart-pole), shared by rollout.hip (one launch per call) and
// solve_cartpole.hip (the device-resident solve loop).  Design notes: rollout.hip.
#pragma once
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

// open-loop rollout + cost of trajectory b by ONE lane (the caller masks lanes without a trajectory)
template <int MODEL, bool RK4>
__device__ __forceinline__ void simulate_body(const quattro_model_params& p, const float* __restrict__ x0,
                                              const float* __restrict__ u, int N, float* __restrict__ x,
                                              double* __restrict__ cost, const int b) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
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

// Fused line search: LPT consecutive lanes per trajectory (8 in the stand-alone kernel, 16 inside the cart-pole's device-resident
// loop, whose sweep owns a 16-lane row per trajectory, 64 inside a user model's, whose sweep owns the wave), lane ai < n_alpha <= 8 of them rolls candidate ai out.  Every candidate
// leaves its (x', u') in the scratch; after the ballot the trajectory's LPT lanes copy the accepted candidate over the nominal.
// `gid` = LPT * trajectory + lane-in-trajectory; `force` treats every trajectory as active whatever its flag says.
template <int MODEL, bool RK4, int LPT>
__device__ __forceinline__ void linesearch_body(const quattro_model_params& p, float* x_nom, float* u_nom,
                                                const float* __restrict__ K, const float* __restrict__ k,
                                                const AlphaList& al, int n_alpha, int B, int N, double tol, double* cost,
                                                int32_t* __restrict__ alpha_idx, int32_t* active, int32_t* iters,
                                                float* __restrict__ scratch, const int gid, const bool force) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU, CS = cand_stride<MODEL>();
  static_assert(LPT == 8 || LPT == 16 || LPT == 64, "8, 16 or 64 lanes per trajectory");
  const int b = gid / LPT, ai = gid % LPT;
  const bool live = (b < B) && (force || active == nullptr || active[b < B ? b : 0] != 0);
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
    J = rollout_closed<MODEL, RK4>(p, xn, un, Kb, kb, al.a[ai & 7], N, ScratchStore<MODEL>{sc + (size_t)ai * N * CS});
    ok = (J <= J0);   // false for NaN, like the reference's comparison
  }
  // first accepted alpha inside this trajectory's 8-lane group
  const unsigned long long bal = __ballot(ok);
  const int lane = threadIdx.x & 63;
  const unsigned grp = (unsigned)((bal >> (lane & ~(LPT - 1))) & 0xffull);      // (candidates sit in the group's first 8 lanes)
  const int first = grp ? (__ffs((int)grp) - 1) : -1;
  if (live && first >= 0) {
    // The candidate records were stored by OTHER lanes of this wave.  For a workgroup-scope release gfx950 emits
    // s_waitcnt lgkmcnt(0) only (the CU is assumed to perform its vector-memory operations in order); the explicit vmcnt(0)
    // makes the hand-over independent of that assumption: every scratch store of this wave is complete (written through to
    // L2) before any lane loads a record (one wait per line search).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const float* src = sc + (size_t)first * N * CS;
    // a block of records per lane in flight: every load of the block is issued before its first store (one round trip to L2 per
    // block instead of one per record)
    constexpr int COPY_U = CS <= 8 ? 8 : 4;
    for (int t0 = ai; t0 < N; t0 += LPT * COPY_U) {
      float rec[COPY_U][CS];
#pragma unroll
      for (int q = 0; q < COPY_U; ++q) {
        const int t = t0 + LPT * q;
        load_vec<CS>(src + (size_t)(t < N ? t : t0) * CS, rec[q]);
      }
#pragma unroll
      for (int q = 0; q < COPY_U; ++q) {
        const int t = t0 + LPT * q;
        if (t < N) {
          store_vec<NX>(xn + (size_t)(t + 1) * NX, rec[q]);
          store_vec<NU>(un + (size_t)t * NU, rec[q] + NX);
        }
      }
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
