// Device-resident iLQR solve loop and receding-horizon (MPC) loop for a USER-COMPILED model (user_model.h): ONE launch runs,
// for every trajectory, the whole `while` loop of iLQR_TF.optimize (quattro_ilqr_tf/quattro_ilqr_tf.py:428-472) and, in MPC
// mode, the caller's loop around it (the shape of examples/quadrotor/quadrotor_mpc.py:102-124: solve, apply u_0, shift the warm
// start), with no host involvement.  Only compiled into a user model's library (-DQT_USER_MODEL_HEADER=...).
//
// Same idea as solve_quad.hip / solve_cartpole.hip with the generic device bodies: a trajectory owns ONE wavefront (a
// workgroup of its own), trajectories are independent, so a wave leaves its loop as soon as its trajectory stops — no grid-wide
// dependency.  The phases hand their data over through global memory in the program order of that wave:
//   nominal rollout : lane 0                                              (rollout_body.h: simulate_body)
//   linearisation   : LPI lanes per step, 64 / LPI steps at a time        (user_linearize.h: forward-mode duals) -> ROWMAJOR records
//   terminal pair   : lanes 0..n-1                                        (user_linearize.h)
//   sweep           : the 64 lanes                                        (sweep_generic_body.h: pivoting, the reference's formulas)
//   line search     : lanes 0..5 roll the candidates out, all 64 copy the accepted one   (rollout_body.h: linesearch_body<.., 64>)
// Bit-identical to the host-driven loop of the same library (all of a user library's translation units are compiled with
// -ffp-contract=off so that the shared device functions round alike wherever they are inlined).
#ifndef QT_USER_MODEL_HEADER
#error "solve_user.hip is part of a user model's library"
#endif
#include "rollout_body.h"
#include "solve_log.h"
#include "sweep_generic_body.h"
#include "sweep_tile16_body.h"
#include "user_linearize.h"

namespace {

struct UserSolveArgs {
  quattro_model_params p;
  const float* x0;      // [B][n]  (MPC: the controllers' current states, == x_cur)
  float* x;             // [B][N+1][n]
  float* u;             // [B][N][m]
  float* K;             // [B][N][m][n]
  float* k;             // [B][N][m]
  double* cost;
  int32_t* alpha_idx;
  int32_t* active;
  int32_t* iters;
  int32_t* status;      // may be NULL
  float* rec;           // [B][N][RowMajorRec stride]
  float* VxN;           // [B][n]
  float* VxxN;          // [B][n][n]
  float* scratch;       // line-search candidates
  AlphaList al;
  int n_alpha, B, N, max_iter, flags;
  float reg;
  double tol;
  int n_ctrl;
  float* x_cur;
  float* traj_x;              // [B][n_ctrl+1][n]
  float* traj_u;              // [B][n_ctrl][m]
  int32_t* traj_iters;        // [B][n_ctrl]
  const float* disturbance;   // [n_ctrl][B][n] or NULL
  SolveLogDev log;            // per-iteration log ring (rec == nullptr: none); plain solves only (n_ctrl == 0)
};

constexpr int US_FLAG_SIMULATE = 1, US_FLAG_FIXED = 2, US_FLAG_RESET = 4;

// every store of this wave has completed before its lanes read what other lanes of the wave wrote
__device__ __forceinline__ void wave_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// (two waves per SIMD: left to itself the allocator takes 300 registers for the dual-number linearisation next to the tile sweep —
//  one wave per SIMD — and the loop runs 3x slower than with the 42 spilled registers this bound costs)
template <bool RK4>
__global__ __launch_bounds__(QT_WAVE, 2) void solve_user_kernel(const UserSolveArgs a) {
  constexpr int MODEL = QUATTRO_MODEL_USER, NX = QT_USER_NX, NU = QT_USER_NU, NZ = NX + NU;
  constexpr int LPI = NZ <= 8 ? 8 : (NZ <= 16 ? 16 : 32), IPP = QT_WAVE / LPI;      // lanes per item, items per pass
  using R = RowMajorRec<NX, NU>;
  // the sweep quattro_model_layout promises for this model, so that this loop and the host-driven one agree bit for bit:
  // ROWMAJOR_TILE (the MFMA tile recursion on the same records, padded inside the kernel) where the problem fits a tile
  constexpr bool TILE = NX <= 12 && NU <= 4 && NX + NU >= 6;
  __shared__ __attribute__((aligned(16))) float s_t[TILE ? 16 * LD : 4];
  __shared__ __attribute__((aligned(16))) float s_vx[64];
  __shared__ __attribute__((aligned(16))) float s_lin[4];
  const int lane = threadIdx.x;
  const int b = blockIdx.x;                      // (grid = B exactly)
  const size_t bb = b;
  const bool force = (a.flags & US_FLAG_FIXED) != 0;
  const int N = a.N;
  float* xb = a.x + bb * (N + 1) * NX;
  float* ub = a.u + bb * N * NU;
  float* recb = a.rec + bb * N * R::STRIDE;
  volatile int32_t* act_flag = a.active + b;     // written by this wave's line search: always re-read from memory
  const int n_ctrl = a.n_ctrl > 0 ? a.n_ctrl : 1;
  for (int cs = 0; cs < n_ctrl; ++cs) {
    if ((a.flags & (US_FLAG_SIMULATE | US_FLAG_RESET)) != 0 || a.n_ctrl > 0) {
      if (lane == 0) {
        if (a.n_ctrl > 0 && cs == 0) {
#pragma unroll
          for (int i = 0; i < NX; ++i) a.traj_x[(bb * (a.n_ctrl + 1)) * NX + i] = a.x0[bb * NX + i];
        }
        if (a.n_ctrl > 0 || (a.flags & US_FLAG_RESET) != 0) {
          a.iters[b] = 0;           // per-solve state of this control step (what a host caller resets before a solve)
          a.active[b] = 1;
          a.alpha_idx[b] = -1;
          if (a.status != nullptr) a.status[b] = 0;
        }
        if ((a.flags & US_FLAG_SIMULATE) != 0 || a.n_ctrl > 0) simulate_body<MODEL, RK4>(a.p, a.x0, a.u, N, a.x, a.cost, b);
      }
      wave_handoff();
    }
    const bool logging = a.log.rec != nullptr && a.n_ctrl == 0;
    for (int it = 0; it < a.max_iter; ++it) {
      if (!(force || *act_flag != 0)) break;       // wave-uniform: one trajectory per wave
      int log_it = 0;
      if (logging) {             // the record of this iteration: nominal, cost, start stamp
        log_it = *(volatile int32_t*)(a.iters + b);
        log_begin(a.log, b, log_it, xb, ub, *(volatile double*)(a.cost + b), lane, QT_WAVE);
      }
      // linearisation about the nominal: LPI lanes per step
      {
        const int j = lane % LPI;
        for (int t0 = 0; t0 < N; t0 += IPP) {
          const int t = t0 + lane / LPI;
          if (t < N && j < NZ)
            user_linearize_item<R, RK4>(a.p, xb + (size_t)t * NX, ub + (size_t)t * NU, recb + (size_t)t * R::STRIDE, j);
        }
        if (lane < NX) user_terminal_row(a.p, xb + (size_t)N * NX, lane, a.VxN + bb * NX, a.VxxN + bb * NX * NX);
      }
      wave_handoff();
      if constexpr (TILE) {
        FusedArgs fa;
        fa.B = a.B;
        fa.k_rows = 0;
        fa.rn = NX;
        fa.rm = NU;
        sweep_tile16_body<MODE_ROWPAD>(a.rec, a.VxN, a.VxxN, N, a.reg, a.K, a.k, a.status, fa, b, lane, s_t, s_vx, s_lin);
      } else {
        sweep_generic_body<NX, NU>(a.rec, a.VxN, a.VxxN, N, a.reg, a.K, a.k, a.status, b, lane);
      }
      if (logging && lane == 0) log_stamp(a.log, b, log_it, 1, 2);
      wave_handoff();
      linesearch_body<MODEL, RK4, 64>(a.p, a.x, a.u, a.K, a.k, a.al, a.n_alpha, a.B, N, a.tol, a.cost, a.alpha_idx, a.active,
                                      a.iters, a.scratch, 64 * b + lane, force);
      wave_handoff();
      if (logging)               // gains, accepted step, cost after the iteration, end stamp
        log_end(a.log, b, log_it, a.K + bb * N * NU * NX, a.k + bb * N * NU, *(volatile int32_t*)(a.alpha_idx + b),
                *(volatile double*)(a.cost + b), lane, QT_WAVE);
    }
    if (a.n_ctrl > 0) {
      // apply u_0 to the plant (the device model itself), record, shift the warm start u <- (u_1 .. u_{N-1}, u_{N-1})
      const int tot = (N - 1) * NU;                              // elements that move
      float u0[NU];
#pragma unroll
      for (int q = 0; q < NU; ++q) u0[q] = ub[q];
      for (int base = 0; base < tot; base += QT_WAVE) {
        const int e = base + lane;
        const float v = e < tot ? ub[e + NU] : 0.0f;
        wave_handoff();                                          // every element of the pass is read before any is written
        if (e < tot) ub[e] = v;
        wave_handoff();
      }
      if (lane == 0) {
        float xo[NX], xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xo[i] = a.x_cur[bb * NX + i];
        qt_step<MODEL, RK4>(a.p, xo, u0, xn);
        if (a.disturbance != nullptr) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xn[i] += a.disturbance[((size_t)cs * a.B + bb) * NX + i];
        }
#pragma unroll
        for (int q = 0; q < NU; ++q) a.traj_u[(bb * a.n_ctrl + cs) * NU + q] = u0[q];
        a.traj_iters[bb * a.n_ctrl + cs] = a.iters[b];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          a.x_cur[bb * NX + i] = xn[i];
          a.traj_x[(bb * (a.n_ctrl + 1) + cs + 1) * NX + i] = xn[i];
        }
      }
      wave_handoff();
    }
  }
}

}  // namespace

// The stand-alone tile sweep of THIS library (layout ROWMAJOR_TILE): the same body, compiled in this translation unit with this
// library's flags (no implicit fma contraction), so that the host-driven loop of a user model rounds exactly like its persistent
// kernel above.  (libquattro_hip.so has its own instance for foreign records, compiled with its flags.)
namespace {
__global__ __launch_bounds__(QT_WAVE) void sweep_rowpad_user_kernel(const float* __restrict__ rec, const float* __restrict__ VxN,
                                                                    const float* __restrict__ VxxN, int S, float reg,
                                                                    float* __restrict__ Kout, float* __restrict__ kout,
                                                                    int32_t* __restrict__ status,
                                                                    const int32_t* __restrict__ active, int B, int n, int m) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B || (active != nullptr && active[b] == 0)) return;
  __shared__ __attribute__((aligned(16))) float s_t[16 * LD];
  __shared__ __attribute__((aligned(16))) float s_vx[64];
  __shared__ __attribute__((aligned(16))) float s_lin[4];
  FusedArgs fa;
  fa.B = B;
  fa.k_rows = 0;
  fa.rn = n;
  fa.rm = m;
  sweep_tile16_body<MODE_ROWPAD>(rec, VxN, VxxN, S, reg, Kout, kout, status, fa, b, lane, s_t, s_vx, s_lin);
}
}  // namespace

int quattro_launch_sweep_rowpad_user(const float* rec, const float* VxN, const float* VxxN, int B, int S, int n, int m, float reg,
                                     float* K, float* k, int32_t* status, const int32_t* active, hipStream_t stream) {
  if (n < 1 || n > 12 || m < 1 || m > 4) return QUATTRO_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sweep_rowpad_user_kernel, dim3((unsigned)B), dim3(QT_WAVE), 0, stream, rec, VxN, VxxN, S, reg, K, k, status,
                     active, B, n, m);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_solve_user(const quattro_model_params& p, const float* x0, float* x, float* u, int B, int N, float reg,
                              const float* alphas, int n_alpha, double tol, int max_iter, int flags, float* K, float* k,
                              double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status, float* rec,
                              float* VxN, float* VxxN, float* scratch, int n_ctrl, float* x_cur, float* traj_x, float* traj_u,
                              int32_t* traj_iters, const float* disturbance, const quattro_solve_log* log, hipStream_t stream) {
  UserSolveArgs a;
  a.p = p;
  a.x0 = n_ctrl > 0 ? x_cur : x0;
  a.x = x;
  a.u = u;
  a.K = K;
  a.k = k;
  a.cost = cost;
  a.alpha_idx = alpha_idx;
  a.active = active;
  a.iters = iters;
  a.status = status;
  a.rec = rec;
  a.VxN = VxN;
  a.VxxN = VxxN;
  a.scratch = scratch;
  for (int i = 0; i < QUATTRO_MAX_ALPHAS; ++i) a.al.a[i] = i < n_alpha ? alphas[i] : 0.0f;
  a.n_alpha = n_alpha;
  a.B = B;
  a.N = N;
  a.max_iter = max_iter;
  a.flags = flags;
  a.reg = reg;
  a.tol = tol;
  a.n_ctrl = n_ctrl;
  a.x_cur = x_cur;
  a.traj_x = traj_x;
  a.traj_u = traj_u;
  a.traj_iters = traj_iters;
  a.disturbance = disturbance;
  a.log = make_log_dev(n_ctrl > 0 ? nullptr : log, QT_USER_NX, QT_USER_NU, N);
  const dim3 grid((unsigned)B);
  if (p.integrator == QUATTRO_INTEGRATOR_EULER)
    hipLaunchKernelGGL((solve_user_kernel<false>), grid, dim3(QT_WAVE), 0, stream, a);
  else if (p.integrator == QUATTRO_INTEGRATOR_RK4)
    hipLaunchKernelGGL((solve_user_kernel<true>), grid, dim3(QT_WAVE), 0, stream, a);
  else
    return QUATTRO_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
