// Device-resident iLQR solve loop and receding-horizon (MPC) loop for the cart-pole: ONE launch runs, for every trajectory, the
// whole `while` loop of iLQR_TF.optimize (quattro_ilqr_tf/quattro_ilqr_tf.py:428-472) and, in MPC mode, the caller's loop
// around it (examples/cartpole/cartpole_mpc.py:326-332: solve, apply u_0, shift the warm start), with no host involvement.
// Same idea as solve_quad.hip (trajectories are independent: no grid-wide dependency anywhere), smaller grain: a trajectory
// owns a 16-lane DPP row of a wave — what the sweep needs (cartpole_body.h) — so a wave carries four trajectories and the
// whole loop is WAVE-private: no workgroup barrier at all, the phases hand their data over through global memory in program
// order of one wave.
//   nominal rollout : lane 0 of the row                               (rollout_body.h: simulate_body)
//   sweep           : the row's 16 lanes                              (cartpole_body.h: sweep16_cartpole_body)
//   line search     : lanes 0..5 of the row roll the candidates out, all 16 copy the accepted one
//                                                                     (rollout_body.h: linesearch_body<.., 16>)
// BASELINE configs[1] (B = 1024) is 256 such waves on 1024 SIMDs: every wave runs alone, an iteration costs the sum of its
// chains and nothing else — no launch, no kernel boundary, no host call.
#include "cartpole_body.h"
#include "rollout_body.h"
#include "solve_log.h"

namespace {

struct CpSolveArgs {
  quattro_model_params p;
  const float* x0;      // [B][4]  (MPC: the controllers' current states, == x_cur)
  float* x;             // [B][N+1][4]
  float* u;             // [B][N][1]
  float* K;             // [B][N][1][4]
  float* k;             // [B][N][1]
  double* cost;
  int32_t* alpha_idx;
  int32_t* active;
  int32_t* iters;
  int32_t* status;      // may be NULL
  float* scratch;
  AlphaList al;
  int n_alpha, B, N, max_iter, flags;
  float reg;
  double tol;
  int n_ctrl;
  float* x_cur;
  float* traj_x;              // [B][n_ctrl+1][4]
  float* traj_u;              // [B][n_ctrl][1]
  int32_t* traj_iters;        // [B][n_ctrl]
  const float* disturbance;   // [n_ctrl][B][4] or NULL
  SolveLogDev log;            // per-iteration log ring (rec == nullptr: none); plain solves only (n_ctrl == 0)
};

constexpr int CP_FLAG_SIMULATE = 1, CP_FLAG_FIXED = 2, CP_FLAG_RESET = 4;

// every store of this wave has completed before its lanes read what other lanes of the wave wrote (the phases of the loop hand
// trajectories over through global memory)
__device__ __forceinline__ void wave_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool RK4>
__global__ __launch_bounds__(QT_WAVE) void solve_cartpole_kernel(const CpSolveArgs a) {
  constexpr int MODEL = QUATTRO_MODEL_CARTPOLE, NX = 4;
  __shared__ __attribute__((aligned(16))) float s_stage[4 * cp16::STAGE_FLOATS];
  const int lane = threadIdx.x;
  const int sub = lane & 15;
  const int b = blockIdx.x * 4 + (lane >> 4);
  const bool have = b < a.B;
  const size_t bb = have ? b : 0;
  const bool force = (a.flags & CP_FLAG_FIXED) != 0;
  const int N = a.N;
  float* stage = s_stage + (lane >> 4) * cp16::STAGE_FLOATS;
  const int n_ctrl = a.n_ctrl > 0 ? a.n_ctrl : 1;
  for (int cs = 0; cs < n_ctrl; ++cs) {
    if ((a.flags & (CP_FLAG_SIMULATE | CP_FLAG_RESET)) != 0 || a.n_ctrl > 0) {
      if (have && sub == 0) {
        if (a.n_ctrl > 0 && cs == 0) {
#pragma unroll
          for (int i = 0; i < NX; ++i) a.traj_x[(bb * (a.n_ctrl + 1)) * NX + i] = a.x0[bb * NX + i];
        }
        if (a.n_ctrl > 0 || (a.flags & CP_FLAG_RESET) != 0) {
          a.iters[bb] = 0;          // per-solve state of this control step (what a host caller resets before a solve)
          a.active[bb] = 1;
          a.alpha_idx[bb] = -1;
          if (a.status != nullptr) a.status[bb] = 0;
        }
        if ((a.flags & CP_FLAG_SIMULATE) != 0 || a.n_ctrl > 0) simulate_body<MODEL, RK4>(a.p, a.x0, a.u, N, a.x, a.cost, b);
      }
      wave_handoff();
    }
    const bool logging = a.log.rec != nullptr && a.n_ctrl == 0;
    for (int it = 0; it < a.max_iter; ++it) {
      const bool act = have && (force || a.active[bb] != 0);
      if (!__any(act)) break;
      int log_it = 0;
      if (logging && act) {      // the record of this iteration: nominal, cost, start stamp (the row's 16 lanes)
        log_it = a.iters[bb];
        log_begin(a.log, b, log_it, a.x + bb * (N + 1) * NX, a.u + bb * N, a.cost[bb], sub, 16);
      }
      sweep16_cartpole_body<RK4>(a.p, a.x, a.u, N, 0, a.reg, a.K, a.k, a.status, b, act, lane, stage);
      if (logging && act && sub == 0) log_stamp(a.log, b, log_it, 1, 2);
      wave_handoff();
      linesearch_body<MODEL, RK4, 16>(a.p, a.x, a.u, a.K, a.k, a.al, a.n_alpha, a.B, N, a.tol, a.cost, a.alpha_idx, a.active,
                                      a.iters, a.scratch, 16 * b + sub, force);
      wave_handoff();
      if (logging && act)        // gains, accepted step, cost after the iteration, end stamp
        log_end(a.log, b, log_it, a.K + bb * N * NX, a.k + bb * N, a.alpha_idx[bb], a.cost[bb], sub, 16);
    }
    if (a.n_ctrl > 0) {
      // apply u_0 to the plant (the device model itself), record, shift the warm start: CartPoleMPC._ilqr_step after
      // optimize() (cartpole_mpc.py:331) and the simulator's step around it
      float u_next[4];
      const int tot = N - 1;                                   // u <- (u_1 .. u_{N-1}, u_{N-1}); 16 lanes x 4 cover N <= 65
      for (int base = 0; base < tot || base == 0; base += 64) {     // (at least once: the plant step below rides on the first pass, also when N = 1 and nothing shifts)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = base + sub + 16 * q;
          u_next[q] = (have && e < tot) ? a.u[bb * N + e + 1] : 0.0f;
        }
        float u0 = 0.0f;
        if (base == 0 && have && sub == 0) u0 = a.u[bb * N];
        wave_handoff();                                        // every element is read before any is written
        if (base == 0 && have && sub == 0) {
          float xo[NX], xn[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) xo[i] = a.x_cur[bb * NX + i];
          const float us[1] = {u0};
          qt_step<MODEL, RK4>(a.p, xo, us, xn);
          if (a.disturbance != nullptr) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] += a.disturbance[((size_t)cs * a.B + bb) * NX + i];
          }
          a.traj_u[bb * a.n_ctrl + cs] = u0;
          a.traj_iters[bb * a.n_ctrl + cs] = a.iters[bb];
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            a.x_cur[bb * NX + i] = xn[i];
            a.traj_x[(bb * (a.n_ctrl + 1) + cs + 1) * NX + i] = xn[i];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = base + sub + 16 * q;
          if (have && e < tot) a.u[bb * N + e] = u_next[q];
        }
      }
      wave_handoff();
    }
  }
}

}  // namespace

int quattro_launch_solve_cartpole(const quattro_model_params& p, const float* x0, float* x, float* u, int B, int N, float reg,
                                  const float* alphas, int n_alpha, double tol, int max_iter, int flags, float* K, float* k,
                                  double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status,
                                  float* scratch, int n_ctrl, float* x_cur, float* traj_x, float* traj_u, int32_t* traj_iters,
                                  const float* disturbance, const quattro_solve_log* log, hipStream_t stream) {
  CpSolveArgs a;
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
  a.log = make_log_dev(n_ctrl > 0 ? nullptr : log, 4, 1, N);
  const dim3 grid((unsigned)((B + 3) / 4));
  if (p.integrator == QUATTRO_INTEGRATOR_EULER)
    hipLaunchKernelGGL((solve_cartpole_kernel<false>), grid, dim3(QT_WAVE), 0, stream, a);
  else if (p.integrator == QUATTRO_INTEGRATOR_RK4)
    hipLaunchKernelGGL((solve_cartpole_kernel<true>), grid, dim3(QT_WAVE), 0, stream, a);
  else
    return QUATTRO_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
