// Device-resident iLQR solve loop and receding-horizon (MPC) loop for the quadrotor: ONE persistent launch runs, for every
// trajectory, the whole `while` loop of iLQR_TF.optimize (quattro_ilqr_tf/quattro_ilqr_tf.py:428-472) — and, in MPC mode,
// the caller's loop around it (examples/quadrotor/quadrotor_mpc.py:102-124: solve, apply u_0, shift the warm start) — with
// no host involvement between iterations or control steps.
//
// Why it can be one launch: trajectories are independent optimisation problems.  Nothing an iteration of
// trajectory b reads was written for another trajectory, so there is no grid-wide dependency anywhere in a solve: a
// workgroup owns its trajectories from the first rollout to the last accepted step and simply leaves when they are done.
// The multi-launch path (quattro_ilqr_iterate_f32 once per iteration) pays, per iteration, two kernel boundaries, the
// drain of the slowest wave of each kernel, a host call, and — because the host cannot know when the last trajectory has
// converged without asking — either a synchronisation every few iterations or empty launches after convergence.
//
// Mapping (the SAME device code as the stand-alone kernels, sweep_tile16_body.h / rollout_quad_body.h, so every number is
// bit-identical to the multi-launch path): a workgroup = 2 wavefronts = 2 trajectories.
//   sweep phase       : wave w linearises and sweeps trajectory 2 blk + w (one wavefront per trajectory, 16x16 tiles,
//                       exact-fp32 MFMA), K / k go to global memory (they are outputs) and stay in the XCD's L2;
//   workgroup barrier
//   line-search phase : wave 0 runs the fused 6-alpha line search of BOTH trajectories (8 candidate quads = 32 lanes per
//                       trajectory), commits the accepted candidate, updates cost / alpha_idx / iters / active;
//   workgroup barrier ; loop until both trajectories have stopped or max_iter.
// Workgroups drift apart in phase, so a SIMD hosts a mix of MFMA-bound sweeps and VALU-bound rollouts instead of four
// copies of the same phase; a trajectory that needs 29 iterations when the mean is 15 runs its last ones on an almost
// empty chip, at the latency of a lone wave (no contention), and the launch ends when the last workgroup leaves.
// Registers: the line search prefetches its nominal data 2 steps ahead here instead of 4 (the gains were written by this
// workgroup a moment ago and come from L2), which keeps the whole loop inside the sweep's 128-register budget: 4 waves
// per SIMD, i.e. B = 4096 resident at once.
#include "rollout_quad_body.h"
#include "solve_log.h"

#ifndef QT_SOLVE_LS_PRIO
#define QT_SOLVE_LS_PRIO 3
#endif
#include "sweep_tile16_body.h"

namespace {

struct SolveArgs {
  FusedArgs fa;         // model parameters + nominal (x [B][N+1][12], u [B][N][4], in/out) as the sweep body takes them: kept
                        // inside the kernel-argument block (a private copy with lane-dependent indexing would live in scratch)
  const float* x0;      // [B][12]  states the rollouts start from (MPC: the controllers' current states, updated in place)
  float* x;             // == fa.x, writable
  float* u;             // == fa.u, writable
  float* K;             // [B][N][4][12]
  float* k;             // [B][N][4]
  double* cost;         // [B]
  int32_t* alpha_idx;   // [B]
  int32_t* active;      // [B]
  int32_t* iters;       // [B]
  int32_t* status;      // [B] (may be NULL)
  float* scratch;       // line-search candidates
  AlphaList al;
  int n_alpha, B, N, max_iter, flags;
  float reg;
  double tol;
  // receding-horizon mode (n_ctrl > 0)
  int n_ctrl;
  float* x_cur;               // [B][12]  == x0 (writable)
  float* traj_x;              // [B][n_ctrl+1][12]
  float* traj_u;              // [B][n_ctrl][4]
  int32_t* traj_iters;        // [B][n_ctrl]
  const float* disturbance;   // [n_ctrl][B][12] or NULL
  unsigned long long* stamps; // diagnostics (may be NULL): [workgroup][2 * (n_ctrl + 1) + 2] = (s_memrealtime at the start / after
                              // each control step, iterations the workgroup ran in that step; wave 1's exit stamp and loop
                              // passes in the last two slots); 100 MHz ticks
  int stamp_rows;             // workgroups the stamps buffer has rows for (others do not stamp)
  SolveLogDev log;            // per-iteration log ring (rec == nullptr: none); plain solves only (n_ctrl == 0)
};

constexpr int FLAG_SIMULATE = 1, FLAG_FIXED = 2, FLAG_RESET = 4;

// apply u_0 to the plant (the device model itself), record, shift the warm start, reset the per-solve state: what
// QuadrotorMPC.control_step does after optimize() (quadrotor_mpc.py:121-122) plus the simulator's step around it.
// Run by wave 0: quad q = lane >> 2 < 2 owns trajectory b0 + q for the plant step; the shift is spread over 32 lanes per
// trajectory.
template <bool RK4>
__device__ __forceinline__ void mpc_advance(const SolveArgs& a, const int b0, const int lane, const int cs) {
  const int q = lane >> 2, tb = b0 + q;
  const bool live = lane < 8 && tb < a.B;
  const size_t bb = live ? tb : 0;
  const LaneConst L = lane_const(a.fa.p, lane & 3);
  const int N = a.N;
  float xo[4], xn[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xo[g] = a.x_cur[bb * NX + 3 * g + L.a];
  const float u0 = a.u[bb * N * NU + L.j];
  const QuadU U(u0);
  quad_step<RK4>(L, xo, U, xn);
  if (a.disturbance != nullptr) {
#pragma unroll
    for (int g = 0; g < 4; ++g) xn[g] += a.disturbance[((size_t)cs * a.B + bb) * NX + 3 * g + L.a];
  }
  if (live) {
    a.traj_u[(bb * a.n_ctrl + cs) * NU + L.j] = u0;
    if (L.j < 3) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        a.x_cur[bb * NX + 3 * g + L.a] = xn[g];
        a.traj_x[(bb * (a.n_ctrl + 1) + cs + 1) * NX + 3 * g + L.a] = xn[g];
      }
    }
    if (L.j == 0) {
      a.traj_iters[bb * a.n_ctrl + cs] = a.iters[bb];
    }
  }
  // warm start u <- (u_1, ..., u_{N-1}, u_{N-1}): every element is read before any is written (the loads below are
  // complete — their values sit in registers — before the first store issues)
  const int half = lane >> 5, l32 = lane & 31, ts = b0 + half;
  if (ts < a.B) {
    float* ub = a.u + (size_t)ts * N * NU;
    const int tot = (N - 1) * NU;
    constexpr int MAXI = 8;    // 32 lanes x 8 = 256 >= (N - 1) * 4 for N <= 65; longer horizons loop
    for (int base = 0; base < tot; base += 32 * MAXI) {
      float v[MAXI];
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        const int e = base + l32 + 32 * i;
        v[i] = e < tot ? ub[e + NU] : 0.0f;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int i = 0; i < MAXI; ++i) {
        const int e = base + l32 + 32 * i;
        if (e < tot) ub[e] = v[i];
      }
    }
  }
}

// Every phase reads the kernel arguments through its OWN opaque copy of the kernel-argument pointer, taken inside the loop.
// With plain by-value access the compiler treats everything computed from the arguments as loop-invariant: the
// parameter-only quotients (arm / Ix, 1 / mass, ...), base addresses and lane constants of ALL phases are hoisted to the top
// of the kernel and stay alive across the other phases — 115 spilled registers at the 128 that 4 waves per SIMD allow.
// Re-deriving them per phase (scalar loads from the constant kernel-argument segment + ~200 vector instructions) is what a
// stand-alone kernel's prologue does anyway.
typedef const SolveArgs __attribute__((address_space(4))) * KernArgPtr;
__device__ __forceinline__ const SolveArgs& fresh_args(KernArgPtr base) {
  asm volatile("" : "+s"(base));
  return *(const SolveArgs*)base;
}

__device__ __forceinline__ bool c_dummy_never(const float* pp) { return pp == nullptr; }

// Workgroup barrier that hands GLOBAL-memory data from one wave to the other.  __syncthreads() alone is not enough here: for
// a workgroup-scope release on gfx950 (waves of a workgroup share their CU's L1) the compiler emits `s_waitcnt lgkmcnt(0)`
// only — no vmcnt(0) — on the premise that the CU performs its vector-memory operations in order.  Across two WAVES that
// premise does not hold: a store issued by wave 0 just before the barrier (an `active` flag, the last gain rows of a sweep)
// can still be in flight when wave 1's load after the barrier is served, and the two waves then disagree about a
// workgroup-uniform loop condition (measured: a wave that runs on alone for milliseconds after its partner has left the
// kernel).  The explicit wait makes every store of this wave complete (written through to L2) before it arrives.
__device__ __forceinline__ void wg_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

template <bool RK4>
__global__ __launch_bounds__(128, 4) void solve_quad_kernel(const SolveArgs) {
  const KernArgPtr kap = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();   // the one by-value argument sits at offset 0
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int b0 = blockIdx.x * 2, b = b0 + wv;

  __shared__ __attribute__((aligned(16))) float s_t_all[2 * 16 * LD];
  __shared__ __attribute__((aligned(16))) float s_vx_all[2 * 64];
  constexpr int SWEEP_MODE = RK4 ? MODE_FUSED_RK4 : MODE_FUSED;
  constexpr int LIN_FLOATS = sweep_lin_floats<SWEEP_MODE>();
  __shared__ __attribute__((aligned(16))) float s_lin_all[2 * LIN_FLOATS];
#ifdef QT_SOLVE_LDS_PAD
  __shared__ float s_pad[QT_SOLVE_LDS_PAD];          // experiment: caps the workgroups per CU
  {
    float* pp = s_pad;
    asm volatile("" : "+v"(pp));
    if (c_dummy_never(pp)) pp[threadIdx.x] = 0.0f;
  }
#endif

  const SolveArgs& c = *(const SolveArgs*)kap;     // loop control only (a handful of scalars)
  const bool have = b < c.B;
  const bool force = (c.flags & FLAG_FIXED) != 0;
  const int n_ctrl = c.n_ctrl > 0 ? c.n_ctrl : 1;
  const bool stamping = c.stamps != nullptr && (int)blockIdx.x < c.stamp_rows;
  if (stamping && threadIdx.x == 0) c.stamps[(size_t)blockIdx.x * (2 * (n_ctrl + 1) + 2)] = __builtin_amdgcn_s_memrealtime();
  const bool logging = c.log.rec != nullptr && c.n_ctrl == 0;
  int total_passes = 0;
  for (int cs = 0; cs < n_ctrl; ++cs) {
    int wg_iters = 0;
    if ((c.flags & (FLAG_SIMULATE | FLAG_RESET)) != 0 || c.n_ctrl > 0) {
      // nominal rollout + cost from the current state (simulate :127-132, compute_total_cost :138-143): quads 0 and 1 of wave 0
      if (wv == 0) {
        const SolveArgs& a = fresh_args(kap);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if ((a.n_ctrl > 0 || (a.flags & FLAG_RESET) != 0) && ln < 8 && b0 + (ln >> 2) < a.B) {
          const size_t tb = b0 + (ln >> 2);
          if (a.n_ctrl > 0 && cs == 0 && (ln & 3) < 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              a.traj_x[(tb * (a.n_ctrl + 1)) * NX + 3 * g + (ln & 3)] = a.x0[tb * NX + 3 * g + (ln & 3)];
          }
          if ((ln & 3) == 0) {       // per-solve state of this control step (what a host caller resets before a solve); after
            a.iters[tb] = 0;         // the last control step it stays as that solve left it
            a.active[tb] = 1;
            a.alpha_idx[tb] = -1;
            if (a.status != nullptr) a.status[tb] = 0;
          }
        }
        if ((a.flags & FLAG_SIMULATE) != 0 || a.n_ctrl > 0)
          simulate_quad_body<RK4>(a.fa.p, a.x0, a.u, a.N, a.x, a.cost, 4 * b0 + ln, ln < 8 && b0 + (ln >> 2) < a.B);
      }
      wg_sync();
    }
    for (int it = 0; it < c.max_iter; ++it) {
      // both waves read both flags (written by wave 0, complete before the last barrier; read from L2, past the L1): the loop
      // condition is workgroup-uniform
      const bool act0 = force || __hip_atomic_load(c.active + b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      const bool act1 = (b0 + 1 < c.B) &&
                        (force || __hip_atomic_load(c.active + b0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
      if (!act0 && !act1) break;
      ++wg_iters;
      ++total_passes;
      const bool mine = have && (wv == 0 ? act0 : act1);
      int log_it = 0;
      if (logging && mine) {     // the record of this iteration: nominal, cost, start stamp (this wave's own trajectory)
        const SolveArgs& a = fresh_args(kap);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        log_it = __hip_atomic_load(a.iters + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        log_begin(a.log, b, log_it, a.x + (size_t)b * (a.N + 1) * NX, a.u + (size_t)b * a.N * NU, a.cost[b], ln, 64);
      }
      if (mine) {
        const SolveArgs& a = fresh_args(kap);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        sweep_tile16_body<SWEEP_MODE>(nullptr, nullptr, nullptr, a.N, a.reg, a.K, a.k, a.status, a.fa, b, ln,
                                      s_t_all + wv * 16 * LD, s_vx_all + wv * 64, s_lin_all + wv * LIN_FLOATS);
      }
      if (logging && mine && lane == 0) log_stamp(fresh_args(kap).log, b, log_it, 1, 2);
      wg_sync();
      if (wv == 0) {
        const SolveArgs& a = fresh_args(kap);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // (the line-search wave above every phase of the sweeps it shares its SIMD with — other workgroups': it is the one wave of
        //  two that works in this phase, and its partner waits for it: 91.2 -> 85.5 us per iteration at B = 4096; 1 / 2: 86.3 / 87.3)
        __builtin_amdgcn_s_setprio(QT_SOLVE_LS_PRIO);
        linesearch_quad_body<RK4, 2>(a.fa.p, a.x, a.u, a.K, a.k, a.al, a.n_alpha, a.B, a.N, a.tol, a.cost, a.alpha_idx, a.active,
                                     a.iters, a.scratch, 32 * b0 + ln, force);
        __builtin_amdgcn_s_setprio(0);
      }
      wg_sync();
      if (logging && mine) {     // gains, accepted step, cost after the iteration, end stamp
        const SolveArgs& a = fresh_args(kap);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int ai = __hip_atomic_load(a.alpha_idx + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        log_end(a.log, b, log_it, a.K + (size_t)b * a.N * NU * NX, a.k + (size_t)b * a.N * NU, ai, a.cost[b], ln, 64);
      }
    }
    if (c.n_ctrl > 0) {
      // BOTH waves must have left the loop — i.e. have read the flags that ended it — before wave 0 raises the flags again
      // for the next control step.  Without this barrier a slower wave 1 reads active = 1, makes one more pass, and the two
      // waves' barriers pair up wrongly from then on (found with the per-wave stamps: wave 1 ran on alone for max_iter
      // passes after wave 0 had left the kernel).
      wg_sync();
      if (wv == 0) {
        const SolveArgs& a = fresh_args(kap);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        mpc_advance<RK4>(a, b0, ln, cs);
      }
      wg_sync();
    }
    if (stamping && threadIdx.x == 0) {
      unsigned long long* st = c.stamps + (size_t)blockIdx.x * (2 * (n_ctrl + 1) + 2);
      st[2 * (cs + 1)] = __builtin_amdgcn_s_memrealtime();
      st[2 * (cs + 1) + 1] = (unsigned long long)wg_iters;
    }
  }
  // slot 1: s_memrealtime at which wave 0 left the kernel; the two extra slots at the end of the row: the same for wave 1 and
  // the total number of loop passes wave 1 made
  if (stamping && lane == 0) {
    unsigned long long* st = c.stamps + (size_t)blockIdx.x * (2 * (n_ctrl + 1) + 2);
    if (wv == 0) {
      st[1] = __builtin_amdgcn_s_memrealtime();
    } else {
      st[2 * (n_ctrl + 1)] = __builtin_amdgcn_s_memrealtime();
      st[2 * (n_ctrl + 1) + 1] = (unsigned long long)total_passes;
    }
  }
}

}  // namespace

// flags: bit 0 = roll the nominal out from x0 first; bit 1 = fixed iteration count (stop flags ignored); bit 2 = reset the
// per-solve state (active, iters, alpha_idx, status) first
int quattro_launch_solve_quad(const quattro_model_params& p, const float* x0, float* x, float* u, int B, int N, float reg,
                              const float* alphas, int n_alpha, double tol, int max_iter, int flags, float* K, float* k,
                              double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status,
                              float* scratch, float* coef, int n_ctrl, float* x_cur, float* traj_x, float* traj_u,
                              int32_t* traj_iters, const float* disturbance, unsigned long long* stamps, int stamp_rows,
                              const quattro_solve_log* log, hipStream_t stream) {
  if (p.integrator != QUATTRO_INTEGRATOR_EULER && p.integrator != QUATTRO_INTEGRATOR_RK4) return QUATTRO_ERR_UNSUPPORTED;
  if (p.integrator == QUATTRO_INTEGRATOR_RK4 && coef == nullptr) return QUATTRO_ERR_WORKSPACE;
  SolveArgs a;
  a.fa.p = p;
  a.fa.x = x;
  a.fa.u = u;
  a.fa.N = N;
  a.fa.t_start = 0;
  a.fa.B = B;
  a.fa.coef = coef;     // RK4: the sweep's coefficient scratch, B * N * 132 floats
  a.fa.k_rows = 0;
  a.fa.rn = 12;
  a.fa.rm = 4;
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
  a.stamps = stamps;
  a.stamp_rows = stamp_rows;
  a.log = make_log_dev(n_ctrl > 0 ? nullptr : log, 12, 4, N);
  if (p.integrator == QUATTRO_INTEGRATOR_RK4)
    hipLaunchKernelGGL((solve_quad_kernel<true>), dim3((unsigned)((B + 1) / 2)), dim3(128), 0, stream, a);
  else
    hipLaunchKernelGGL((solve_quad_kernel<false>), dim3((unsigned)((B + 1) / 2)), dim3(128), 0, stream, a);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
