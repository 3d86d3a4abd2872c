// Stand-alone form of the per-iteration solve log (solve_log.h): one small launch per phase for loops that are enqueued kernel
// by kernel — the hybrid iteration (tail sweep, predictor, line search) and models without a persistent kernel.  The records it
// writes are the ones the persistent kernels write between their phases; what iLQR_TF.optimize logs per iteration
// (quattro_ilqr_tf/quattro_ilqr_tf.py:453-466 / :565-578) and times (:16-42).
#include "solve_log.h"

namespace {

struct LogRecordArgs {
  SolveLogDev lg;
  const float* x;
  const float* u;
  const float* K;
  const float* k;
  const double* cost;
  const int32_t* alpha_idx;
  const int32_t* active;
  const int32_t* iters;
  int B, phase, force;
};

// one wave per trajectory
__global__ __launch_bounds__(QT_WAVE) void solve_log_record_kernel(const LogRecordArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= a.B) return;
  const int it = a.iters[b];
  if (a.phase == QUATTRO_LOG_PHASE_END) {
    // the line search has run: iters[b] was incremented for exactly the trajectories that were live at BEGIN, whose record is
    // still pending
    if (it <= 0) return;
    const LogHeader* h = (const LogHeader*)log_slot(a.lg, b, it - 1);
    if (h->alpha_idx != LOG_PENDING || h->iteration != it - 1) return;
    log_end(a.lg, b, it - 1, a.K + (size_t)b * a.lg.nK, a.k + (size_t)b * a.lg.nk, a.alpha_idx[b], a.cost[b], lane, QT_WAVE);
    return;
  }
  if (!(a.force || a.active[b] != 0)) return;
  if (a.phase == QUATTRO_LOG_PHASE_BEGIN) {
    log_begin(a.lg, b, it, a.x + (size_t)b * a.lg.nx, a.u + (size_t)b * a.lg.nu, a.cost[b], lane, QT_WAVE);
  } else if (lane == 0) {
    log_stamp(a.lg, b, it, a.phase == QUATTRO_LOG_PHASE_BACKWARD_DONE ? 1 : 2, 2);
  }
}

}  // namespace

size_t quattro_solve_log_record_bytes_impl(int n, int m, int N, int flags) {
  quattro_solve_log l{(void*)16, 1, flags};
  return make_log_dev(&l, n, m, N).rec_bytes;
}

size_t quattro_solve_log_offset_impl(int n, int m, int N, int flags, int field) {
  quattro_solve_log l{(void*)16, 1, flags};
  const SolveLogDev d = make_log_dev(&l, n, m, N);
  switch (field) {
    case QUATTRO_LOG_FIELD_X: return (flags & QUATTRO_LOG_TRAJ) ? d.off_x : 0;
    case QUATTRO_LOG_FIELD_U: return (flags & QUATTRO_LOG_TRAJ) ? d.off_u : 0;
    case QUATTRO_LOG_FIELD_K: return (flags & QUATTRO_LOG_GAINS) ? d.off_K : 0;
    case QUATTRO_LOG_FIELD_KFF: return (flags & QUATTRO_LOG_GAINS) ? d.off_k : 0;
    default: return 0;
  }
}

int quattro_launch_solve_log_record(const quattro_solve_log& log, int phase, const float* x, const float* u, const float* K,
                                    const float* k, const double* cost, const int32_t* alpha_idx, const int32_t* active,
                                    const int32_t* iters, int B, int N, int n, int m, int force, hipStream_t stream) {
  LogRecordArgs a;
  a.lg = make_log_dev(&log, n, m, N);
  a.x = x;
  a.u = u;
  a.K = K;
  a.k = k;
  a.cost = cost;
  a.alpha_idx = alpha_idx;
  a.active = active;
  a.iters = iters;
  a.B = B;
  a.phase = phase;
  a.force = force;
  hipLaunchKernelGGL(solve_log_record_kernel, dim3((unsigned)B), dim3(QT_WAVE), 0, stream, a);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
