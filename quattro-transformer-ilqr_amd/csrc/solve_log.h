// Per-iteration log ring of a solve (include/quattro_hip.h: quattro_solve_log): what iLQR_TF.optimize appends to self.logs
// every iteration (quattro_ilqr_tf/quattro_ilqr_tf.py:453-466 / :565-578) and what its measure_time decorators append to the
// *_time lists (:16-42), written by the DEVICE so that a logged solve still is one launch and one download.
// Shared by the persistent solve kernels (solve_quad.hip, solve_cartpole.hip, solve_user.hip: the record is filled by the lanes
// that own the trajectory, between the phases of their loop) and by the stand-alone record kernel (solve_log.hip: the same
// records for loops that are enqueued kernel by kernel — hybrid iterations, models without a persistent kernel).
#pragma once
#include "quattro_device.h"

namespace {

// record = [header 64 B | x (N+1) n | u N m | pad 16 | K N m n | k N m | pad 16]; the two optional parts by `flags`
struct SolveLogDev {
  char* rec;        // nullptr = no log
  int capacity;     // slots per trajectory; iteration i -> slot i % capacity
  int flags;        // QUATTRO_LOG_TRAJ | QUATTRO_LOG_GAINS
  unsigned rec_bytes, off_x, off_u, off_K, off_k;
  int nx, nu, nK, nk;   // floats of x, u, K, k of one trajectory
};

struct LogHeader {           // 64 bytes
  unsigned long long stamp[4];   // s_memrealtime (100 MHz): iteration begins / backward pass done / gains complete / line search done
  double cost_pre, cost_new;
  int32_t alpha_idx;             // accepted step (index into the alphas), -1: none; QUATTRO_LOG_PENDING between BEGIN and END
  int32_t iteration;
  int32_t pad[2];
};
static_assert(sizeof(LogHeader) == 64, "log header is 64 bytes");

constexpr int LOG_PENDING = -2;

QT_HD unsigned log_round16(unsigned b) { return (b + 15u) & ~15u; }

inline SolveLogDev make_log_dev(const quattro_solve_log* log, int n, int m, int N) {
  SolveLogDev d{};
  if (log == nullptr || log->records == nullptr || log->capacity <= 0) return d;
  d.rec = (char*)log->records;
  d.capacity = log->capacity;
  d.flags = log->flags;
  d.nx = (N + 1) * n;
  d.nu = N * m;
  d.nK = N * m * n;
  d.nk = N * m;
  unsigned off = 64;
  d.off_x = off;
  d.off_u = off + 4u * d.nx;
  if (log->flags & QUATTRO_LOG_TRAJ) off = log_round16(d.off_u + 4u * d.nu);
  d.off_K = off;
  d.off_k = off + 4u * d.nK;
  if (log->flags & QUATTRO_LOG_GAINS) off = log_round16(d.off_k + 4u * d.nk);
  d.rec_bytes = off;
  return d;
}

__device__ __forceinline__ char* log_slot(const SolveLogDev& lg, int b, int it) {
  return lg.rec + ((size_t)b * lg.capacity + (size_t)(it % lg.capacity)) * lg.rec_bytes;
}

__device__ __forceinline__ void log_copy(float* __restrict__ dst, const float* __restrict__ src, int count, int l, int nl) {
  for (int i = l; i < count; i += nl) dst[i] = src[i];
}

// iteration `it` of trajectory b begins: nominal (x_b, u_b: this trajectory's rows) and its cost.  l = lane in the group of nl
// lanes that owns the trajectory.
__device__ __forceinline__ void log_begin(const SolveLogDev& lg, int b, int it, const float* x_b, const float* u_b,
                                          double cost_pre, int l, int nl) {
  char* r = log_slot(lg, b, it);
  if (l == 0) {
    LogHeader* h = (LogHeader*)r;
    h->stamp[0] = __builtin_amdgcn_s_memrealtime();
    h->cost_pre = cost_pre;
    h->alpha_idx = LOG_PENDING;
    h->iteration = it;
  }
  if (lg.flags & QUATTRO_LOG_TRAJ) {
    log_copy((float*)(r + lg.off_x), x_b, lg.nx, l, nl);
    log_copy((float*)(r + lg.off_u), u_b, lg.nu, l, nl);
  }
}

// stamps `first` .. `last` of the record take the current time (one lane calls this)
__device__ __forceinline__ void log_stamp(const SolveLogDev& lg, int b, int it, int first, int last) {
  LogHeader* h = (LogHeader*)log_slot(lg, b, it);
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  for (int s = first; s <= last; ++s) h->stamp[s] = now;
}

// the iteration is over: gains it used, accepted step, cost after it
__device__ __forceinline__ void log_end(const SolveLogDev& lg, int b, int it, const float* K_b, const float* k_b, int alpha_idx,
                                        double cost_new, int l, int nl) {
  char* r = log_slot(lg, b, it);
  if (lg.flags & QUATTRO_LOG_GAINS) {
    log_copy((float*)(r + lg.off_K), K_b, lg.nK, l, nl);
    log_copy((float*)(r + lg.off_k), k_b, lg.nk, l, nl);
  }
  if (l == 0) {
    LogHeader* h = (LogHeader*)r;
    h->stamp[3] = __builtin_amdgcn_s_memrealtime();
    h->cost_new = cost_new;
    h->alpha_idx = alpha_idx;
  }
}

}  // namespace
