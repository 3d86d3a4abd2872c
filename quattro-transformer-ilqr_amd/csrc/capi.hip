// extern "C" surface of libquattro_hip.so: argument checks + dispatch to the kernels.  See include/quattro_hip.h.
#include "quattro_device.h"

int quattro_launch_sweep_generic(const float*, const float*, const float*, int, int, int, int, float, float*, float*,
                                 int32_t*, const int32_t*, hipStream_t);
int quattro_launch_sweep_tile16(const float*, const float*, const float*, int, int, float, float*, float*, int32_t*,
                                const int32_t*, int, int, int, hipStream_t);
int quattro_launch_sweep_fused(const quattro_model_params&, const float*, const float*, int, int, int, float, float*, float*,
                               int, int32_t*, const int32_t*, hipStream_t);
int quattro_launch_sweep_fused_rk4(const quattro_model_params&, const float*, const float*, int, int, int, float, float*, float*,
                                   int, int32_t*, const int32_t*, float*, hipStream_t);
size_t quattro_sweep_fused_rk4_scratch_floats(int, int);
int quattro_launch_sweep_lane_cartpole(const quattro_model_params&, const float*, const float*, int, int, int, float, float*,
                                       float*, int, int32_t*, const int32_t*, hipStream_t);
int quattro_launch_linearize(const quattro_model_params&, const float*, const float*, int, int, int, int, float*,
                             float*, float*, hipStream_t);
int quattro_launch_pack(const float*, const float*, const float*, const float*, const float*, const float*,
                        const float*, int, int, int, int, int, float*, hipStream_t);
int quattro_launch_unpack(const float*, int, int, int, int, int, float*, float*, float*, float*, float*, float*, float*,
                          hipStream_t);
int quattro_launch_simulate(const quattro_model_params&, const float*, const float*, int, int, float*, double*,
                            hipStream_t);
int quattro_launch_total_cost(const quattro_model_params&, const float*, const float*, int, int, double*, hipStream_t);
int quattro_launch_rollout(const quattro_model_params&, const float*, const float*, const float*, const float*,
                           const float*, int, int, int, float*, float*, double*, const int32_t*, hipStream_t);
int quattro_launch_linesearch(const quattro_model_params&, float*, float*, const float*, const float*, const float*,
                              int, int, int, double, double*, int32_t*, int32_t*, int32_t*, float*, hipStream_t);
size_t quattro_linesearch_scratch_bytes_impl(int, int, int, int);
int quattro_launch_solve_cartpole(const quattro_model_params&, const float*, float*, float*, int, int, float, const float*, int,
                                  double, int, int, float*, float*, double*, int32_t*, int32_t*, int32_t*, int32_t*, float*, int,
                                  float*, float*, float*, int32_t*, const float*, const quattro_solve_log*, hipStream_t);
int quattro_launch_solve_quad(const quattro_model_params&, const float*, float*, float*, int, int, float, const float*, int,
                              double, int, int, float*, float*, double*, int32_t*, int32_t*, int32_t*, int32_t*, float*, float*,
                              int, float*, float*, float*, int32_t*, const float*, unsigned long long*, int,
                              const quattro_solve_log*, hipStream_t);
size_t quattro_solve_log_record_bytes_impl(int, int, int, int);
size_t quattro_solve_log_offset_impl(int, int, int, int, int);
int quattro_launch_solve_log_record(const quattro_solve_log&, int, const float*, const float*, const float*, const float*,
                                    const double*, const int32_t*, const int32_t*, const int32_t*, int, int, int, int, int,
                                    hipStream_t);
#ifdef QT_USER_MODEL_HEADER
int quattro_launch_sweep_rowpad_user(const float*, const float*, const float*, int, int, int, int, float, float*, float*, int32_t*,
                                     const int32_t*, hipStream_t);
int quattro_launch_solve_user(const quattro_model_params&, const float*, float*, float*, int, int, float, const float*, int,
                              double, int, int, float*, float*, double*, int32_t*, int32_t*, int32_t*, int32_t*, float*, float*,
                              float*, float*, int, float*, float*, float*, int32_t*, const float*, const quattro_solve_log*,
                              hipStream_t);
#endif
int quattro_launch_tf_stream(const quattro_tf_weights&, const float*, const float*, int, float*, float*, float*,
                             const int32_t*, int, int, int, hipStream_t);
int quattro_launch_tf_pack(const quattro_tf_weights&, uint16_t*, float*, hipStream_t);
size_t quattro_tf_stream_elems_impl(const quattro_tf_weights&);
size_t quattro_tf_param_floats_impl(const quattro_tf_weights&);

namespace {
bool model_ok(const quattro_model_params* p) {
  if (p == nullptr) return false;
  if (p->model_id == QUATTRO_MODEL_CARTPOLE) return p->n == 4 && p->m == 1;
  if (p->model_id == QUATTRO_MODEL_QUADROTOR) return p->n == 12 && p->m == 4;
#ifdef QT_USER_MODEL_HEADER
  if (p->model_id == QUATTRO_MODEL_USER) return p->n == QT_USER_NX && p->m == QT_USER_NU;
#endif
  return false;
}
}  // namespace

extern "C" {

int quattro_version(void) { return QUATTRO_VERSION; }

const char* quattro_status_string(int status) {
  switch (status) {
    case QUATTRO_OK: return "ok";
    case QUATTRO_ERR_BAD_ARG: return "bad argument";
    case QUATTRO_ERR_UNSUPPORTED: return "unsupported (n, m) / layout / model / integrator combination";
    case QUATTRO_ERR_LAUNCH: return "kernel launch failed";
    case QUATTRO_ERR_WORKSPACE: return "workspace too small or misaligned";
    default: return "unknown status";
  }
}

int quattro_record_stride(int n, int m, int layout) {
  if (layout == QUATTRO_LAYOUT_ROWMAJOR_TILE) {      // the same records as ROWMAJOR, swept by the MFMA tile kernel
    if (n < 1 || n > 12 || m < 1 || m > 4) return 0;
    return (2 * n * n + 2 * n * m + m * m + n + m + 3) / 4 * 4;    // RowMajorRec<n, m>::STRIDE for any (n, m): the kernel takes them at run time
  }
  if (layout == QUATTRO_LAYOUT_ROWMAJOR) {
    if (n == 4 && m == 1) return RowMajorRec<4, 1>::STRIDE;
    if (n == 12 && m == 4) return RowMajorRec<12, 4>::STRIDE;
#ifdef QT_USER_MODEL_HEADER
    if (n == QT_USER_NX && m == QT_USER_NU) return RowMajorRec<QT_USER_NX, QT_USER_NU>::STRIDE;
#endif
    return 0;
  }
  if (layout == QUATTRO_LAYOUT_TILE16) return (n == 12 && m == 4) ? Tile16Rec::STRIDE : 0;
  if (layout == QUATTRO_LAYOUT_TILE16C) return (n == 12 && m == 4) ? Tile16CRec::STRIDE : 0;
  if (layout == QUATTRO_LAYOUT_TILE16R) return (n == 12 && m == 4) ? Tile16RRec::STRIDE : 0;
  return 0;
}

int quattro_record_header(int n, int m, int layout) {
  return ((layout == QUATTRO_LAYOUT_TILE16C || layout == QUATTRO_LAYOUT_TILE16R) && n == 12 && m == 4) ? Tile16Rec::STRIDE : 0;
}

int quattro_model_layout(const quattro_model_params* p) {
  if (!model_ok(p)) return -1;
  if (p->model_id == QUATTRO_MODEL_QUADROTOR && p->integrator == QUATTRO_INTEGRATOR_EULER) return QUATTRO_LAYOUT_TILE16C;
  if (p->model_id == QUATTRO_MODEL_QUADROTOR && p->integrator == QUATTRO_INTEGRATOR_RK4) return QUATTRO_LAYOUT_TILE16R;
  // a user model's kernels (user_linearize.h, solve_user.hip) produce and sweep ROWMAJOR records whatever its (n, m): the
  // host-driven path, the one-call iteration and the persistent kernel must run the SAME sweep to agree bit for bit
  // (n <= 12, m <= 4: those records on the MFMA tile sweep, padded inside the kernel; larger problems on the generic one)
  if (p->model_id == QUATTRO_MODEL_USER)
    return (p->n <= 12 && p->m <= 4 && p->n + p->m >= 6) ? QUATTRO_LAYOUT_ROWMAJOR_TILE : QUATTRO_LAYOUT_ROWMAJOR;
  return quattro_preferred_layout(p->n, p->m);
}

int quattro_preferred_layout(int n, int m) {
  return (n == 12 && m == 4) ? QUATTRO_LAYOUT_TILE16 : QUATTRO_LAYOUT_ROWMAJOR;
}

int quattro_pack_derivs_f32(const float* A, const float* Bm, const float* lx, const float* lu, const float* lxx,
                            const float* luu, const float* lux, int B, int S, int n, int m, int layout, float* rec,
                            void* stream) {
  if (!A || !Bm || !lx || !lu || !lxx || !luu || !lux || !rec || B <= 0 || S <= 0) return QUATTRO_ERR_BAD_ARG;
  if (quattro_record_stride(n, m, layout) == 0 || layout == QUATTRO_LAYOUT_TILE16C || layout == QUATTRO_LAYOUT_TILE16R)
    return QUATTRO_ERR_UNSUPPORTED;
  if (layout == QUATTRO_LAYOUT_ROWMAJOR_TILE) layout = QUATTRO_LAYOUT_ROWMAJOR;
  return quattro_launch_pack(A, Bm, lx, lu, lxx, luu, lux, B, S, n, m, layout, rec, (hipStream_t)stream);
}

int quattro_unpack_derivs_f32(const float* rec, int B, int S, int n, int m, int layout, float* A, float* Bm, float* lx,
                              float* lu, float* lxx, float* luu, float* lux, void* stream) {
  if (!A || !Bm || !lx || !lu || !lxx || !luu || !lux || !rec || B <= 0 || S <= 0) return QUATTRO_ERR_BAD_ARG;
  if (quattro_record_stride(n, m, layout) == 0) return QUATTRO_ERR_UNSUPPORTED;
  if (layout == QUATTRO_LAYOUT_ROWMAJOR_TILE) layout = QUATTRO_LAYOUT_ROWMAJOR;
  return quattro_launch_unpack(rec, B, S, n, m, layout, A, Bm, lx, lu, lxx, luu, lux, (hipStream_t)stream);
}

int quattro_riccati_sweep_f32(const float* rec, const float* VxN, const float* VxxN, int B, int N, int t_start, int n,
                              int m, int layout, float reg, float* K, float* k, int32_t* status,
                              const int32_t* active, void* stream) {
  if (!rec || !VxN || !VxxN || !K || !k || B <= 0 || N <= 0 || t_start < 0 || t_start >= N) return QUATTRO_ERR_BAD_ARG;
  if (quattro_record_stride(n, m, layout) == 0) return QUATTRO_ERR_UNSUPPORTED;
  const int S = N - t_start;
#ifdef QT_USER_MODEL_HEADER
  if (layout == QUATTRO_LAYOUT_ROWMAJOR_TILE)      // this library's own instance (its flags: rounds like its persistent kernel)
    return quattro_launch_sweep_rowpad_user(rec, VxN, VxxN, B, S, n, m, reg, K, k, status, active, (hipStream_t)stream);
#endif
  if (layout == QUATTRO_LAYOUT_TILE16 || layout == QUATTRO_LAYOUT_TILE16C || layout == QUATTRO_LAYOUT_TILE16R ||
      layout == QUATTRO_LAYOUT_ROWMAJOR_TILE)
    return quattro_launch_sweep_tile16(rec, VxN, VxxN, B, S, reg, K, k, status, active, layout, n, m, (hipStream_t)stream);
  return quattro_launch_sweep_generic(rec, VxN, VxxN, B, S, n, m, reg, K, k, status, active, (hipStream_t)stream);
}

int quattro_linearize_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, int t_start,
                          int layout, float* rec, float* VxN, float* VxxN, const int32_t* active, void* stream) {
  (void)active;  // records of inactive trajectories are still produced; the sweep skips them
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x || !u || !rec || B <= 0 || N <= 0 || t_start < 0 || t_start >= N) return QUATTRO_ERR_BAD_ARG;
  if ((VxN == nullptr) != (VxxN == nullptr)) return QUATTRO_ERR_BAD_ARG;
  if (quattro_record_stride(p->n, p->m, layout) == 0) return QUATTRO_ERR_UNSUPPORTED;
  if ((layout == QUATTRO_LAYOUT_TILE16C || layout == QUATTRO_LAYOUT_TILE16R) && quattro_model_layout(p) != layout)
    return QUATTRO_ERR_UNSUPPORTED;
  if (layout == QUATTRO_LAYOUT_ROWMAJOR_TILE) layout = QUATTRO_LAYOUT_ROWMAJOR;       // (same records; the sweep differs)
  return quattro_launch_linearize(*p, x, u, B, N, t_start, layout, rec, VxN, VxxN, (hipStream_t)stream);
}

int quattro_model_fuses_sweep(const quattro_model_params* p) {
  if (!model_ok(p)) return 0;
  if (p->model_id == QUATTRO_MODEL_QUADROTOR)   // 2: available (quattro_linearize_sweep_f32 works) but two launches through records are faster
    return p->integrator == QUATTRO_INTEGRATOR_EULER ? 1 : (p->integrator == QUATTRO_INTEGRATOR_RK4 ? 2 : 0);
  if (p->model_id == QUATTRO_MODEL_CARTPOLE)
    return p->integrator == QUATTRO_INTEGRATOR_EULER || p->integrator == QUATTRO_INTEGRATOR_RK4 ? 1 : 0;
  return 0;
}

size_t quattro_linearize_sweep_scratch_bytes(const quattro_model_params* p, int B, int N, int t_start) {
  if (!model_ok(p) || B <= 0 || N <= 0 || t_start < 0 || t_start >= N) return 0;
  if (p->model_id == QUATTRO_MODEL_QUADROTOR && p->integrator == QUATTRO_INTEGRATOR_RK4)
    return quattro_sweep_fused_rk4_scratch_floats(B, N - t_start) * sizeof(float);
  return 0;
}

int quattro_linearize_sweep_rows_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, int t_start,
                                     float reg, float* K, float* k, int k_rows, int32_t* status, const int32_t* active,
                                     void* scratch, size_t scratch_bytes, void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x || !u || !K || !k || B <= 0 || N <= 0 || t_start < 0 || t_start >= N) return QUATTRO_ERR_BAD_ARG;
  if (k_rows != 0 && k_rows < N - t_start) return QUATTRO_ERR_BAD_ARG;
  if (!quattro_model_fuses_sweep(p)) return QUATTRO_ERR_UNSUPPORTED;
  if (p->model_id == QUATTRO_MODEL_CARTPOLE)
    return quattro_launch_sweep_lane_cartpole(*p, x, u, B, N, t_start, reg, K, k, k_rows, status, active, (hipStream_t)stream);
  if (p->integrator == QUATTRO_INTEGRATOR_RK4) {
    if (!scratch || ((uintptr_t)scratch & 15) != 0 || scratch_bytes < quattro_linearize_sweep_scratch_bytes(p, B, N, t_start))
      return QUATTRO_ERR_WORKSPACE;
    return quattro_launch_sweep_fused_rk4(*p, x, u, B, N, t_start, reg, K, k, k_rows, status, active, (float*)scratch,
                                          (hipStream_t)stream);
  }
  return quattro_launch_sweep_fused(*p, x, u, B, N, t_start, reg, K, k, k_rows, status, active, (hipStream_t)stream);
}

int quattro_linearize_sweep_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, int t_start,
                                float reg, float* K, float* k, int32_t* status, const int32_t* active, void* scratch,
                                size_t scratch_bytes, void* stream) {
  return quattro_linearize_sweep_rows_f32(p, x, u, B, N, t_start, reg, K, k, 0, status, active, scratch, scratch_bytes, stream);
}

int quattro_simulate_f32(const quattro_model_params* p, const float* x0, const float* u, int B, int N, float* x,
                         double* cost, void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x0 || !u || !x || B <= 0 || N <= 0) return QUATTRO_ERR_BAD_ARG;
  return quattro_launch_simulate(*p, x0, u, B, N, x, cost, (hipStream_t)stream);
}

int quattro_total_cost_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, double* cost,
                           void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x || !u || !cost || B <= 0 || N <= 0) return QUATTRO_ERR_BAD_ARG;
  return quattro_launch_total_cost(*p, x, u, B, N, cost, (hipStream_t)stream);
}

int quattro_rollout_f32(const quattro_model_params* p, const float* x_nom, const float* u_nom, const float* K,
                        const float* k, const float* alphas, int n_alpha, int B, int N, float* x_new, float* u_new,
                        double* cost, const int32_t* active, void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x_nom || !u_nom || !K || !k || !alphas || !cost || B <= 0 || N <= 0) return QUATTRO_ERR_BAD_ARG;
  if (n_alpha <= 0 || n_alpha > QUATTRO_MAX_ALPHAS) return QUATTRO_ERR_BAD_ARG;
  if ((x_new == nullptr) != (u_new == nullptr)) return QUATTRO_ERR_BAD_ARG;
  return quattro_launch_rollout(*p, x_nom, u_nom, K, k, alphas, n_alpha, B, N, x_new, u_new, cost, active,
                                (hipStream_t)stream);
}

int quattro_linesearch_f32(const quattro_model_params* p, float* x_nom, float* u_nom, const float* K, const float* k,
                           const float* alphas, int n_alpha, int B, int N, double tol, double* cost,
                           int32_t* alpha_idx, int32_t* active, int32_t* iters, void* scratch, size_t scratch_bytes,
                           void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x_nom || !u_nom || !K || !k || !alphas || !cost || B <= 0 || N <= 0) return QUATTRO_ERR_BAD_ARG;
  if (n_alpha <= 0 || n_alpha > QUATTRO_MAX_ALPHAS) return QUATTRO_ERR_BAD_ARG;
  if (!scratch || ((uintptr_t)scratch & 15) != 0 || scratch_bytes < quattro_linesearch_scratch_bytes(p->n, p->m, B, N))
    return QUATTRO_ERR_WORKSPACE;
  return quattro_launch_linesearch(*p, x_nom, u_nom, K, k, alphas, n_alpha, B, N, tol, cost, alpha_idx, active, iters,
                                   (float*)scratch, (hipStream_t)stream);
}

size_t quattro_linesearch_scratch_bytes(int n, int m, int B, int N) {
  if (n <= 0 || m <= 0 || B <= 0 || N <= 0) return 0;
  return quattro_linesearch_scratch_bytes_impl(n, m, B, N);
}

namespace {
constexpr size_t WS_ALIGN = 256;
inline size_t ws_round(size_t b) { return (b + WS_ALIGN - 1) / WS_ALIGN * WS_ALIGN; }
struct WorkspacePlan {          // offsets in bytes into the caller's workspace
  size_t rec, vx, vxx, scratch, scratch_bytes, total;
};
WorkspacePlan plan_workspace(int n, int m, int B, int N, int layout) {
  WorkspacePlan w{};
  const size_t stride = (size_t)quattro_record_stride(n, m, layout), header = (size_t)quattro_record_header(n, m, layout);
  w.rec = 0;
  w.vx = ws_round((header + (size_t)B * N * stride) * sizeof(float));
  w.vxx = w.vx + ws_round((size_t)B * n * sizeof(float));
  w.scratch = w.vxx + ws_round((size_t)B * n * n * sizeof(float));
  w.scratch_bytes = quattro_linesearch_scratch_bytes_impl(n, m, B, N);
  w.total = w.scratch + ws_round(w.scratch_bytes);
  return w;
}
}  // namespace

size_t quattro_workspace_bytes(int n, int m, int B, int N) {
  if (B <= 0 || N <= 0 || quattro_record_stride(n, m, quattro_preferred_layout(n, m)) == 0) return 0;
  return plan_workspace(n, m, B, N, quattro_preferred_layout(n, m)).total;   // upper bound over the model layouts
}

size_t quattro_model_workspace_bytes(const quattro_model_params* p, int B, int N) {
  if (!model_ok(p) || B <= 0 || N <= 0) return 0;
  return plan_workspace(p->n, p->m, B, N, quattro_model_layout(p)).total;
}

// one iteration; with a log, the "backward pass done" stamp goes between the sweep and the line search
static int quattro_ilqr_iterate_logged(const quattro_model_params* p, float* x_nom, float* u_nom, int B, int N, float reg,
                                       const float* alphas, int n_alpha, double tol, float* K, float* k, double* cost,
                                       int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status, void* workspace,
                                       size_t workspace_bytes, const quattro_solve_log* log, int force, void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x_nom || !u_nom || !K || !k || !alphas || !cost || !alpha_idx || !active || B <= 0 || N <= 0)
    return QUATTRO_ERR_BAD_ARG;
  if (n_alpha <= 0 || n_alpha > QUATTRO_MAX_ALPHAS) return QUATTRO_ERR_BAD_ARG;
  const int layout = quattro_model_layout(p);
  const WorkspacePlan w = plan_workspace(p->n, p->m, B, N, layout);
  if (!workspace || ((uintptr_t)workspace & (WS_ALIGN - 1)) != 0 || workspace_bytes < w.total)
    return QUATTRO_ERR_WORKSPACE;
  char* base = (char*)workspace;
  float* rec = (float*)(base + w.rec);
  float* VxN = (float*)(base + w.vx);
  float* VxxN = (float*)(base + w.vxx);
  int rc;
  if (quattro_model_fuses_sweep(p) == 1) {
    // (RK4 quadrotor: the record area of the workspace doubles as the sweep's coefficient scratch — 528 of its 624 B per step)
    rc = quattro_linearize_sweep_f32(p, x_nom, u_nom, B, N, 0, reg, K, k, status, active, rec,
                                     w.vx - w.rec, stream);
    if (rc != QUATTRO_OK) return rc;
  } else {
    rc = quattro_linearize_f32(p, x_nom, u_nom, B, N, 0, layout, rec, VxN, VxxN, active, stream);
    if (rc != QUATTRO_OK) return rc;
    rc = quattro_riccati_sweep_f32(rec, VxN, VxxN, B, N, 0, p->n, p->m, layout, reg, K, k, status, active, stream);
    if (rc != QUATTRO_OK) return rc;
  }
  if (log != nullptr && iters != nullptr) {
    rc = quattro_launch_solve_log_record(*log, QUATTRO_LOG_PHASE_BACKWARD_DONE, x_nom, u_nom, K, k, cost, alpha_idx, active, iters,
                                         B, N, p->n, p->m, force, (hipStream_t)stream);
    if (rc != QUATTRO_OK) return rc;
  }
  return quattro_linesearch_f32(p, x_nom, u_nom, K, k, alphas, n_alpha, B, N, tol, cost, alpha_idx, active, iters,
                                base + w.scratch, w.scratch_bytes, stream);
}

int quattro_ilqr_iterate_f32(const quattro_model_params* p, float* x_nom, float* u_nom, int B, int N, float reg,
                             const float* alphas, int n_alpha, double tol, float* K, float* k, double* cost,
                             int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return quattro_ilqr_iterate_logged(p, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, K, k, cost, alpha_idx, active, iters,
                                     status, workspace, workspace_bytes, nullptr, 0, stream);
}

// Diagnostics hook (not part of include/quattro_hip.h; scripts/diag_device_loop_stamps.py): a device buffer of
// `rows` x (2 (n_steps + 1) + 2) uint64 that the quadrotor's persistent kernel fills with per-workgroup s_memrealtime stamps and
// iteration counts (row layout: solve_quad.hip); workgroups >= rows do not stamp, so rows = ceil(B / 2) covers a run and a
// smaller buffer is never overrun; NULL (the default) = no stamps.  Process-wide and not thread-safe, like any debug switch.
static unsigned long long* g_solve_stamps = nullptr;
static int g_solve_stamp_rows = 0;
void quattro_debug_set_solve_stamps(unsigned long long* buf, int rows) {
  g_solve_stamps = rows > 0 ? buf : nullptr;
  g_solve_stamp_rows = buf != nullptr && rows > 0 ? rows : 0;
}

int quattro_model_has_device_loop(const quattro_model_params* p) {
  if (!model_ok(p)) return 0;
#ifdef QT_USER_MODEL_HEADER
  if (p->model_id == QUATTRO_MODEL_USER)      // 2: there is a persistent kernel (solve_user.hip), but enqueued iterations are faster
    return (p->integrator == QUATTRO_INTEGRATOR_EULER || p->integrator == QUATTRO_INTEGRATOR_RK4) ? 2 : 0;
#endif
  if (p->model_id == QUATTRO_MODEL_CARTPOLE)
    return (p->integrator == QUATTRO_INTEGRATOR_EULER || p->integrator == QUATTRO_INTEGRATOR_RK4) ? 1 : 0;
  return (p->model_id == QUATTRO_MODEL_QUADROTOR &&
          (p->integrator == QUATTRO_INTEGRATOR_EULER || p->integrator == QUATTRO_INTEGRATOR_RK4)) ? 1 : 0;
}

size_t quattro_solve_log_record_bytes(int n, int m, int N, int flags) {
  if (n <= 0 || m <= 0 || N <= 0) return 0;
  return quattro_solve_log_record_bytes_impl(n, m, N, flags);
}

size_t quattro_solve_log_offset(int n, int m, int N, int flags, int field) {
  if (n <= 0 || m <= 0 || N <= 0) return 0;
  return quattro_solve_log_offset_impl(n, m, N, flags, field);
}

namespace {
bool log_ok(const quattro_solve_log* log) {
  return log == nullptr || log->records == nullptr || (log->capacity > 0 && ((uintptr_t)log->records & 15) == 0);
}
}  // namespace

int quattro_solve_log_record_f32(const quattro_solve_log* log, int phase, const float* x_nom, const float* u_nom,
                                 const float* K, const float* k, const double* cost, const int32_t* alpha_idx,
                                 const int32_t* active, const int32_t* iters, int B, int N, int n, int m, int force,
                                 void* stream) {
  if (!log || !log->records || !log_ok(log) || B <= 0 || N <= 0 || n <= 0 || m <= 0 || !iters) return QUATTRO_ERR_BAD_ARG;
  if (phase < QUATTRO_LOG_PHASE_BEGIN || phase > QUATTRO_LOG_PHASE_END) return QUATTRO_ERR_BAD_ARG;
  if (phase != QUATTRO_LOG_PHASE_END && !force && !active) return QUATTRO_ERR_BAD_ARG;
  if (phase == QUATTRO_LOG_PHASE_BEGIN && (!cost || ((log->flags & QUATTRO_LOG_TRAJ) && (!x_nom || !u_nom))))
    return QUATTRO_ERR_BAD_ARG;
  if (phase == QUATTRO_LOG_PHASE_END && (!cost || !alpha_idx || ((log->flags & QUATTRO_LOG_GAINS) && (!K || !k))))
    return QUATTRO_ERR_BAD_ARG;
  return quattro_launch_solve_log_record(*log, phase, x_nom, u_nom, K, k, cost, alpha_idx, active, iters, B, N, n, m, force,
                                         (hipStream_t)stream);
}

int quattro_ilqr_solve_logged_f32(const quattro_model_params* p, const float* x0, float* x_nom, float* u_nom, int B, int N,
                                  float reg, const float* alphas, int n_alpha, double tol, int max_iter, int flags, float* K,
                                  float* k, double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status,
                                  void* workspace, size_t workspace_bytes, const quattro_solve_log* log, void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x_nom || !u_nom || !K || !k || !alphas || !cost || !alpha_idx || !active || !iters || B <= 0 || N <= 0 || max_iter < 0)
    return QUATTRO_ERR_BAD_ARG;
  if ((flags & QUATTRO_SOLVE_SIMULATE) && !x0) return QUATTRO_ERR_BAD_ARG;
  if (n_alpha <= 0 || n_alpha > QUATTRO_MAX_ALPHAS) return QUATTRO_ERR_BAD_ARG;
  if (!log_ok(log)) return QUATTRO_ERR_BAD_ARG;
  if (log != nullptr && log->records == nullptr) log = nullptr;
  const int layout = quattro_model_layout(p);
  const WorkspacePlan w = plan_workspace(p->n, p->m, B, N, layout);
  if (!workspace || ((uintptr_t)workspace & (WS_ALIGN - 1)) != 0 || workspace_bytes < w.total)
    return QUATTRO_ERR_WORKSPACE;
  char* base = (char*)workspace;
  // the persistent kernel where it is the model's fastest form (1), or on request where it merely exists (2: a user model's)
  const int loop = quattro_model_has_device_loop(p);
  const bool persistent = !(flags & QUATTRO_SOLVE_ENQUEUE) && (loop == 1 || (loop == 2 && (flags & QUATTRO_SOLVE_PERSISTENT)));
  const int kflags = flags & (QUATTRO_SOLVE_SIMULATE | QUATTRO_SOLVE_FIXED_ITERS | QUATTRO_SOLVE_RESET);
#ifdef QT_USER_MODEL_HEADER
  if (persistent && p->model_id == QUATTRO_MODEL_USER)
    return quattro_launch_solve_user(*p, x0, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, max_iter, kflags, K, k, cost, alpha_idx,
                                     active, iters, status, (float*)(base + w.rec), (float*)(base + w.vx), (float*)(base + w.vxx),
                                     (float*)(base + w.scratch), 0, nullptr, nullptr, nullptr, nullptr, nullptr, log,
                                     (hipStream_t)stream);
#endif
  if (persistent && p->model_id == QUATTRO_MODEL_CARTPOLE)
    return quattro_launch_solve_cartpole(*p, x0, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, max_iter, kflags, K, k, cost,
                                         alpha_idx, active, iters, status, (float*)(base + w.scratch), 0, nullptr, nullptr,
                                         nullptr, nullptr, nullptr, log, (hipStream_t)stream);
  if (persistent && p->model_id == QUATTRO_MODEL_QUADROTOR)
    return quattro_launch_solve_quad(*p, x0, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, max_iter, kflags, K, k, cost,
                                     alpha_idx, active, iters, status, (float*)(base + w.scratch), (float*)(base + w.rec), 0,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, g_solve_stamps, g_solve_stamp_rows, log,
                                     (hipStream_t)stream);
  // Enqueued form: the same loop as max_iter iterations of quattro_ilqr_iterate_f32.  Still no host round trip — every
  // kernel skips the trajectories whose `active` flag is down, so the iterations after the last stop are (nearly) empty
  // launches — but max_iter x 2-3 launches are issued whatever the solve needs.
  int rc;
  hipStream_t st = (hipStream_t)stream;
  if (flags & QUATTRO_SOLVE_RESET) {
    if (hipMemsetD32Async((hipDeviceptr_t)active, 1, (size_t)B, st) != hipSuccess ||
        hipMemsetD32Async((hipDeviceptr_t)iters, 0, (size_t)B, st) != hipSuccess ||
        hipMemsetD32Async((hipDeviceptr_t)alpha_idx, -1, (size_t)B, st) != hipSuccess ||
        (status != nullptr && hipMemsetD32Async((hipDeviceptr_t)status, 0, (size_t)B, st) != hipSuccess))
      return QUATTRO_ERR_LAUNCH;
  }
  if (flags & QUATTRO_SOLVE_SIMULATE) {
    rc = quattro_simulate_f32(p, x0, u_nom, B, N, x_nom, cost, stream);
    if (rc != QUATTRO_OK) return rc;
  }
  const int force = (flags & QUATTRO_SOLVE_FIXED_ITERS) ? 1 : 0;
  for (int it = 0; it < max_iter; ++it) {
    if (force)
      if (hipMemsetD32Async((hipDeviceptr_t)active, 1, (size_t)B, st) != hipSuccess) return QUATTRO_ERR_LAUNCH;
    if (log != nullptr) {
      rc = quattro_launch_solve_log_record(*log, QUATTRO_LOG_PHASE_BEGIN, x_nom, u_nom, K, k, cost, alpha_idx, active, iters, B, N,
                                           p->n, p->m, force, st);
      if (rc != QUATTRO_OK) return rc;
    }
    rc = quattro_ilqr_iterate_logged(p, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, K, k, cost, alpha_idx, active, iters,
                                     status, workspace, workspace_bytes, log, force, stream);
    if (rc != QUATTRO_OK) return rc;
    if (log != nullptr) {
      rc = quattro_launch_solve_log_record(*log, QUATTRO_LOG_PHASE_END, x_nom, u_nom, K, k, cost, alpha_idx, active, iters, B, N,
                                           p->n, p->m, force, st);
      if (rc != QUATTRO_OK) return rc;
    }
  }
  return QUATTRO_OK;
}

int quattro_ilqr_solve_f32(const quattro_model_params* p, const float* x0, float* x_nom, float* u_nom, int B, int N,
                           float reg, const float* alphas, int n_alpha, double tol, int max_iter, int flags, float* K,
                           float* k, double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status,
                           void* workspace, size_t workspace_bytes, void* stream) {
  return quattro_ilqr_solve_logged_f32(p, x0, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, max_iter, flags, K, k, cost,
                                       alpha_idx, active, iters, status, workspace, workspace_bytes, nullptr, stream);
}

int quattro_mpc_run_f32(const quattro_model_params* p, float* x_cur, float* x_nom, float* u_nom, int B, int N, float reg,
                        const float* alphas, int n_alpha, double tol, int max_iter, int n_steps, float* traj_x, float* traj_u,
                        int32_t* traj_iters, const float* disturbance, float* K, float* k, double* cost, int32_t* alpha_idx,
                        int32_t* active, int32_t* iters, int32_t* status, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (!model_ok(p)) return p ? QUATTRO_ERR_UNSUPPORTED : QUATTRO_ERR_BAD_ARG;
  if (!x_cur || !x_nom || !u_nom || !K || !k || !alphas || !cost || !alpha_idx || !active || !iters || !traj_x || !traj_u ||
      !traj_iters || B <= 0 || N <= 0 || max_iter < 0 || n_steps <= 0)
    return QUATTRO_ERR_BAD_ARG;
  if (n_alpha <= 0 || n_alpha > QUATTRO_MAX_ALPHAS) return QUATTRO_ERR_BAD_ARG;
  if (!quattro_model_has_device_loop(p)) return QUATTRO_ERR_UNSUPPORTED;
  const WorkspacePlan w = plan_workspace(p->n, p->m, B, N, quattro_model_layout(p));
  if (!workspace || ((uintptr_t)workspace & (WS_ALIGN - 1)) != 0 || workspace_bytes < w.total)
    return QUATTRO_ERR_WORKSPACE;
#ifdef QT_USER_MODEL_HEADER
  if (p->model_id == QUATTRO_MODEL_USER) {
    char* base = (char*)workspace;
    return quattro_launch_solve_user(*p, x_cur, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, max_iter, 0, K, k, cost, alpha_idx,
                                     active, iters, status, (float*)(base + w.rec), (float*)(base + w.vx), (float*)(base + w.vxx),
                                     (float*)(base + w.scratch), n_steps, x_cur, traj_x, traj_u, traj_iters, disturbance,
                                     nullptr, (hipStream_t)stream);
  }
#endif
  if (p->model_id == QUATTRO_MODEL_CARTPOLE)
    return quattro_launch_solve_cartpole(*p, x_cur, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, max_iter, 0, K, k, cost,
                                         alpha_idx, active, iters, status, (float*)((char*)workspace + w.scratch), n_steps,
                                         x_cur, traj_x, traj_u, traj_iters, disturbance, nullptr, (hipStream_t)stream);
  return quattro_launch_solve_quad(*p, x_cur, x_nom, u_nom, B, N, reg, alphas, n_alpha, tol, max_iter, 0, K, k, cost, alpha_idx,
                                   active, iters, status, (float*)((char*)workspace + w.scratch), (float*)((char*)workspace + w.rec),
                                   n_steps, x_cur, traj_x, traj_u, traj_iters, disturbance, g_solve_stamps, g_solve_stamp_rows,
                                   nullptr, (hipStream_t)stream);
}

namespace {
// the PyTorch-layout arrays (inputs of the packing)
int tf_check_arrays(const quattro_tf_weights* w) {
  if (!w) return QUATTRO_ERR_BAD_ARG;
  if (!w->x_mean || !w->x_std || !w->u_mean || !w->u_std || !w->w_state || !w->state_b || !w->ctrl_w || !w->ctrl_b ||
      !w->w_out || !w->b_out)
    return QUATTRO_ERR_BAD_ARG;
  if (w->n_layers < 1 || w->n_layers > QUATTRO_TF_MAX_LAYERS) return QUATTRO_ERR_UNSUPPORTED;
  for (int l = 0; l < w->n_layers; ++l)
    if (!w->w_qkv[l] || !w->b_qkv[l] || !w->w_o[l] || !w->b_o[l] || !w->w_1[l] || !w->b_1[l] || !w->w_2[l] ||
        !w->b_2[l] || !w->ln1_g[l] || !w->ln1_b[l] || !w->ln2_g[l] || !w->ln2_b[l])
      return QUATTRO_ERR_BAD_ARG;
  return QUATTRO_OK;
}
// what the forward reads
int tf_check(const quattro_tf_weights* w, const float* x_err, const float* prompt, const float* pred, int B) {
  if (!w || !x_err || !prompt || !pred || B <= 0) return QUATTRO_ERR_BAD_ARG;
  if (!w->x_mean || !w->x_std || !w->u_mean || !w->u_std || !w->tok_bias_t || !w->w_stream || !w->p_stream)
    return QUATTRO_ERR_BAD_ARG;
  if (w->n_layers < 1 || w->n_layers > QUATTRO_TF_MAX_LAYERS) return QUATTRO_ERR_UNSUPPORTED;
  return QUATTRO_OK;
}
}  // namespace

size_t quattro_tf_stream_elems(const quattro_tf_weights* w) { return w ? quattro_tf_stream_elems_impl(*w) : 0; }
size_t quattro_tf_param_floats(const quattro_tf_weights* w) { return w ? quattro_tf_param_floats_impl(*w) : 0; }

namespace {
int tf_pack_any(const quattro_tf_weights* w, int precision, uint16_t* w_stream, float* p_stream, void* stream) {
  const int rc = tf_check_arrays(w);
  if (rc != QUATTRO_OK) return rc;
  if (!w_stream || !p_stream || w->precision != precision) return QUATTRO_ERR_BAD_ARG;
  return quattro_launch_tf_pack(*w, w_stream, p_stream, (hipStream_t)stream);
}
int tf_forward_any(const quattro_tf_weights* w, int precision, const float* x_err, const float* prompt, int B, float* pred,
                   void* stream) {
  const int rc = tf_check(w, x_err, prompt, pred, B);
  if (rc != QUATTRO_OK) return rc;
  if (w->precision != precision) return QUATTRO_ERR_BAD_ARG;
  return quattro_launch_tf_stream(*w, x_err, prompt, B, pred, nullptr, nullptr, nullptr, 0, 0, 0, (hipStream_t)stream);
}
int tf_gains_any(const quattro_tf_weights* w, int precision, const float* x_err, const float* prompt, int B, int N, int n,
                 int m, float* K, float* k, const int32_t* active, void* stream) {
  if (!K || !k || N <= 0 || n <= 0 || m <= 0) return QUATTRO_ERR_BAD_ARG;
  if (w && (w->c_dim != m * (n + 1) || w->precision != precision)) return QUATTRO_ERR_BAD_ARG;
  // prompt == NULL: the P prompt rows are read from the gain stacks themselves (rows N - P .. N - 1: the swept tail), which
  // the kernel must not also write: the predicted rows t < T have to end below them
  if (!prompt && w && w->target_len + w->prompt_len > N) return QUATTRO_ERR_BAD_ARG;
  float dummy;   // never written in gains mode; only makes the shared argument check below pass
  const int rc = tf_check(w, x_err, prompt ? prompt : &dummy, &dummy, B);
  if (rc != QUATTRO_OK) return rc;
  return quattro_launch_tf_stream(*w, x_err, prompt, B, nullptr, K, k, active, N, n, m, (hipStream_t)stream);
}
}  // namespace

int quattro_tf_pack_stream_bf16(const quattro_tf_weights* w, uint16_t* w_stream, float* p_stream, void* stream) {
  return tf_pack_any(w, QUATTRO_TF_PRECISION_BF16, w_stream, p_stream, stream);
}
int quattro_tf_forward_bf16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, float* pred,
                            void* stream) {
  return tf_forward_any(w, QUATTRO_TF_PRECISION_BF16, x_err, prompt, B, pred, stream);
}
int quattro_tf_gains_bf16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, int N, int n, int m,
                          float* K, float* k, const int32_t* active, void* stream) {
  return tf_gains_any(w, QUATTRO_TF_PRECISION_BF16, x_err, prompt, B, N, n, m, K, k, active, stream);
}
int quattro_tf_pack_stream_f16(const quattro_tf_weights* w, uint16_t* w_stream, float* p_stream, void* stream) {
  return tf_pack_any(w, QUATTRO_TF_PRECISION_F16, w_stream, p_stream, stream);
}
int quattro_tf_forward_f16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, float* pred,
                           void* stream) {
  return tf_forward_any(w, QUATTRO_TF_PRECISION_F16, x_err, prompt, B, pred, stream);
}
int quattro_tf_gains_f16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, int N, int n, int m,
                         float* K, float* k, const int32_t* active, void* stream) {
  return tf_gains_any(w, QUATTRO_TF_PRECISION_F16, x_err, prompt, B, N, n, m, K, k, active, stream);
}

}  // extern "C"
