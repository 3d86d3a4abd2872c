/*
 * quattro_hip.h — C ABI of libquattro_hip.so, the MI355X (gfx950) implementation of the Quattro iLQR
 * hot path.  Plain C: device pointers as `float*`/`int*`, sizes as int, a HIP stream as `void*`.
 *
 * The reference (salemon/quattro-transformer-ilqr) has NO native/FFI boundary: the path lives in the
 * Python class quattro_ilqr_tf/quattro_ilqr_tf.py:50 (iLQR_TF) and in
 * quattro_ilqr_tf/transformer_model.py:85 (TransformerPredictor).  Each entry point below names the
 * reference method whose arithmetic it replaces; the Python host package
 * (quattro-transformer-ilqr_amd/quattro_ilqr_amd) binds them with ctypes and re-exposes the reference's
 * own interface (iLQR_TF.optimize/backward_pass/…, TransformerILQR.predict) on top.  INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns a status: 0 = QUATTRO_OK, < 0 = argument / launch error (nothing launched);
 *   - all array arguments are DEVICE pointers (fp32 unless stated), row-major, indices [b][t][...], 16-byte aligned
 *     (the kernels move state rows, gain rows and records with 16-byte accesses; any allocator's base pointer is);
 *   - launches are asynchronous on `stream` (a hipStream_t, may be NULL = default stream);
 *   - the library allocates nothing and keeps no global state; scratch comes from the caller
 *     (quattro_*_workspace_bytes);
 *   - per-trajectory numerical trouble is reported in an int32 status word per trajectory
 *     (QUATTRO_TRAJ_*), never by aborting.
 */
#ifndef QUATTRO_HIP_H
#define QUATTRO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QUATTRO_VERSION 100 /* 0.1.0 */

/* status codes */
#define QUATTRO_OK 0
#define QUATTRO_ERR_BAD_ARG (-1)     /* null pointer, negative size, t_start outside [0,N) ...            */
#define QUATTRO_ERR_UNSUPPORTED (-2) /* (n,m)/layout/model/integrator combination without a device kernel */
#define QUATTRO_ERR_LAUNCH (-3)      /* hipGetLastError() after the launch was not hipSuccess             */
#define QUATTRO_ERR_WORKSPACE (-4)   /* workspace too small / misaligned                                  */

/* per-trajectory status bits (int32 status[b]) */
#define QUATTRO_TRAJ_NONFINITE 1 /* a non-finite value appeared in the recursion / rollout            */
#define QUATTRO_TRAJ_SINGULAR 2  /* Q_uu + reg*I had a zero / non-finite pivot (np.linalg.inv would raise) */
#define QUATTRO_TRAJ_ILLCOND 4   /* TILE16 records only: the quadrotor-shaped sweep eliminates Q_uu + reg*I WITHOUT pivoting,
                                  * which assumes a symmetric positive definite block (any convex cost).  A pivot <= 0 or
                                  * <= 1e-6 x the diagonal entry it started from sets this bit: K, k of that trajectory
                                  * may be inaccurate.  The ROWMAJOR path pivots like the reference's LAPACK inverse
                                  * (quattro_ilqr_tf.py:306) and never sets it; re-run flagged trajectories there.     */

/* device models: the reference takes Python callables f, L, Lf (quattro_ilqr_tf.py:82-84); a kernel
 * cannot call Python, so the two shipped problems are built in and selected by id.                   */
#define QUATTRO_MODEL_CARTPOLE 1  /* examples/cartpole/cartpole_dynamics.py:32-108 + cartpole_mpc.py:187-269     */
#define QUATTRO_MODEL_QUADROTOR 2 /* examples/quadrotor/quadrotor_dynamics.py:47-198 + quadrotor_mpc.py:40-100  */
/* Any other problem (what the reference's callable arguments allow, quattro_ilqr_tf.py:66-84): written as three function
 * templates over the scalar type (csrc/user_model.h), compiled together with the generic kernels into a library of its own
 * that exports THIS header's ABI (quattro_ilqr_amd.user_model.compile_model does it: hipcc, ~20 s, cached).  In such a
 * library model_id QUATTRO_MODEL_USER selects the compiled-in problem with its (n, m) <= (QUATTRO_MAX_NX, QUATTRO_MAX_NU):
 * exact derivatives by forward-mode differentiation through the whole integrator step (csrc/dual.h) where the reference takes
 * finite differences, ROWMAJOR records, the pivoting generic sweep, one lane per line-search candidate.  libquattro_hip.so
 * itself answers QUATTRO_ERR_UNSUPPORTED for this id.  phys[0..7] are the model's free parameters.                          */
#define QUATTRO_MODEL_USER 3

#define QUATTRO_INTEGRATOR_EULER 0
#define QUATTRO_INTEGRATOR_RK4 1

#define QUATTRO_MAX_NX 16
#define QUATTRO_MAX_NU 8
#define QUATTRO_MAX_ALPHAS 8

/* Problem definition handed to the kernels BY VALUE (host struct).  Costs are
 *   L(x,u)  = sum_i q[i] (x_i - x_ref_i)^2 + sum_a r[a] u_a^2 + barrier_alpha * sum_a softplus_beta(-u_a)^2
 *   Lf(x)   = sum_i qf[i] (x_i - x_ref_i)^2
 * which covers both shipped costs (diagonal Q, R, Qf; barrier_alpha = 0 for the cart-pole).
 * phys[]: cart-pole {m_cart, m_pole, length, gravity}; quadrotor {mass, Ix, Iy, Iz, arm, gravity, k_yaw}. */
typedef struct quattro_model_params {
  int32_t model_id;
  int32_t integrator;
  int32_t n; /* state dim   (4 / 12) */
  int32_t m; /* control dim (1 / 4)  */
  float dt;
  float barrier_alpha;
  float barrier_beta;
  float reserved0;
  float phys[8];
  float q[QUATTRO_MAX_NX];
  float qf[QUATTRO_MAX_NX];
  float x_ref[QUATTRO_MAX_NX];
  float r[QUATTRO_MAX_NU];
} quattro_model_params;

/* Derivative-record layouts.  One record per (trajectory b, step t) holds the blocks the sweep consumes
 * — A (n x n), B (n x m), l_xx (n x n), l_ux (m x n), l_uu (m x m), l_x (n), l_u (m) —
 * 2n^2 + 2nm + m^2 + n + m floats (416 for the quadrotor, 46 for the cart-pole), stored contiguously so a
 * wave streams it with full-width coalesced loads.  Record (b,t) starts at float offset
 * (b*N + t) * quattro_record_stride(n,m,layout).
 *   ROWMAJOR: [A | B | l_xx | l_ux | l_uu | l_x | l_u], each block row-major, stride padded to 4 floats.
 *   TILE16  : (n,m) = (12,4) only.  The same 416 floats permuted into the per-lane order of the 16x16
 *             wave tile the quadrotor sweep kernel computes on (see DESIGN.md "tile16 record").        */
#define QUATTRO_LAYOUT_ROWMAJOR 0
#define QUATTRO_LAYOUT_TILE16 1
/*   TILE16C : quadrotor with the Euler integrator only, produced by quattro_linearize_f32 (not by quattro_pack_derivs_f32).
 *             Most of an Euler record is a constant of the problem (identity / dt entries of A, the torque rows of B,
 *             l_xx = 2Q, l_ux = 0): those live once in a HEADER record at the start of the buffer, and each (b,t) record
 *             keeps only the 76 floats that depend on (x_t, u_t).  The buffer is
 *             quattro_record_header(n,m,layout) floats of header followed by B*(N - t_start) records of
 *             quattro_record_stride floats: 304 B per step through HBM instead of 1,664 B, same sweep arithmetic.     */
#define QUATTRO_LAYOUT_TILE16C 2
/*   TILE16R : quadrotor with the RK4 integrator, produced by quattro_linearize_f32.  Ten columns of [A | B] (the angle, body-rate
 *             and control directions) are dense and change every step; the position and velocity directions give constant
 *             columns (unit vectors / e_v + dt e_p through all four stages), which sit in the header record together with
 *             l_xx = 2Q, l_ux = 0 of the built-in cost.  Each (b,t) record keeps those ten columns, l_uu and l_z: 156 floats
 *             = 624 B per step instead of 1,664 B.                                                                        */
#define QUATTRO_LAYOUT_TILE16R 3
/*   ROWMAJOR_TILE : ROWMAJOR records (the same buffer, byte for byte) of a problem with n <= 12, m <= 4, swept by the MFMA
 *             tile kernel instead of the generic one: the kernel zero-pads the problem into its 16 x 16 tile as it loads
 *             (unit pivots for the controls that are not there), so the records stay as small as the problem.  Like every
 *             tile sweep it eliminates WITHOUT pivoting (QUATTRO_TRAJ_ILLCOND): sweep flagged trajectories again with
 *             layout ROWMAJOR — no repacking needed.  What a user-compiled model with n <= 12, m <= 4 gets.              */
#define QUATTRO_LAYOUT_ROWMAJOR_TILE 4

int quattro_version(void);
const char* quattro_status_string(int status);

/* floats between consecutive records; 0 if the combination is unsupported */
int quattro_record_stride(int n, int m, int layout);
/* floats of header in front of the first record (0 except for TILE16C / TILE16R) */
int quattro_record_header(int n, int m, int layout);
/* the layout the fastest sweep kernel for (n,m) wants when the records come from anywhere (quattro_pack_derivs_f32) */
int quattro_preferred_layout(int n, int m);
/* the layout quattro_linearize_f32 + quattro_riccati_sweep_f32 are fastest with for this model: TILE16C for the
 * Euler quadrotor, TILE16R for the RK4 quadrotor, quattro_preferred_layout(n, m) otherwise; -1 for an unknown model */
int quattro_model_layout(const quattro_model_params* p);

/* Gather separately stored row-major blocks into records (test/utility path; the linearisation kernel writes
 * records directly).  A[B*S][n][n], Bm[B*S][n][m], lx[B*S][n], lu[B*S][m], lxx[B*S][n][n], luu[B*S][m][m],
 * lux[B*S][m][n]  ->  rec[B*S][stride].  Replaces nothing in the reference (pure data movement).        */
int quattro_pack_derivs_f32(const float* A, const float* Bm, const float* lx, const float* lu, const float* lxx,
                            const float* luu, const float* lux, int B, int S, int n, int m, int layout, float* rec,
                            void* stream);

/* The inverse: records (any layout, TILE16C included: rec points at the header) -> separately stored row-major blocks,
 * i.e. the arrays the reference's own methods return: A, B of _compute_dynamics_jacobians (:182-204) and L_x, L_u, L_xx,
 * L_uu, L_ux (= L_xu^T) of _compute_cost_derivatives (:217-275) for every (b, t) of the buffer.                      */
int quattro_unpack_derivs_f32(const float* rec, int B, int S, int n, int m, int layout, float* A, float* Bm, float* lx,
                              float* lu, float* lxx, float* luu, float* lux, void* stream);

/* Riccati-like backward sweep.  Replaces the recursion of iLQR_TF.backward_pass
 * (quattro_ilqr_tf.py:290-317) and, with t_start > 0, iLQR_TF.backward_pass_segment (:336-364).
 *   rec   : records for steps t_start..N-1 of every trajectory, [B][N - t_start][stride]
 *   VxN   : [B][n]    terminal gradient   (reference: _finite_diff_gradient_final, :149)
 *   VxxN  : [B][n][n] terminal Hessian    (reference: _finite_diff_hessian_final,  :163; used as given)
 *   reg   : added to diag(Q_uu) before inversion only (1e-6 in the reference, :304)
 *   K     : [B][N - t_start][m][n], k: [B][N - t_start][m]   (index t - t_start, as :357-359)
 *   status: [B] QUATTRO_TRAJ_* bits (may be NULL)
 *   active: [B] optional mask; trajectories with active[b] == 0 are skipped (outputs untouched); NULL = all */
int quattro_riccati_sweep_f32(const float* rec, const float* VxN, const float* VxxN, int B, int N, int t_start, int n,
                              int m, int layout, float reg, float* K, float* k, int32_t* status,
                              const int32_t* active, void* stream);

/* Linearisation about a nominal trajectory: exact derivatives of the device model, written as records.
 * Replaces _compute_dynamics_jacobians (:182-204), _compute_cost_derivatives (:217-275) for every step
 * t in [t_start, N) and the two _finite_diff_*_final (:149-174) at x_N.
 *   x: [B][N+1][n], u: [B][N][m]  ->  rec [B][N - t_start][stride], VxN [B][n], VxxN [B][n][n]          */
int quattro_linearize_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, int t_start,
                          int layout, float* rec, float* VxN, float* VxxN, const int32_t* active, void* stream);

/* Linearisation and sweep in ONE launch, no record buffer (every built-in model; `scratch` may be NULL where
 * quattro_linearize_sweep_scratch_bytes is 0, else 16-byte aligned device memory of at least that size):
 *   - the Euler-discretised quadrotor: the sweep's own wave linearises its trajectory ahead of the recursion (17 steps
 *     at a time into LDS); per step 64 B instead of 304 B come from HBM and one launch disappears;
 *   - the RK4-discretised quadrotor: the same, in two stages (stage points -> Jacobian coefficients by one lane per step,
 *     then 4 steps x 16 unit directions per pass through the four stages); 0.9 KB per step of TILE16R records and the
 *     linearisation launch disappear;
 *   - the cart-pole (Euler and RK4): a 16-lane DPP row per trajectory runs the 4 x 4 recursion on records formed ahead of
 *     the chain in LDS — instead of a 64-lane wave per trajectory with seven barriers per step (one launch instead of three).
 * Same K, k as quattro_linearize_f32 (model layout) followed by quattro_riccati_sweep_f32: bit-identical for the quadrotor,
 * to fp32 round-off (<= 2e-6 per step) for the cart-pole.  Replaces
 * _compute_dynamics_jacobians / _compute_cost_derivatives / _finite_diff_*_final + backward_pass(_segment)
 * (quattro_ilqr_tf.py:149-275, :290-317 / :336-364).  QUATTRO_ERR_UNSUPPORTED for other models (quattro_model_fuses_sweep
 * says which): use the two calls above.
 *   x [B][N+1][n], u [B][N][m]  ->  K [B][N - t_start][m][n], k [B][N - t_start][m], status [B] (may be NULL)          */
/* 0: no fused kernel; 1: quattro_linearize_sweep_f32 is this model's fastest backward pass; 2: it works, but
 * quattro_linearize_f32 + quattro_riccati_sweep_f32 through records are faster as stand-alone launches (RK4 quadrotor: 180 us
 * against 58 + 69 us at B = 4096 — the fused form exists for the device-resident loop, which cannot run a second kernel) */
int quattro_model_fuses_sweep(const quattro_model_params* p);
/* device scratch quattro_linearize_sweep_f32 needs for this model and size (0 for all but the RK4 quadrotor, whose wave leaves
 * the coefficients of its four stage Jacobians — 528 B per step — in it between the two stages of its linearisation) */
size_t quattro_linearize_sweep_scratch_bytes(const quattro_model_params* p, int B, int N, int t_start);
int quattro_linearize_sweep_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, int t_start,
                                float reg, float* K, float* k, int32_t* status, const int32_t* active, void* scratch,
                                size_t scratch_bytes, void* stream);

/* The same launch writing into gain arrays of `k_rows` rows per trajectory (K [B][k_rows][m][n], k [B][k_rows][m]): step t lands
 * at row k_rows - (N - t).  k_rows = 0 or N - t_start: the segment form above (index t - t_start, quattro_ilqr_tf.py:357-359);
 * k_rows = N: the FULL gain stacks, the swept tail written in place at rows t_start .. N - 1 — where the hybrid iteration's
 * line search reads it and where quattro_tf_gains_* (prompt = NULL) reads its prompt from: no segment buffers, no packing and
 * no concatenation (np.concatenate of :515-518) in a hybrid iteration.                                                      */
int quattro_linearize_sweep_rows_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, int t_start,
                                     float reg, float* K, float* k, int k_rows, int32_t* status, const int32_t* active,
                                     void* scratch, size_t scratch_bytes, void* stream);

/* Open-loop rollout + total cost.  Replaces iLQR_TF.simulate (:127-132) + compute_total_cost (:138-143).
 *   x0 [B][n], u [B][N][m]  ->  x [B][N+1][n], cost [B] (fp64)                                          */
int quattro_simulate_f32(const quattro_model_params* p, const float* x0, const float* u, int B, int N, float* x,
                         double* cost, void* stream);

/* Total cost of given sequences (they need not satisfy the dynamics).  Replaces compute_total_cost (:138-143).
 *   x [B][N+1][n], u [B][N][m]  ->  cost [B] (fp64)                                                        */
int quattro_total_cost_f32(const quattro_model_params* p, const float* x, const float* u, int B, int N, double* cost,
                           void* stream);

/* Closed-loop rollouts for n_alpha step sizes at once.  Replaces iLQR_TF.forward_pass (:377-390) for every
 * alpha of the line search (:440 / :552).
 *   x_nom [B][N+1][n], u_nom [B][N][m], K [B][N][m][n], k [B][N][m], alphas: HOST array of n_alpha floats
 *   -> cost [n_alpha][B] (fp64);  x_new / u_new: optional [n_alpha][B][N+1][n] / [n_alpha][B][N][m] (NULL = costs only) */
int quattro_rollout_f32(const quattro_model_params* p, const float* x_nom, const float* u_nom, const float* K,
                        const float* k, const float* alphas, int n_alpha, int B, int N, float* x_new, float* u_new,
                        double* cost, const int32_t* active, void* stream);

/* Line search + accept, fused.  Replaces the alpha loop of iLQR_TF.optimize (:433-451 / :546-563) and its stop
 * test (:472 / :584): evaluates every alpha, takes the FIRST one (in the given order) whose cost is <= cost[b],
 * and for accepting trajectories overwrites x_nom/u_nom/cost in place with the accepted candidate.
 *   alpha_idx [B] : index of the accepted alpha, -1 if none
 *   active    [B] : in/out; cleared when no alpha was accepted or |cost_old - cost_new| < tol (converged).
 *                   Inactive trajectories are left untouched.
 *   iters     [B] : incremented for every trajectory that was active on entry (may be NULL)
 *   scratch       : device buffer of at least quattro_linesearch_scratch_bytes(n, m, B, N) bytes, 16-byte aligned;
 *                   holds the candidate trajectories between the rollouts and the commit                   */
size_t quattro_linesearch_scratch_bytes(int n, int m, int B, int N);
int quattro_linesearch_f32(const quattro_model_params* p, float* x_nom, float* u_nom, const float* K, const float* k,
                           const float* alphas, int n_alpha, int B, int N, double tol, double* cost,
                           int32_t* alpha_idx, int32_t* active, int32_t* iters, void* scratch, size_t scratch_bytes,
                           void* stream);

/* One whole pure-iLQR iteration for B trajectories: the body of the while-loop of iLQR_TF.optimize (:428-472) —
 * linearise about (x_nom, u_nom), Riccati sweep, 6-alpha line search with accept/commit and the stop test — as three
 * launches on `stream` from ONE host call (the `quattro_ilqr_iterate` fused driver) — two where the model fuses the
 * linearisation into the sweep (quattro_model_fuses_sweep).  Exactly equivalent to
 * quattro_linearize_f32 (t_start = 0, preferred layout) + quattro_riccati_sweep_f32 + quattro_linesearch_f32 on the
 * same buffers.  `workspace` (device, 256-byte aligned, >= quattro_model_workspace_bytes(p, B, N) bytes) holds the
 * derivative records, V_x(N), V_xx(N) and the candidate trajectories; nothing in it needs to survive between calls.
 *   in/out: x_nom [B][N+1][n], u_nom [B][N][m], cost [B] fp64 (cost of the nominal on entry), active [B], iters [B]
 *   out   : K [B][N][m][n], k [B][N][m], alpha_idx [B], status [B] (QUATTRO_TRAJ_* bits, may be NULL)             */
size_t quattro_workspace_bytes(int n, int m, int B, int N);                            /* enough for any model of these dims */
size_t quattro_model_workspace_bytes(const quattro_model_params* p, int B, int N);    /* exact for this model (<= the above) */
int quattro_ilqr_iterate_f32(const quattro_model_params* p, float* x_nom, float* u_nom, int B, int N, float reg,
                             const float* alphas, int n_alpha, double tol, float* K, float* k, double* cost,
                             int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status, void* workspace,
                             size_t workspace_bytes, void* stream);

/* The whole solve, device-resident: the `while` loop of iLQR_TF.optimize (quattro_ilqr_tf.py:428-472) — up to max_iter
 * iterations, every trajectory stopping on its own test (:472: no accepted step, or |cost_old - cost_new| < tol) — from ONE
 * host call with no host involvement in between.  For models with a persistent kernel (quattro_model_has_device_loop
 * != 0) it is ONE launch: the quadrotor (csrc/solve_quad.hip: a workgroup owns two trajectories from the first rollout to their
 * last accepted step and leaves when both have stopped), the cart-pole (csrc/solve_cartpole.hip: a 16-lane row per trajectory),
 * both integrators, and — in a user model's library — the compiled-in problem (csrc/solve_user.hip: one wave per trajectory
 * through the generic device bodies).  quattro_model_has_device_loop returns 1 where that launch is also the fastest form and 2
 * where it exists but enqueued iterations are faster (user models: the host mirror then enqueues unless asked otherwise); for
 * models without one (0) the same loop is enqueued as max_iter iterations of quattro_ilqr_iterate_f32, which skip stopped
 * trajectories.  Results are bit-identical to calling quattro_simulate_f32 once
 * and quattro_ilqr_iterate_f32 until every `active` flag is down.
 *   flags: QUATTRO_SOLVE_SIMULATE    roll the nominal out from x0 [B][n] first (x_nom, cost are outputs); without it x_nom and
 *                                    cost must hold the nominal rollout of u_nom and its cost (as after quattro_simulate_f32)
 *          QUATTRO_SOLVE_FIXED_ITERS ignore the stop flags: exactly max_iter iterations for every trajectory (benchmarking)
 *   in/out: u_nom [B][N][m]; active [B] (1 = solve this trajectory), iters [B] (incremented per iteration), as
 *           quattro_ilqr_iterate_f32;   out: x_nom, K, k, cost, alpha_idx, status (may be NULL)
 *   workspace: >= quattro_model_workspace_bytes(p, B, N), 256-byte aligned                                           */
#define QUATTRO_SOLVE_SIMULATE 1
#define QUATTRO_SOLVE_FIXED_ITERS 2
/*          QUATTRO_SOLVE_RESET       the call itself sets the per-solve state first (active = 1, iters = 0, alpha_idx = -1,
 *                                    status = 0) — what a caller otherwise does with four fills before a solve              */
#define QUATTRO_SOLVE_RESET 4
/*          QUATTRO_SOLVE_ENQUEUE     never the persistent kernel: max_iter enqueued iterations of quattro_ilqr_iterate_f32
 *                                    (what models with quattro_model_has_device_loop == 0 get anyway); without this flag a
 *                                    model whose persistent kernel exists but is slower (== 2) runs the persistent kernel
 *                                    only when QUATTRO_SOLVE_PERSISTENT is set                                              */
#define QUATTRO_SOLVE_ENQUEUE 8
#define QUATTRO_SOLVE_PERSISTENT 16
int quattro_model_has_device_loop(const quattro_model_params* p);
int quattro_ilqr_solve_f32(const quattro_model_params* p, const float* x0, float* x_nom, float* u_nom, int B, int N,
                           float reg, const float* alphas, int n_alpha, double tol, int max_iter, int flags, float* K,
                           float* k, double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Per-iteration log of a solve, written by the device: what iLQR_TF.optimize appends to self.logs every iteration
 * (quattro_ilqr_tf.py:453-466 pure / :565-578 hybrid: x_seq, u_seq, current_cost, k_seq, K_seq, alpha, new_cost, found_update;
 * new_x_seq / new_u_seq of iteration i are x_seq / u_seq of iteration i + 1 or the solve's result) and what its measure_time
 * decorators append to backward_pass_time / inference_time / forward_pass_time (:16-42) — so that a LOGGED solve is still one
 * launch and one download instead of a host round trip per iteration.
 *   records : device memory, B x capacity records of quattro_solve_log_record_bytes(n, m, N, flags) bytes, 16-byte aligned,
 *             zero-filled once by the caller; iteration i (0-based) of trajectory b is record b * capacity + (i % capacity)
 *   flags   : QUATTRO_LOG_TRAJ  (x_seq, u_seq entering the iteration), QUATTRO_LOG_GAINS (K, k of the iteration); the 64-byte
 *             header is always written:
 *               uint64 stamp[4]  s_memrealtime ticks (100 MHz): iteration begins / backward pass done / gains complete (equal to
 *                                the second in pure mode; after the predictor in hybrid mode) / line search done
 *               double cost_pre, cost_new   cost of the nominal entering the iteration / after it (unchanged if no step accepted)
 *               int32  alpha_idx            accepted step (index into `alphas`), -1 = none (found_update false)
 *               int32  iteration, 2 x int32 padding
 *   field offsets inside a record: quattro_solve_log_offset(n, m, N, flags, QUATTRO_LOG_FIELD_*)  (0 for a part the flags exclude) */
#define QUATTRO_LOG_TRAJ 1
#define QUATTRO_LOG_GAINS 2
#define QUATTRO_LOG_FIELD_X 0
#define QUATTRO_LOG_FIELD_U 1
#define QUATTRO_LOG_FIELD_K 2
#define QUATTRO_LOG_FIELD_KFF 3 /* k, the feed-forward term */
typedef struct quattro_solve_log {
  void* records;
  int32_t capacity;
  int32_t flags;
} quattro_solve_log;
size_t quattro_solve_log_record_bytes(int n, int m, int N, int flags);
size_t quattro_solve_log_offset(int n, int m, int N, int flags, int field);

/* quattro_ilqr_solve_f32 with a log (NULL = none: exactly quattro_ilqr_solve_f32).  The persistent kernels fill the records
 * between the phases of their loop; for models solved by enqueued iterations the record kernel below runs between the launches. */
int quattro_ilqr_solve_logged_f32(const quattro_model_params* p, const float* x0, float* x_nom, float* u_nom, int B, int N,
                                  float reg, const float* alphas, int n_alpha, double tol, int max_iter, int flags, float* K,
                                  float* k, double* cost, int32_t* alpha_idx, int32_t* active, int32_t* iters, int32_t* status,
                                  void* workspace, size_t workspace_bytes, const quattro_solve_log* log, void* stream);

/* The same records for a loop the CALLER enqueues kernel by kernel (hybrid iterations: sweep of the tail, predictor, line
 * search): one small launch per phase, trajectories selected on the device from `active` / `iters`, no host involvement.
 *   QUATTRO_LOG_PHASE_BEGIN          before the backward pass: header (stamp 0, cost_pre, iteration = iters[b]) + x, u for every
 *                                    trajectory with active[b] != 0 (or every one with `force`)
 *   QUATTRO_LOG_PHASE_BACKWARD_DONE  stamps 1 and 2;  QUATTRO_LOG_PHASE_GAINS_DONE  stamp 2 (after the predictor)
 *   QUATTRO_LOG_PHASE_END            after quattro_linesearch_f32 (which has incremented iters[b]): K, k, alpha_idx, cost_new,
 *                                    stamp 3 for the trajectories whose record iters[b] - 1 is pending                       */
#define QUATTRO_LOG_PHASE_BEGIN 0
#define QUATTRO_LOG_PHASE_BACKWARD_DONE 1
#define QUATTRO_LOG_PHASE_GAINS_DONE 2
#define QUATTRO_LOG_PHASE_END 3
int quattro_solve_log_record_f32(const quattro_solve_log* log, int phase, const float* x_nom, const float* u_nom,
                                 const float* K, const float* k, const double* cost, const int32_t* alpha_idx,
                                 const int32_t* active, const int32_t* iters, int B, int N, int n, int m, int force,
                                 void* stream);

/* Receding-horizon loop, device-resident: B controllers advance n_steps control steps in ONE launch (models with
 * quattro_model_has_device_loop; QUATTRO_ERR_UNSUPPORTED otherwise).  Per control step and controller, what
 * QuadrotorMPC.control_step (examples/quadrotor/quadrotor_mpc.py:102-124) and the simulator's loop around it do: solve from
 * the current state with the warm start (the loop above: nominal rollout, iterations until the stop test or max_iter), apply
 * u_0 to the plant — the device model itself; the reference's plant is MuJoCo, out of scope — plus an optional additive state
 * disturbance, shift the warm start (u_1 .. u_{N-1}, u_{N-1}), continue.  A controller never waits for another one.
 *   in/out: x_cur [B][n] current states (advanced n_steps times), u_nom [B][N][m] warm start (shifted result on return)
 *   out   : traj_x [B][n_steps+1][n] (traj_x[:,0] = the initial x_cur), traj_u [B][n_steps][m] applied controls,
 *           traj_iters [B][n_steps] iLQR iterations per control step; x_nom, K, k, cost, alpha_idx, status: of the LAST solve
 *   disturbance [n_steps][B][n] or NULL; active / iters [B]: scratch state of the loop (any contents on entry)            */
int quattro_mpc_run_f32(const quattro_model_params* p, float* x_cur, float* x_nom, float* u_nom, int B, int N, float reg,
                        const float* alphas, int n_alpha, double tol, int max_iter, int n_steps, float* traj_x, float* traj_u,
                        int32_t* traj_iters, const float* disturbance, float* K, float* k, double* cost, int32_t* alpha_idx,
                        int32_t* active, int32_t* iters, int32_t* status, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Transformer gain predictor: weights of the reference's TransformerPredictor (quattro_ilqr_tf/transformer_model.py:85-138)
 * as DEVICE pointers, plus the DataNormalizer vectors (:15-50).  Matrices are PyTorch Linear layout [out][in];
 * the `w_*` matrices are 16-bit (raw uint16 bit patterns: bf16, or IEEE half when `precision` says so), everything else
 * fp32.
 *   precision            : QUATTRO_TF_PRECISION_BF16 (the *_bf16 entry points) or QUATTRO_TF_PRECISION_F16 (the *_f16 entry
 *                          points: the same kernel with fp16 MFMA operands — what the shipped checkpoints store and what
 *                          the reference's own predict() computes in, transformer_ilqr.py:317-319; fp32 accumulation,
 *                          LayerNorm, softmax and residual stream either way)
 *   tok_bias             : unused (kept for layout compatibility; see tok_bias_t below)
 *   w_out    [64][d]     : output_linear.weight zero-padded to 64 rows
 * Supported shape family: d_model = 128, n_head = 4, d_ff % 64 == 0 (64 .. 1024), L <= 128, c_dim <= 64 (both shipped models). */
#define QUATTRO_TF_MAX_LAYERS 8
#define QUATTRO_TF_PRECISION_BF16 0
#define QUATTRO_TF_PRECISION_F16 1
typedef struct quattro_tf_weights {
  int32_t n_x, c_dim, d_model, n_head, d_ff, n_layers, n_state_tok, prompt_len, target_len, precision;
  const float *x_mean, *x_std, *u_mean, *u_std;
  const uint16_t* w_state; /* state_embed.weight as bf16 [d][16], columns >= n_x zero (one MFMA k-step) */
  const float *state_b, *ctrl_w, *ctrl_b;
  const float* tok_bias;
  const uint16_t* w_qkv[QUATTRO_TF_MAX_LAYERS]; /* in_proj_weight [3d][d] */
  const float* b_qkv[QUATTRO_TF_MAX_LAYERS];
  const uint16_t* w_o[QUATTRO_TF_MAX_LAYERS];   /* out_proj.weight [d][d] */
  const float* b_o[QUATTRO_TF_MAX_LAYERS];
  const uint16_t* w_1[QUATTRO_TF_MAX_LAYERS];   /* linear1.weight [ff][d] */
  const float* b_1[QUATTRO_TF_MAX_LAYERS];
  const uint16_t* w_2[QUATTRO_TF_MAX_LAYERS];   /* linear2.weight [d][ff] */
  const float* b_2[QUATTRO_TF_MAX_LAYERS];
  const float* ln1_g[QUATTRO_TF_MAX_LAYERS];
  const float* ln1_b[QUATTRO_TF_MAX_LAYERS];
  const float* ln2_g[QUATTRO_TF_MAX_LAYERS];
  const float* ln2_b[QUATTRO_TF_MAX_LAYERS];
  const uint16_t* w_out;
  const float* b_out;
  /* What the kernel actually reads (built from the arrays above, which stay in the reference's own layout):
   *   tok_bias_t [d][128] : tok_bias transposed and zero-padded to 128 tokens, with the embedding bias of the token's kind
   *                         folded in (state_b on state tokens, ctrl_b on prompt tokens); depends on n_state_tok
   *   w_stream            : every weight matrix as 1-KB MFMA fragments in the order the kernel consumes them
   *                         (quattro_tf_stream_elems bf16 elements, written by quattro_tf_pack_stream_bf16)
   *   p_stream            : biases / LayerNorm vectors / output de-normalisation per layer (quattro_tf_param_floats)      */
  const float* tok_bias_t;
  const uint16_t* w_stream;
  const float* p_stream;
} quattro_tf_weights;

/* Sizes of the two streams for the shape in `w` (0 = unsupported shape), and the packing itself: reads the PyTorch-layout
 * device arrays named in `w` (w_state, ctrl_w, w_qkv .. w_out, every bias / LayerNorm / normaliser vector) and writes
 * w_stream / p_stream (device buffers of the caller).  Run once per set of weights; pure data movement.               */
size_t quattro_tf_stream_elems(const quattro_tf_weights* w);
size_t quattro_tf_param_floats(const quattro_tf_weights* w);
int quattro_tf_pack_stream_bf16(const quattro_tf_weights* w, uint16_t* w_stream, float* p_stream, void* stream);

/* Batched predictor forward, bf16 MFMA with fp32 accumulation, one launch for the whole model (reads tok_bias_t, w_stream,
 * p_stream and the normaliser vectors x_mean — which a caller may point at a shifted mean — only).  Replaces
 * TransformerILQR.predict (quattro_ilqr_tf/transformer_ilqr.py:311-325: normalise, forward, de-normalise) around
 * TransformerPredictor.forward (transformer_model.py:122-138, PositionalEncoding :77-80) for B sequences at once.
 *   x_err  [B][n_state_tok][n_x] : x_seq - x_ref + state_offset (raw, un-normalised)
 *   prompt [B][P][c]             : [k | K.flat] rows of the swept tail (raw)
 *   pred   [B][T][c]             : de-normalised prediction                                                   */
int quattro_tf_forward_bf16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, float* pred,
                            void* stream);

/* The same forward with the prediction written straight into gain stacks: row t of the (T, c) prediction viewed as
 * (m, 1 + n) is k_t (column 0) and K_t (the rest), exactly the unpacking of iLQR_TF.optimize (quattro_ilqr_tf.py:510-514
 * / :536-540); c_dim must equal m (1 + n).  Rows t >= N of a prediction longer than the horizon are dropped (the
 * reference's forward_pass never reads them).  Replaces predict + reshape + slicing + the np.concatenate head of
 * :515-518 for a batch.
 *   K [B][N][m][n], k [B][N][m] : rows t < min(T, N) are overwritten, the swept tail rows are the caller's
 *   active [B] (may be NULL)    : trajectories with active[b] == 0 are skipped entirely
 *   prompt == NULL              : the P prompt rows [k | K.flat] (quattro_ilqr_tf.py:498-502) are read from rows N - P .. N - 1
 *                                 of K, k themselves — the tail quattro_linearize_sweep_rows_f32 (k_rows = N) has just swept;
 *                                 requires T + P <= N (the kernel must not write the rows it reads)                          */
int quattro_tf_gains_bf16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, int N, int n,
                          int m, float* K, float* k, const int32_t* active, void* stream);

/* The three entry points above with fp16 MFMA operands (w->precision == QUATTRO_TF_PRECISION_F16, `w_*` and w_stream
 * holding IEEE half bit patterns); same arguments, same kernel source (second instantiation: 238 instead of 196
 * registers, measured ~5 % slower at B = 4096; 12x closer to the reference module's fp32 output on the shipped
 * checkpoints).  Each family rejects a weight set of the other precision with QUATTRO_ERR_BAD_ARG.                                                                          */
int quattro_tf_pack_stream_f16(const quattro_tf_weights* w, uint16_t* w_stream, float* p_stream, void* stream);
int quattro_tf_forward_f16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, float* pred,
                           void* stream);
int quattro_tf_gains_f16(const quattro_tf_weights* w, const float* x_err, const float* prompt, int B, int N, int n, int m,
                         float* K, float* k, const int32_t* active, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training of the gain predictor, fp32, hand-written forward + backward + Adam.
 * Replaces the body of the mini-batch loop of TransformerILQR.fit (quattro_ilqr_tf/transformer_ilqr.py:150-172:
 * model(x, u_prompt) -> nn.MSELoss -> loss.backward() -> Adam.step()) around TransformerPredictor
 * (transformer_model.py:85-138).  Parameters, gradients and Adam moments are ONE flat fp32 device array each, in the
 * order of quattro_tf_train_param_offset (every block 16-byte aligned; padding floats stay zero); each block has the
 * reference module's own shape and layout (PyTorch [out][in] matrices), so a state dict is a set of slices of it.
 * Shapes: head dimension d_model / nhead <= 32 (the reference's default constructor is 64 / 8), d_model <= 512,
 * L = n_state_tok + prompt_len + target_len <= 128; anything else returns QUATTRO_ERR_UNSUPPORTED.                                                   */
typedef struct quattro_tf_train_desc {
  int32_t state_dim, control_dim, d_model, nhead, n_layers, d_ff;
  int32_t n_state_tok, prompt_len, target_len; /* tokens: N+1 states, P prompt rows, T learnable target rows */
  float dropout;                               /* transformer_model.py:97 (applied only when `training` != 0) */
} quattro_tf_train_desc;

/* blocks of the flat parameter array (reference state_dict name in the comment); the last twelve are per layer */
#define QUATTRO_TF_P_TARGET 0   /* target_embedding [T][d]                    */
#define QUATTRO_TF_P_STATE_W 1  /* state_embed.weight [d][n]                  */
#define QUATTRO_TF_P_STATE_B 2  /* state_embed.bias [d]                       */
#define QUATTRO_TF_P_CTRL_W 3   /* control_embed.weight [d][c]                */
#define QUATTRO_TF_P_CTRL_B 4   /* control_embed.bias [d]                     */
#define QUATTRO_TF_P_OUT_W 5    /* output_linear.weight [c][d]                */
#define QUATTRO_TF_P_OUT_B 6    /* output_linear.bias [c]                     */
#define QUATTRO_TF_P_QKV_W 7    /* layers.l.self_attn.in_proj_weight [3d][d]  */
#define QUATTRO_TF_P_QKV_B 8    /* layers.l.self_attn.in_proj_bias [3d]       */
#define QUATTRO_TF_P_O_W 9      /* layers.l.self_attn.out_proj.weight [d][d]  */
#define QUATTRO_TF_P_O_B 10     /* layers.l.self_attn.out_proj.bias [d]       */
#define QUATTRO_TF_P_FF1_W 11   /* layers.l.linear1.weight [ff][d]            */
#define QUATTRO_TF_P_FF1_B 12   /* layers.l.linear1.bias [ff]                 */
#define QUATTRO_TF_P_FF2_W 13   /* layers.l.linear2.weight [d][ff]            */
#define QUATTRO_TF_P_FF2_B 14   /* layers.l.linear2.bias [d]                  */
#define QUATTRO_TF_P_LN1_G 15   /* layers.l.norm1.weight [d]                  */
#define QUATTRO_TF_P_LN1_B 16   /* layers.l.norm1.bias [d]                    */
#define QUATTRO_TF_P_LN2_G 17   /* layers.l.norm2.weight [d]                  */
#define QUATTRO_TF_P_LN2_B 18   /* layers.l.norm2.bias [d]                    */
#define QUATTRO_TF_P_COUNT 19

/* floats in the flat array (0 = unsupported shape); float offset of a block (-1 = no such block) */
size_t quattro_tf_train_param_count(const quattro_tf_train_desc* desc);
long quattro_tf_train_param_offset(const quattro_tf_train_desc* desc, int which, int layer);
/* bytes of caller-provided scratch for a mini-batch of `batch` sequences (saved activations + gradient temporaries) */
size_t quattro_tf_train_workspace_bytes(const quattro_tf_train_desc* desc, int batch);

/* Forward (+ loss) (+ backward) of one mini-batch.
 *   x_norm [B][n_state_tok][n], prompt_norm [B][P][c], target_norm [B][T][c] : normalised inputs / targets
 *     (DataNormalizer, transformer_model.py:37-50; the slices of transformer_ilqr.py:112-118)
 *   pe [L][d]            : rows of the sinusoidal table pos_encoder.pe (a buffer of the module, not a parameter)
 *   training             : != 0 applies dropout (masks = hash of dropout_seed, site and element index)
 *   loss (may be NULL)   : device scalar, mean squared error over B*T*c
 *   pred_out (may be NULL) : [B][T][c] normalised prediction
 *   grads (may be NULL)  : flat gradient array, overwritten; NULL = forward (+ loss) only (evaluation on the test set,
 *                          transformer_ilqr.py:177-184)
 *   workspace            : 256-byte aligned, >= quattro_tf_train_workspace_bytes                                     */
int quattro_tf_train_step_f32(const quattro_tf_train_desc* desc, const float* params, float* grads, void* workspace,
                              size_t workspace_bytes, const float* x_norm, const float* prompt_norm,
                              const float* target_norm, const float* pe, int batch, uint64_t dropout_seed, int training,
                              float* loss, float* pred_out, void* stream);

/* torch.optim.Adam with its defaults as the reference uses it (transformer_ilqr.py:140; no weight decay, bias-corrected
 * moments), over the flat arrays; `step` counts from 1.                                                               */
int quattro_tf_adam_f32(float* params, const float* grads, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                        float eps, int step, void* stream);

/* Test hook: the dropout factors (0 or 1 / (1 - p)) the training step applies at `site` (0: positions; 1 + 4 l + k for
 * layer l: k = 0 attention weights [B][H][L][L], 1 out-projection [B L][d], 2 feed-forward hidden [B L][ff],
 * 3 feed-forward output [B L][d]) for elements 0 .. n-1.                                                              */
int quattro_tf_train_dropout_mask_f32(uint64_t dropout_seed, float p, int site, size_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QUATTRO_HIP_H */
