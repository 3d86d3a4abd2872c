// Quadrotor-shaped Riccati-like sweep (n = 12, m = 4, n + m = 16): one wavefront per trajectory, the whole
// recursion kept in the wave's registers as 16x16 tiles.
//
// Arithmetic replaced: iLQR_TF.backward_pass / backward_pass_segment of the reference
// (quattro_ilqr_tf/quattro_ilqr_tf.py:297-317, :343-364).
//
// Tile layout (the C/D layout of v_mfma_f32_16x16x4_f32): lane l = 16r + c holds elements [4r + s][c], s = 0..3,
// of a 16x16 matrix over the augmented index z = (x, u); tile index 4g + s' is x_{3g+s'} for s' < 3 and u_g for
// s' = 3 (Tile16Rec in quattro_device.h).  With that interleaving
//   * [A|B] (12 x 16) is three registers per lane, read straight from the TILE16 record (coalesced, no LDS),
//   * P = V_xx [A|B] and Q = L_zz + [A|B]^T P are 3 + 3 exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32 is a k-ordered
//     fmaf chain, bit-identical to VALU fp32): the accumulator of the first product is already the B operand of
//     the second, and the symmetric V_xx in accumulator layout is already the A operand of the next step,
//   * control row u_r of Q, (Q_ux | Q_uu)[r][:], sits in accumulator register 3 of lane group r: a 4 x 16 matrix
//     with one element per lane.  (Q_uu + reg I)^-1 [Q_ux | Q_u] is a Gauss-Jordan elimination on that matrix in
//     place: the pivot row is broadcast across lane groups with ds_bpermute, the pivot column across the 16 lanes of
//     a group with a DPP row_newbcast, the pivot itself with v_readlane.  What is left in the state columns is
//     K (up to sign), already in the one-element-per-lane operand layout of the rank-4 update
//     V_xx' = Q_xx + (Q_ux - reg K)^T K, which is 1 more MFMA.
// One LDS round trip per step transposes V_xx' (symmetrisation) and redistributes V_x'.
//
// V update: the reference computes Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K with K = -(Q_uu + reg I)^-1 Q_ux.
// Since (Q_uu + reg I) K = -Q_ux exactly, Q_uu K + Q_ux = -reg K, so the same quantity is
// Q_xx + (Q_ux - reg K)^T K (and V_x' = Q_x + (Q_ux - reg K)^T k): algebraically identical, one product
// instead of three, and free of the fp32 cancellation in Q_uu K + Q_ux.
// The elimination does not pivot (LAPACK's inverse in the reference does): identical in exact arithmetic, and stable for
// the symmetric positive definite Q_uu + reg I that a convex cost gives (both built-in models).  A zero pivot raises
// QUATTRO_TRAJ_SINGULAR.  Records that come from outside (plain TILE16 layout: quattro_pack_derivs_f32 callers) may
// hold an indefinite or badly scaled Q_uu: that instantiation also compares every pivot with the diagonal entry it
// started from — pivot <= 1e-6 |diagonal| (the cancellation that produced it lost six digits) or pivot <= 0 (not
// positive definite: elimination without pivoting has no stability guarantee) raises QUATTRO_TRAJ_ILLCOND, and the
// caller re-runs those trajectories through the generic kernel, which pivots (ops.riccati_sweep does so itself).
// Parity vs the reference's outputs: tests/test_kernels_gpu.py.
#include "sweep_tile16_body.h"

namespace {

template <int MODE>
__global__ __launch_bounds__(QT_WAVE * WPB, 4) void sweep_tile16_kernel(const float* __restrict__ rec,
                                                               const float* __restrict__ VxN,
                                                               const float* __restrict__ VxxN, int S, float reg,
                                                               float* __restrict__ Kout, float* __restrict__ kout,
                                                               int32_t* __restrict__ status,
                                                               const int32_t* __restrict__ active,
                                                               const FusedArgs fa QT_SWEEP_DBG_PARAM) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * WPB + wv;
  const int lane = threadIdx.x & 63;
  if (WPB > 1 && b >= fa.B) return;
  if (active != nullptr && active[b] == 0) return;
  __shared__ __attribute__((aligned(16))) float s_t_all[WPB * 16 * LD];
  __shared__ __attribute__((aligned(16))) float s_vx_all[WPB * 64];
  // MODE_FUSED: [header record (TILE16) | FUSED_BATCH compact records (TILE16F)]
  constexpr int LIN_FLOATS = sweep_lin_floats<MODE>();
  __shared__ __attribute__((aligned(16))) float s_lin_all[WPB * LIN_FLOATS];
#ifdef QT_SWEEP_PROFILE
  sweep_tile16_body<MODE>(rec, VxN, VxxN, S, reg, Kout, kout, status, fa, b, lane, s_t_all + wv * 16 * LD, s_vx_all + wv * 64,
                          s_lin_all + wv * LIN_FLOATS, dbg);
#else
  sweep_tile16_body<MODE>(rec, VxN, VxxN, S, reg, Kout, kout, status, fa, b, lane, s_t_all + wv * 16 * LD, s_vx_all + wv * 64,
                          s_lin_all + wv * LIN_FLOATS);
#endif
}

}  // namespace

#ifdef QT_SWEEP_PROFILE
extern "C" int quattro_sweep_profile(const float* rec, const float* VxN, const float* VxxN, int B, int S, float reg,
                                     float* K, float* k, int compact, unsigned long long* dbg, void* stream) {
  FusedArgs none{};
  none.B = B;
  none.k_rows = 0;
  none.rn = 12;
  none.rm = 4;
  if (compact)
    hipLaunchKernelGGL(sweep_tile16_kernel<MODE_COMPACT>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, (hipStream_t)stream, rec, VxN, VxxN, S,
                       reg, K, k, nullptr, nullptr, none, dbg);
  else
    hipLaunchKernelGGL(sweep_tile16_kernel<MODE_TILE16>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, (hipStream_t)stream, rec, VxN, VxxN, S,
                       reg, K, k, nullptr, nullptr, none, dbg);
  return (int)hipGetLastError();
}
#else
int quattro_launch_sweep_tile16(const float* rec, const float* VxN, const float* VxxN, int B, int S, float reg,
                                float* K, float* k, int32_t* status, const int32_t* active, int layout, int n, int m,
                                hipStream_t stream) {
  FusedArgs none{};
  none.B = B;
  none.k_rows = 0;
  none.rn = n;
  none.rm = m;
  if (layout == QUATTRO_LAYOUT_ROWMAJOR_TILE) {
    if (n < 1 || n > 12 || m < 1 || m > 4) return QUATTRO_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sweep_tile16_kernel<MODE_ROWPAD>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, stream, rec, VxN, VxxN, S, reg, K, k,
                       status, active, none);
  } else if (layout == QUATTRO_LAYOUT_TILE16C)
    hipLaunchKernelGGL(sweep_tile16_kernel<MODE_COMPACT>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, stream, rec, VxN, VxxN, S, reg, K, k,
                       status, active, none);
  else if (layout == QUATTRO_LAYOUT_TILE16R)
    hipLaunchKernelGGL(sweep_tile16_kernel<MODE_DENSEF>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, stream, rec, VxN, VxxN, S, reg, K, k,
                       status, active, none);
  else
    hipLaunchKernelGGL(sweep_tile16_kernel<MODE_TILE16>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, stream, rec, VxN, VxxN, S, reg, K, k,
                       status, active, none);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

// linearise + sweep in one launch (Euler quadrotor): steps t_start .. N-1 of every trajectory
int quattro_launch_sweep_fused(const quattro_model_params& p, const float* x, const float* u, int B, int N, int t_start,
                               float reg, float* K, float* k, int k_rows, int32_t* status, const int32_t* active,
                               hipStream_t stream) {
  FusedArgs fa;
  fa.p = p;
  fa.x = x;
  fa.u = u;
  fa.N = N;
  fa.t_start = t_start;
  fa.B = B;
  fa.coef = nullptr;
  fa.k_rows = k_rows;
  fa.rn = 12;
  fa.rm = 4;
  hipLaunchKernelGGL(sweep_tile16_kernel<MODE_FUSED>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, stream, nullptr, nullptr, nullptr,
                     N - t_start, reg, K, k, status, active, fa);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

// linearise + sweep in one launch, RK4 quadrotor: `coef` = global scratch of B * (N - t_start) * Rk4Coef::STRIDE floats
size_t quattro_sweep_fused_rk4_scratch_floats(int B, int S) { return (size_t)B * S * Rk4Coef::STRIDE; }
int quattro_launch_sweep_fused_rk4(const quattro_model_params& p, const float* x, const float* u, int B, int N, int t_start,
                                   float reg, float* K, float* k, int k_rows, int32_t* status, const int32_t* active,
                                   float* coef, hipStream_t stream) {
  FusedArgs fa;
  fa.p = p;
  fa.x = x;
  fa.u = u;
  fa.N = N;
  fa.t_start = t_start;
  fa.B = B;
  fa.coef = coef;
  fa.k_rows = k_rows;
  fa.rn = 12;
  fa.rm = 4;
  hipLaunchKernelGGL(sweep_tile16_kernel<MODE_FUSED_RK4>, dim3((B + WPB - 1) / WPB), dim3(QT_WAVE * WPB), 0, stream, nullptr, nullptr,
                     nullptr, N - t_start, reg, K, k, status, active, fa);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
#endif
