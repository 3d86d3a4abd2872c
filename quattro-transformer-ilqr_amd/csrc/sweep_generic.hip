// Generic Riccati-like backward sweep: one wavefront per trajectory, derivative blocks staged in LDS.
//
// Arithmetic replaced: iLQR_TF.backward_pass / backward_pass_segment of the reference
// (quattro_ilqr_tf/quattro_ilqr_tf.py:297-317, :343-364), formula for formula (4-term V update with the
// un-regularised Q_uu, explicit inverse of Q_uu + reg I with partial pivoting like LAPACK getrf/getri,
// V_xx symmetrised every step, terminal V_xx used as given).
//
// Works for any (NX, NU) instantiated below from ROWMAJOR records.  Each step the wave
//   1. streams the record (coalesced 16-B loads, prefetched one step ahead in registers) into LDS,
//   2. forms P = V_xx [A|B], Q = L_zz + [A|B]^T P, q = l_z + [A|B]^T V_x with the 64 lanes striding over outputs,
//   3. inverts Q_uu + reg I redundantly in registers (unrolled Gauss-Jordan, partial pivoting),
//   4. writes K, k, and updates V_x, V_xx.
// It is the semantic baseline; the quadrotor shape has a faster specialisation in sweep_tile16.hip.
#include "sweep_generic_body.h"

namespace {

template <int NX, int NU>
__global__ __launch_bounds__(QT_WAVE) void sweep_generic_kernel(const float* __restrict__ rec,
                                                                const float* __restrict__ VxN,
                                                                const float* __restrict__ VxxN, int S, float reg,
                                                                float* __restrict__ Kout, float* __restrict__ kout,
                                                                int32_t* __restrict__ status,
                                                                const int32_t* __restrict__ active) {
  const int b = blockIdx.x;
  if (active != nullptr && active[b] == 0) return;
  sweep_generic_body<NX, NU>(rec, VxN, VxxN, S, reg, Kout, kout, status, b, (int)threadIdx.x);
}

}  // namespace

int quattro_launch_sweep_generic(const float* rec, const float* VxN, const float* VxxN, int B, int S, int n, int m,
                                 float reg, float* K, float* k, int32_t* status, const int32_t* active,
                                 hipStream_t stream) {
  if (n == 4 && m == 1) {
    hipLaunchKernelGGL((sweep_generic_kernel<4, 1>), dim3(B), dim3(QT_WAVE), 0, stream, rec, VxN, VxxN, S, reg, K, k,
                       status, active);
  } else if (n == 12 && m == 4) {
    hipLaunchKernelGGL((sweep_generic_kernel<12, 4>), dim3(B), dim3(QT_WAVE), 0, stream, rec, VxN, VxxN, S, reg, K, k,
                       status, active);
#ifdef QT_USER_MODEL_HEADER
  } else if (n == QT_USER_NX && m == QT_USER_NU) {
    hipLaunchKernelGGL((sweep_generic_kernel<QT_USER_NX, QT_USER_NU>), dim3(B), dim3(QT_WAVE), 0, stream, rec, VxN, VxxN, S,
                       reg, K, k, status, active);
#endif
  } else {
    return QUATTRO_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
