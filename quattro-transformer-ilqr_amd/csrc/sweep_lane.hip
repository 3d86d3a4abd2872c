// Cart-pole-shaped problems (n = 4, m = 1): linearisation + Riccati-like sweep with ONE LANE PER TRAJECTORY.
//
// Arithmetic replaced: _compute_dynamics_jacobians / _compute_cost_derivatives / _finite_diff_*_final +
// iLQR_TF.backward_pass / backward_pass_segment (quattro_ilqr_tf/quattro_ilqr_tf.py:149-275, :290-317, :336-364).
//
// A 4 x 4 value function, a 4 x 5 [A | B] and a scalar Q_uu are ~150 multiply-adds per step: handing such a problem a
// whole 64-lane wave (sweep_generic_kernel: LDS-staged blocks, seven workgroup barriers per step) spends the step on
// synchronisation, not on arithmetic — 42 us for B = 1024, N = 50, plus a 13 us linearisation launch in front.  Here
// everything of a trajectory lives in its lane's registers: the lane forms the step's derivative record with the SAME
// device-model code the record kernels use (EulerRecord::fill_const / fill_state, or the RK4 forward-mode columns),
// runs the recursion on it with the generic kernel's formulas term for term (the gains agree with
// quattro_linearize_f32 + quattro_riccati_sweep_f32 through ROWMAJOR records to fp32 round-off), and requests (x, u)
// of the next step one step ahead.  No record buffer, no LDS, no barrier; B = 1024 is 16 waves whose 50-step chains run side by side.
#include "models_device.h"

namespace {

template <bool RK4>
__global__ __launch_bounds__(QT_WAVE) void sweep_lane_cartpole_kernel(const quattro_model_params p,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ u, int B, int N,
                                                                       int t_start, float reg, float* __restrict__ Kout,
                                                                       float* __restrict__ kout,
                                                                       int32_t* __restrict__ status,
                                                                       const int32_t* __restrict__ active) {
  constexpr int MODEL = QUATTRO_MODEL_CARTPOLE, NX = 4, NU = 1, NZ = 5;
  using R = RowMajorRec<NX, NU>;
  const int b = blockIdx.x * QT_WAVE + threadIdx.x;
  if (b >= B) return;
  if (active != nullptr && active[b] == 0) return;
  const int S = N - t_start;
  const float4* px = reinterpret_cast<const float4*>(x) + (size_t)b * (N + 1);
  const float* pu = u + (size_t)b * N;

  // terminal pair (terminal_kernel): V_x(N) = 2 Qf (x_N - x_ref), V_xx(N) = 2 Qf, used as given
  float V[NX][NX], vx[NX];
  {
    const float4 xN = px[N];
    const float xn[NX] = {xN.x, xN.y, xN.z, xN.w};
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      vx[i] = 2.0f * p.qf[i] * (xn[i] - p.x_ref[i]);
#pragma unroll
      for (int j = 0; j < NX; ++j) V[i][j] = (i == j) ? 2.0f * p.qf[i] : 0.0f;
    }
  }
  bool bad = false, singular = false;
  float4 xq = px[t_start + S - 1];
  float uq = pu[t_start + S - 1];
  for (int s = S - 1; s >= 0; --s) {
    const float xs[NX] = {xq.x, xq.y, xq.z, xq.w};
    const float us[NU] = {uq};
    if (s > 0) {                                             // next (earlier) step's inputs fly during this step
      xq = px[t_start + s - 1];
      uq = pu[t_start + s - 1];
    }
    // ---- the step's derivative record, exactly as the record kernels produce it
    float rec[R::STRIDE];
#pragma unroll
    for (int i = 0; i < R::STRIDE; ++i) rec[i] = 0.0f;
    if constexpr (!RK4) {
      EulerRecord<MODEL, R>::fill_const(rec, p);
      EulerRecord<MODEL, R>::fill_state(rec, p, xs, us);
    } else {                                                 // linearize_rk4_kernel, one direction after the other
      const float dt = p.dt;
#pragma unroll
      for (int j = 0; j < NZ; ++j) {
        float dx0[NX], du[NU], k[NX], dk[NX], xst[NX], dxs[NX], acc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) dx0[i] = (i == j) ? 1.0f : 0.0f;
        du[0] = (j == NX) ? 1.0f : 0.0f;
        qt_rate<MODEL>(p, xs, us, k);
        qt_rate_jvp<MODEL>(p, xs, us, dx0, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = dk[i]; xst[i] = fmaf(0.5f * dt, k[i], xs[i]); dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
        qt_rate<MODEL>(p, xst, us, k);
        qt_rate_jvp<MODEL>(p, xst, us, dxs, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); xst[i] = fmaf(0.5f * dt, k[i], xs[i]); dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
        qt_rate<MODEL>(p, xst, us, k);
        qt_rate_jvp<MODEL>(p, xst, us, dxs, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); xst[i] = fmaf(dt, k[i], xs[i]); dxs[i] = fmaf(dt, dk[i], dx0[i]); }
        qt_rate_jvp<MODEL>(p, xst, us, dxs, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const float v = fmaf(dt / 6.0f, acc[i] + dk[i], dx0[i]);
          if (j < NX) rec[R::a(i, j < NX ? j : 0)] = v;
          else rec[R::b(i, 0)] = v;
        }
      }
      fill_cost_entries<MODEL, R>(rec, p, xs, us);
    }
    auto F = [&](int k, int j) -> float { return j < NX ? rec[R::a(k, j)] : rec[R::b(k, j - NX)]; };
    // ---- the recursion: sweep_generic_kernel's formulas and fmaf order, term for term
    float P[NX][NZ], Q[NZ][NZ], qz[NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j < NZ; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) acc = fmaf(V[i][kk], F(kk, j), acc);
        P[i][j] = acc;
      }
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
      for (int j = 0; j < NZ; ++j) {
        if (i < NX && j >= NX) { Q[i][j] = 0.0f; continue; }
        float acc = (i < NX) ? rec[R::lxx(i < NX ? i : 0, j < NX ? j : 0)]
                             : (j < NX ? rec[R::lux(0, j < NX ? j : 0)] : rec[R::luu(0, 0)]);
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) acc = fmaf(F(kk, i), P[kk][j], acc);
        Q[i][j] = acc;
      }
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
      float acc = (j < NX) ? rec[R::lx(j < NX ? j : 0)] : rec[R::lu(0)];
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) acc = fmaf(F(kk, j), vx[kk], acc);
      qz[j] = acc;
    }
    // W = 1 / (Q_uu + reg)  (invert_gj<1>)
    const float piv = Q[NX][NX] + reg;
    singular = singular || !((piv != 0.0f) && qt_finite(piv));
    const float w = 1.0f * (1.0f / piv);
    float Kk[NX + 1];                                        // [K | k]
#pragma unroll
    for (int c = 0; c <= NX; ++c) {
      const float q = (c < NX) ? Q[NX][c < NX ? c : 0] : qz[NX];
      float acc = fmaf(w, q, 0.0f);
      acc = -acc;
      bad = bad || !qt_finite(acc);
      Kk[c] = acc;
    }
    *reinterpret_cast<float4*>(Kout + ((size_t)b * S + s) * NX) = make_float4(Kk[0], Kk[1], Kk[2], Kk[3]);
    kout[(size_t)b * S + s] = Kk[NX];
    float G[NX + 1];                                         // Q_uu [K | k] + [Q_ux | Q_u]
#pragma unroll
    for (int c = 0; c <= NX; ++c) {
      float acc = (c < NX) ? Q[NX][c < NX ? c : 0] : qz[NX];
      G[c] = fmaf(Q[NX][NX], Kk[c], acc);
    }
    float Vn[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        float acc = Q[i][j];
        acc = fmaf(Kk[i], G[j], acc);
        acc = fmaf(Q[NX][i], Kk[j], acc);
        Vn[i][j] = acc;
      }
      float acc = qz[i];
      acc = fmaf(Kk[i], G[NX], acc);
      acc = fmaf(Q[NX][i], Kk[NX], acc);
      vx[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int j = 0; j < NX; ++j) V[i][j] = 0.5f * (Vn[i][j] + Vn[j][i]);
  }
  if (status != nullptr) status[b] = (bad ? QUATTRO_TRAJ_NONFINITE : 0) | (singular ? QUATTRO_TRAJ_SINGULAR : 0);
}

}  // namespace

int quattro_launch_sweep_lane_cartpole(const quattro_model_params& p, const float* x, const float* u, int B, int N,
                                       int t_start, float reg, float* K, float* k, int32_t* status,
                                       const int32_t* active, hipStream_t stream) {
  const int blocks = (B + QT_WAVE - 1) / QT_WAVE;
  if (p.integrator == QUATTRO_INTEGRATOR_RK4)
    hipLaunchKernelGGL(sweep_lane_cartpole_kernel<true>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg,
                       K, k, status, active);
  else
    hipLaunchKernelGGL(sweep_lane_cartpole_kernel<false>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg,
                       K, k, status, active);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
