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
#include "cartpole_body.h"

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
    cartpole_record<RK4>(p, xs, us, rec);
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

// ---------------------------------------------------------------------------------------------------------------
// The same recursion with FOUR lanes (one DPP quad) per trajectory: a lone wave issues one instruction per ~4 cycles, so
// the one-lane kernel's ~450 instructions per step ARE its step time (B = 1024 is 16 waves on 1024 SIMDs; 43 us).  Lane
// j of a quad owns column j of P = V F and of Q = L_zz + F^T P (and their control column, which all four lanes carry),
// its entry of K, its column of V' — and, under RK4, direction j of the forward-mode linearisation (the one-lane kernel
// pushes all five directions through the four stages one after the other: ~1 500 of its instructions).  Every lane
// keeps the whole of V, V_x and F (16 + 4 + 20 registers): columns computed by one lane reach the others as DPP
// quad_perm broadcasts, no LDS.  Formulas and fmaf order are the one-lane kernel's (= sweep_generic_kernel's).
#define QT_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int I>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), QT_QP(I, I, I, I), 0xf, 0xf, true));
}
__device__ __forceinline__ float sel4(int j, float a0, float a1, float a2, float a3) {
  const float lo = (j & 1) ? a1 : a0, hi = (j & 1) ? a3 : a2;
  return (j & 2) ? hi : lo;
}

template <bool RK4>
__global__ __launch_bounds__(QT_WAVE) void sweep_quad_cartpole_kernel(const quattro_model_params p,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ u, int B, int N,
                                                                       int t_start, float reg, float* __restrict__ Kout,
                                                                       float* __restrict__ kout,
                                                                       int32_t* __restrict__ status,
                                                                       const int32_t* __restrict__ active) {
  constexpr int MODEL = QUATTRO_MODEL_CARTPOLE, NX = 4, NU = 1, NZ = 5;
  using R = RowMajorRec<NX, NU>;
  const int gid = blockIdx.x * QT_WAVE + threadIdx.x;
  const int b = gid >> 2, j = gid & 3;
  if (b >= B) return;                                        // (whole quads leave together: the exchanges stay inside a quad)
  if (active != nullptr && active[b] == 0) return;
  const int S = N - t_start;
  const float4* px = reinterpret_cast<const float4*>(x) + (size_t)b * (N + 1);
  const float* pu = u + (size_t)b * N;

  float V[NX][NX], vx[NX];
  {
    const float4 xN = px[N];
    const float xn[NX] = {xN.x, xN.y, xN.z, xN.w};
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      vx[i] = 2.0f * p.qf[i] * (xn[i] - p.x_ref[i]);
#pragma unroll
      for (int c = 0; c < NX; ++c) V[i][c] = (i == c) ? 2.0f * p.qf[i] : 0.0f;
    }
  }
  bool bad = false, singular = false;
  float4 xq = px[t_start + S - 1];
  float uq = pu[t_start + S - 1];
  for (int s = S - 1; s >= 0; --s) {
    const float xs[NX] = {xq.x, xq.y, xq.z, xq.w};
    const float us[NU] = {uq};
    if (s > 0) {
      xq = px[t_start + s - 1];
      uq = pu[t_start + s - 1];
    }
    // ---- the step's derivative record: F = [A | B] whole in every lane, the cost entries of the own column
    float F[NX][NZ];
    float rec[R::STRIDE];
#pragma unroll
    for (int i = 0; i < R::STRIDE; ++i) rec[i] = 0.0f;
    if constexpr (!RK4) {
      EulerRecord<MODEL, R>::fill_const(rec, p);
      EulerRecord<MODEL, R>::fill_state(rec, p, xs, us);
#pragma unroll
      for (int k = 0; k < NX; ++k)
#pragma unroll
        for (int c = 0; c < NZ; ++c) F[k][c] = c < NX ? rec[R::a(k, c < NX ? c : 0)] : rec[R::b(k, 0)];
    } else {
      // direction j (a state) by lane j and the control direction by every lane, through the four stages
      // (linearize_rk4_kernel's arithmetic); the stage points are the same for all directions
      const float dt = p.dt;
      float k1[NX], k2[NX], k3[NX], x2[NX], x3[NX], x4[NX];
      qt_rate<MODEL>(p, xs, us, k1);
#pragma unroll
      for (int i = 0; i < NX; ++i) x2[i] = fmaf(0.5f * dt, k1[i], xs[i]);
      qt_rate<MODEL>(p, x2, us, k2);
#pragma unroll
      for (int i = 0; i < NX; ++i) x3[i] = fmaf(0.5f * dt, k2[i], xs[i]);
      qt_rate<MODEL>(p, x3, us, k3);
#pragma unroll
      for (int i = 0; i < NX; ++i) x4[i] = fmaf(dt, k3[i], xs[i]);
      float own[NX], ctl[NX];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        float dx0[NX], du[NU], dk[NX], dxs[NX], acc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) dx0[i] = (d == 0 && i == j) ? 1.0f : 0.0f;
        du[0] = d == 1 ? 1.0f : 0.0f;
        qt_rate_jvp<MODEL>(p, xs, us, dx0, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = dk[i]; dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
        qt_rate_jvp<MODEL>(p, x2, us, dxs, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
        qt_rate_jvp<MODEL>(p, x3, us, dxs, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); dxs[i] = fmaf(dt, dk[i], dx0[i]); }
        qt_rate_jvp<MODEL>(p, x4, us, dxs, du, dk);
#pragma unroll
        for (int i = 0; i < NX; ++i) (d == 0 ? own : ctl)[i] = fmaf(dt / 6.0f, acc[i] + dk[i], dx0[i]);
      }
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        F[k][0] = quad_bcast<0>(own[k]);
        F[k][1] = quad_bcast<1>(own[k]);
        F[k][2] = quad_bcast<2>(own[k]);
        F[k][3] = quad_bcast<3>(own[k]);
        F[k][4] = ctl[k];
      }
      fill_cost_entries<MODEL, R>(rec, p, xs, us);
    }
    float Fj[NX], lxxj[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      Fj[k] = sel4(j, F[k][0], F[k][1], F[k][2], F[k][3]);
      lxxj[k] = sel4(j, rec[R::lxx(k, 0)], rec[R::lxx(k, 1)], rec[R::lxx(k, 2)], rec[R::lxx(k, 3)]);
    }
    const float luxj = sel4(j, rec[R::lux(0, 0)], rec[R::lux(0, 1)], rec[R::lux(0, 2)], rec[R::lux(0, 3)]);
    const float lxj = sel4(j, rec[R::lx(0)], rec[R::lx(1)], rec[R::lx(2)], rec[R::lx(3)]);
    // ---- own column of P and Q, the control column of both, own entry of q_z
    float Pj[NX], P4[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      float a = 0.0f, c4 = 0.0f;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) {
        a = fmaf(V[i][kk], Fj[kk], a);
        c4 = fmaf(V[i][kk], F[kk][NX], c4);
      }
      Pj[i] = a;
      P4[i] = c4;
    }
    float Qj[NZ];                                            // Q[0..3][j] and Q[4][j] = Q_ux[j]
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      float a = i < NX ? lxxj[i < NX ? i : 0] : luxj;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) a = fmaf(F[kk][i], Pj[kk], a);
      Qj[i] = a;
    }
    float Q44 = rec[R::luu(0, 0)], qzj = lxj, qz4 = rec[R::lu(0)];
#pragma unroll
    for (int kk = 0; kk < NX; ++kk) {
      Q44 = fmaf(F[kk][NX], P4[kk], Q44);
      qzj = fmaf(Fj[kk], vx[kk], qzj);
      qz4 = fmaf(F[kk][NX], vx[kk], qz4);
    }
    const float piv = Q44 + reg;
    singular = singular || !((piv != 0.0f) && qt_finite(piv));
    const float w = 1.0f * (1.0f / piv);
    const float Kj = -fmaf(w, Qj[NX], 0.0f), kk4 = -fmaf(w, qz4, 0.0f);
    bad = bad || !qt_finite(Kj) || !qt_finite(kk4);
    Kout[((size_t)b * S + s) * NX + j] = Kj;
    if (j == 0) kout[(size_t)b * S + s] = kk4;
    const float Gj = fmaf(Q44, Kj, Qj[NX]), G4 = fmaf(Q44, kk4, qz4);
    // ---- own column of V', own entry of V_x'; then everybody gets everything
    const float K0 = quad_bcast<0>(Kj), K1 = quad_bcast<1>(Kj), K2 = quad_bcast<2>(Kj), K3 = quad_bcast<3>(Kj);
    const float U0 = quad_bcast<0>(Qj[NX]), U1 = quad_bcast<1>(Qj[NX]), U2 = quad_bcast<2>(Qj[NX]), U3 = quad_bcast<3>(Qj[NX]);
    const float Ki[NX] = {K0, K1, K2, K3}, Ui[NX] = {U0, U1, U2, U3};
    float Vnj[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) Vnj[i] = fmaf(Ui[i], Kj, fmaf(Ki[i], Gj, Qj[i]));
    const float vxj = fmaf(Qj[NX], kk4, fmaf(Kj, G4, qzj));
    float Vn[NX][NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      Vn[i][0] = quad_bcast<0>(Vnj[i]);
      Vn[i][1] = quad_bcast<1>(Vnj[i]);
      Vn[i][2] = quad_bcast<2>(Vnj[i]);
      Vn[i][3] = quad_bcast<3>(Vnj[i]);
    }
    vx[0] = quad_bcast<0>(vxj);
    vx[1] = quad_bcast<1>(vxj);
    vx[2] = quad_bcast<2>(vxj);
    vx[3] = quad_bcast<3>(vxj);
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int c = 0; c < NX; ++c) V[i][c] = 0.5f * (Vn[i][c] + Vn[c][i]);
  }
  // a trajectory's flags: any lane of its quad
  const unsigned long long badm = __ballot(bad), sinm = __ballot(singular);
  if (status != nullptr && j == 0) {
    const int q4 = threadIdx.x & ~3;
    status[b] = (((badm >> q4) & 0xfull) ? QUATTRO_TRAJ_NONFINITE : 0) | (((sinm >> q4) & 0xfull) ? QUATTRO_TRAJ_SINGULAR : 0);
  }
}
#undef QT_QP

// sixteen lanes (one DPP row) per trajectory, four trajectories per wave: cartpole_body.h
template <bool RK4>
__global__ __launch_bounds__(QT_WAVE) void sweep16_cartpole_kernel(const quattro_model_params p, const float* __restrict__ x,
                                                                    const float* __restrict__ u, int B, int N, int t_start,
                                                                    float reg, float* __restrict__ Kout,
                                                                    float* __restrict__ kout, int32_t* __restrict__ status,
                                                                    const int32_t* __restrict__ active, int k_rows) {
  __shared__ __attribute__((aligned(16))) float s_stage[4 * cp16::STAGE_FLOATS];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 4 + (lane >> 4);
  const bool live = b < B && (active == nullptr || active[b < B ? b : 0] != 0);
  if (!__any(live)) return;
  sweep16_cartpole_body<RK4>(p, x, u, N, t_start, reg, Kout, kout, status, b, live, lane, s_stage + (lane >> 4) * cp16::STAGE_FLOATS,
                             k_rows);
}

}  // namespace

int quattro_launch_sweep_lane_cartpole(const quattro_model_params& p, const float* x, const float* u, int B, int N,
                                       int t_start, float reg, float* K, float* k, int k_rows, int32_t* status,
                                       const int32_t* active, hipStream_t stream) {
#if !defined(QT_CARTPOLE_ONE_LANE) && !defined(QT_CARTPOLE_QUAD)
  const int blocks = (B + 3) / 4;                          // sixteen lanes per trajectory
  if (p.integrator == QUATTRO_INTEGRATOR_RK4)
    hipLaunchKernelGGL(sweep16_cartpole_kernel<true>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg, K, k,
                       status, active, k_rows);
  else
    hipLaunchKernelGGL(sweep16_cartpole_kernel<false>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg, K,
                       k, status, active, k_rows);
#elif !defined(QT_CARTPOLE_ONE_LANE)
  if (k_rows > 0 && k_rows != N - t_start) return QUATTRO_ERR_UNSUPPORTED;   // (the in-place row form exists in the default kernel only)
  const int blocks = (4 * B + QT_WAVE - 1) / QT_WAVE;      // four lanes per trajectory
  if (p.integrator == QUATTRO_INTEGRATOR_RK4)
    hipLaunchKernelGGL(sweep_quad_cartpole_kernel<true>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg,
                       K, k, status, active);
  else
    hipLaunchKernelGGL(sweep_quad_cartpole_kernel<false>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg,
                       K, k, status, active);
#else
  if (k_rows > 0 && k_rows != N - t_start) return QUATTRO_ERR_UNSUPPORTED;
  const int blocks = (B + QT_WAVE - 1) / QT_WAVE;
  if (p.integrator == QUATTRO_INTEGRATOR_RK4)
    hipLaunchKernelGGL(sweep_lane_cartpole_kernel<true>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg,
                       K, k, status, active);
  else
    hipLaunchKernelGGL(sweep_lane_cartpole_kernel<false>, dim3(blocks), dim3(QT_WAVE), 0, stream, p, x, u, B, N, t_start, reg,
                       K, k, status, active);
#endif
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
