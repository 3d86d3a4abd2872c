// Cart-pole-shaped problems (n = 4, m = 1): device-side bodies shared by sweep_lane.hip (stand-alone launches) and
// solve_cartpole.hip (the device-resident solve loop).
//
//   cartpole_record         the derivative record of one step (ROWMAJOR layout) from (x_t, u_t): Euler closed form or the RK4
//                           forward-mode columns — the code every cart-pole sweep of this library shares, so their records are
//                           bit-identical
//   sweep16_cartpole_body   linearisation + Riccati-like sweep with SIXTEEN lanes per trajectory
//
// Arithmetic replaced: _compute_dynamics_jacobians / _compute_cost_derivatives / _finite_diff_*_final +
// iLQR_TF.backward_pass / backward_pass_segment (quattro_ilqr_tf/quattro_ilqr_tf.py:149-275, :290-317, :336-364).
//
// Why sixteen lanes.  BASELINE configs[1] (B = 1024) leaves the chip almost empty: a wave sits alone on its SIMD and issues
// one instruction per ~4-5 cycles, so a sweep's time IS its instruction count per step x 50 steps.  One lane per trajectory
// is ~450 instructions per step (43 us), a DPP quad per trajectory ~300 (33 us) — a third of them the linearisation, which
// does not depend on the value function at all.  Here
//   * the records of up to CH steps are formed AHEAD of the chain, one step per lane (16 at a time), into an LDS stage;
//   * the recursion runs on a 4 x 4 lane grid: lane (i, j) = 4 i + j of the trajectory's 16-lane DPP row owns V[i][j],
//     P[i][j], Q[i][j], V'[i][j].  Rows of V reach a lane as DPP quad broadcasts folded into the multiply-add
//     (v_fmac_f32_dpp), columns of P through ds_bpermute, V'^T (symmetrisation) and the row forms of K and Q_ux through one
//     transposing ds_bpermute each: ~90 instructions per step.
// Every dot product keeps the one-lane kernel's operand order (k = 0..3 ascending, same fmaf chain), so K, k are
// bit-identical to sweep_lane_cartpole_kernel / sweep_quad_cartpole_kernel.
#pragma once
#include "models_device.h"

namespace {

template <bool RK4>
__device__ __forceinline__ void cartpole_record(const quattro_model_params& p, const float* xs, const float* us,
                                                float* rec) {
  constexpr int MODEL = QUATTRO_MODEL_CARTPOLE, NX = 4, NU = 1, NZ = 5;
  using R = RowMajorRec<NX, NU>;
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
}

namespace cp16 {

constexpr int CH = 32;                                   // steps per LDS stage (two record passes of 16 lanes)
constexpr int RS = RowMajorRec<4, 1>::STRIDE;            // 48 floats per record
constexpr int STAGE_FLOATS = CH * RS + 16;               // per trajectory: 6 KB + 16 floats of skew — a step's record is 48 floats, the rows
                                                         // of a wave read theirs at the same offsets, and with stages a multiple of
                                                         // 32 floats apart two rows of a 32-lane group met on the same 16 banks
                                                         // (SQ_LDS_BANK_CONFLICT 45 % of the LDS cycles of the persistent kernel)

#define QT_QP16(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
// Row i of V times two columns of F (four entries each), V[i][k] taken from lane k of this lane's quad by the DPP modifier of
// the multiply-add itself (v_fmac_f32_dpp): the compiler does not fold a quad_perm move into a multiply-add, and eight
// broadcasts per step would double this part of the chain.  k ascending, one fmaf chain per accumulator — the one-lane
// kernel's order.  The leading s_nop covers the VALU-write -> DPP-read hazard on `v`, which the hazard recogniser cannot
// see inside inline asm.
__device__ __forceinline__ void row_dot2(float v, const float* c0, const float* c1, float& acc0, float& acc1) {
#define QT_FD(acc, b, l) "v_fmac_f32_dpp " acc ", %2, " b " quad_perm:[" l "," l "," l "," l "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  asm volatile("s_nop 1\n\t"
               QT_FD("%0", "%3", "0") QT_FD("%1", "%7", "0") QT_FD("%0", "%4", "1") QT_FD("%1", "%8", "1")
               QT_FD("%0", "%5", "2") QT_FD("%1", "%9", "2") QT_FD("%0", "%6", "3") QT_FD("%1", "%10", "3")
               : "+v"(acc0), "+v"(acc1)
               : "v"(v), "v"(c0[0]), "v"(c0[1]), "v"(c0[2]), "v"(c0[3]), "v"(c1[0]), "v"(c1[1]), "v"(c1[2]), "v"(c1[3]));
#undef QT_FD
}
template <int I>
__device__ __forceinline__ float quad_bcast16(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), QT_QP16(I, I, I, I), 0xf, 0xf, true));
}
__device__ __forceinline__ float bperm(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

// what lane (i, j) reads of a step's record
struct LaneRec {
  float Fc[4];   // F[k][j]  (column j of A)
  float Fr[4];   // F[k][i]  (column i of A)
  float FB[4];   // F[k][4]  (B)
  float Lij, luxj, luu, lxj, lu;
  __device__ __forceinline__ void load(const float* __restrict__ rec, int i, int j) {
    using R = RowMajorRec<4, 1>;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Fc[k] = rec[R::A + 4 * k + j];
      Fr[k] = rec[R::A + 4 * k + i];
    }
    const float4 b = *reinterpret_cast<const float4*>(rec + R::B);
    FB[0] = b.x; FB[1] = b.y; FB[2] = b.z; FB[3] = b.w;
    Lij = rec[R::LXX + 4 * i + j];
    luxj = rec[R::LUX + j];
    luu = rec[R::LUU];
    lxj = rec[R::LX + j];
    lu = rec[R::LU];
  }
};

}  // namespace cp16

// One trajectory per 16-lane DPP row: `lane` = lane id in the wave, trajectory `b` (of B; `live` = this row has one and it is to
// be swept — idle rows run along on trajectory 0 without storing, so that every row of the wave stays on one path),
// `stage` = this row's LDS stage (cp16::STAGE_FLOATS floats, 16-byte aligned).  Steps t_start .. N-1; outputs indexed
// t - t_start.  status may be NULL.
template <bool RK4>
__device__ __forceinline__ void sweep16_cartpole_body(const quattro_model_params& p, const float* __restrict__ x,
                                                      const float* __restrict__ u, int N, int t_start, float reg,
                                                      float* __restrict__ Kout, float* __restrict__ kout,
                                                      int32_t* __restrict__ status, const int b, const bool live,
                                                      const int lane, float* stage, const int k_rows = 0) {
  using namespace cp16;
  constexpr int NX = 4;
  const int sub = lane & 15, i = sub >> 2, j = sub & 3, row0 = lane & 48;
  const size_t bb = live ? b : 0;
  const int S = N - t_start;
  const int KR = k_rows > 0 ? k_rows : S;
  const float4* px = reinterpret_cast<const float4*>(x) + bb * (N + 1);
  const float* pu = u + bb * N;
  int gat[4];                                              // byte addresses for ds_bpermute: lane (k, j) of this row
#pragma unroll
  for (int k = 0; k < 4; ++k) gat[k] = 4 * (row0 + 4 * k + j);
  const int tra = 4 * (row0 + 4 * j + i);                  // the transposed lane (j, i)

  // terminal pair: V_x(N) = 2 Qf (x_N - x_ref), V_xx(N) = 2 Qf, used as given
  float Vij, vx[NX];
  {
    const float4 xN = px[N];
    const float xn[NX] = {xN.x, xN.y, xN.z, xN.w};
#pragma unroll
    for (int c = 0; c < NX; ++c) vx[c] = 2.0f * p.qf[c] * (xn[c] - p.x_ref[c]);
    Vij = (i == j) ? 2.0f * p.qf[i] : 0.0f;
  }
  bool bad = false, singular = false;

  auto step = [&](const LaneRec& r, int s) __attribute__((always_inline)) {
    // P = V F: own entry and the control column (rows of V by DPP quad broadcast inside the multiply-add)
    float Pij = 0.0f, Pi4 = 0.0f;
    row_dot2(Vij, r.Fc, r.FB, Pij, Pi4);
    // column j of P and the control column, all four rows
    float Pk[4], P4k[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Pk[k] = bperm(gat[k], Pij);
      P4k[k] = bperm(gat[k], Pi4);
    }
    float Qij = r.Lij, Quxj = r.luxj, Q44 = r.luu, qzj = r.lxj, qz4 = r.lu;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Qij = fmaf(r.Fr[k], Pk[k], Qij);
      Quxj = fmaf(r.FB[k], Pk[k], Quxj);
      Q44 = fmaf(r.FB[k], P4k[k], Q44);
      qzj = fmaf(r.Fc[k], vx[k], qzj);
      qz4 = fmaf(r.FB[k], vx[k], qz4);
    }
    // W = 1 / (Q_uu + reg)
    const float piv = Q44 + reg;
    singular = singular || !((piv != 0.0f) && qt_finite(piv));
    const float w = 1.0f * (1.0f / piv);
    const float Kj = -fmaf(w, Quxj, 0.0f), k4 = -fmaf(w, qz4, 0.0f);
    bad = bad || !qt_finite(Kj) || !qt_finite(k4);
    if (live && i == 0) Kout[(bb * KR + (KR - S) + s) * NX + j] = Kj;      // (KR rows per trajectory; step s at row KR - S + s)
    if (live && sub == 0) kout[bb * KR + (KR - S) + s] = k4;
    const float Gj = fmaf(Q44, Kj, Quxj), G4 = fmaf(Q44, k4, qz4);
    const float Ki = bperm(tra, Kj), Quxi = bperm(tra, Quxj);            // row forms: the transposed lane holds K[i], Q_ux[i]
    float Vn = Qij;
    Vn = fmaf(Ki, Gj, Vn);
    Vn = fmaf(Quxi, Kj, Vn);
    float vxj = qzj;
    vxj = fmaf(Kj, G4, vxj);
    vxj = fmaf(Quxj, k4, vxj);
    Vij = 0.5f * (Vn + bperm(tra, Vn));
    vx[0] = quad_bcast16<0>(vxj);
    vx[1] = quad_bcast16<1>(vxj);
    vx[2] = quad_bcast16<2>(vxj);
    vx[3] = quad_bcast16<3>(vxj);
  };

  for (int base = ((S - 1) / CH) * CH; base >= 0; base -= CH) {
    const int cnt = S - base < CH ? S - base : CH;
    // records of steps base .. base + cnt - 1, one step per lane, 16 at a time (the previous stage's reads are complete: the
    // chain below ends with ds_bpermute / LDS reads of this same wave, in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int ls = sub; ls < cnt; ls += 16) {
      const int t = t_start + base + ls;
      const float4 xq = px[t];
      const float xs[4] = {xq.x, xq.y, xq.z, xq.w};
      const float us[1] = {pu[t]};
      float rec[RS];
      cartpole_record<RK4>(p, xs, us, rec);
      float4* dst = reinterpret_cast<float4*>(stage + ls * RS);
#pragma unroll
      for (int q = 0; q < RS / 4; ++q) dst[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the chain: records double-buffered in registers, the next one requested before the current step runs
    LaneRec r0, r1;
    r0.load(stage + (cnt - 1) * RS, i, j);
    int ls = cnt - 1;
    for (; ls >= 1; ls -= 2) {
      r1.load(stage + (ls - 1) * RS, i, j);
      step(r0, base + ls);
      r0.load(stage + (ls >= 2 ? ls - 2 : 0) * RS, i, j);
      step(r1, base + ls - 1);
    }
    if (ls == 0) step(r0, base);
  }
  // a trajectory's flags: any lane of its row
  const unsigned long long badm = __ballot(bad), sinm = __ballot(singular);
  if (status != nullptr && live && sub == 0)
    status[bb] = (((badm >> row0) & 0xffffull) ? QUATTRO_TRAJ_NONFINITE : 0) | (((sinm >> row0) & 0xffffull) ? QUATTRO_TRAJ_SINGULAR : 0);
}

#undef QT_QP16

}  // namespace
