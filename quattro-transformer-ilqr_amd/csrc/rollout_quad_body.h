// Device-side bodies of the quadrotor rollouts (four lanes — one DPP quad — per trajectory / line-search candidate), shared
// by rollout_quad.hip (one launch per call) and solve_quad.hip (the device-resident solve loop).  Design notes:
// rollout_quad.hip.
#pragma once
#include "models_device.h"

namespace {

struct AlphaList {
  float a[QUATTRO_MAX_ALPHAS];
};

constexpr int NX = 12, NU = 4, CS = 16;   // candidate record: x'_{t+1} (12) | u'_t (4)

// Diagnostic build only (-DQT_ABLATE_LS=n, scripts/ablate_linesearch.sh): one segment of a rollout step is left out (wrong numbers
// on purpose) to see its real share of the kernels' time.  1: the gain product K dx; 2: the stage cost and its fp64 accumulation;
// 3: the rate function (trig and all); 4: the candidate / state stores; 5: the per-step nominal loads (line search) ; 6: the
// commit copy of the accepted candidate.  The shipped library is built without it.
#ifndef QT_ABLATE_LS
#define QT_ABLATE_LS 0
#endif

#ifndef QT_ROLLOUT_PRIO
#define QT_ROLLOUT_PRIO 0
#endif
#ifndef QT_ROLLOUT_PRIO_LO
#define QT_ROLLOUT_PRIO_LO 0
#endif

#define QT_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int I>
__device__ __forceinline__ float quad_bcast(float v) {   // value of lane I of this quad, in every lane of the quad
  return quad_perm<QT_QP(I, I, I, I)>(v);
}
__device__ __forceinline__ double quad_sum(double v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  return v;
}

// per-lane constants, selected once from the model parameters
struct LaneConst {
  bool is0, is1;          // axis 0 / axis 1 (else axis 2); lane 3 counts as axis 0
  int a, j;
  float qw[4], qfw[4], xr[4];   // weights / reference of the own states (p_a, v_a, angle_a, omega_a); 0 weights on lane 3
  float rj;                     // R weight of control j
  float tc[4];                  // omega'_a = gy * (omega_b omega_c) + sum_i tc[i] u_i     (torque row / inertia)
  float gy, gz, inv_mass, dt;
  __device__ __forceinline__ float sel(float v0, float v1, float v2) const { return is0 ? v0 : (is1 ? v1 : v2); }
};

__device__ __forceinline__ LaneConst lane_const(const quattro_model_params& p, int j) {
  LaneConst L;
  L.j = j;
  L.a = j == 3 ? 0 : j;
  L.is0 = L.a == 0;
  L.is1 = L.a == 1;
  const float live = j == 3 ? 0.0f : 1.0f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    L.qw[g] = live * L.sel(p.q[3 * g], p.q[3 * g + 1], p.q[3 * g + 2]);
    L.qfw[g] = live * L.sel(p.qf[3 * g], p.qf[3 * g + 1], p.qf[3 * g + 2]);
    L.xr[g] = L.sel(p.x_ref[3 * g], p.x_ref[3 * g + 1], p.x_ref[3 * g + 2]);
  }
  L.rj = p.r[j];   // (indexed, not a chain of selects: the compiler turns those into a phi of POINTERS into the parameter block,
                   //  which pins a private copy of the whole block in scratch memory inside the fused solve kernel)
  const float Ix = p.phys[1], Iy = p.phys[2], Iz = p.phys[3], arm = p.phys[4], kyaw = p.phys[6];
  // tau_phi = arm((u1+u2)-(u0+u3)), tau_theta = arm((u0+u1)-(u2+u3)), tau_psi = kyaw(u0-u1+u2-u3)   (quadrotor_dynamics.py:139-154)
  const float s0 = L.sel(-1.0f, 1.0f, 1.0f), s1 = L.sel(1.0f, 1.0f, -1.0f), s2 = L.sel(1.0f, -1.0f, 1.0f),
              s3 = L.sel(-1.0f, -1.0f, -1.0f);
  const float gain = L.sel(arm / Ix, arm / Iy, kyaw / Iz);
  L.tc[0] = s0 * gain; L.tc[1] = s1 * gain; L.tc[2] = s2 * gain; L.tc[3] = s3 * gain;
  L.gy = L.sel((Iy - Iz) / Ix, (Iz - Ix) / Iy, (Ix - Iy) / Iz);
  L.gz = L.sel(0.0f, 0.0f, -p.phys[5]);
  L.inv_mass = 1.0f / p.phys[0];
  L.dt = p.dt;
  return L;
}

// the four controls of the quad (lane j owns u_j), in every lane: broadcast ONCE per step and shared by the rate
// function (four calls under RK4) and the store
struct QuadU {
  float u0, u1, u2, u3;
  __device__ __forceinline__ explicit QuadU(float uo)
      : u0(quad_bcast<0>(uo)), u1(quad_bcast<1>(uo)), u2(quad_bcast<2>(uo)), u3(quad_bcast<3>(uo)) {}
};

#ifndef QT_RK4_STAGE_TRIG
#define QT_RK4_STAGE_TRIG true
#endif

// time derivative of the own states xo = (p_a, v_a, angle_a, omega_a) given the quad's controls
// sin / cos of the own angle at an RK4 stage point from the step's base values by angle addition: the stage angle is the base
// angle + delta with |delta| = O(dt * rate), so sin(delta), cos(delta) are short Taylor polynomials (odd to delta^7, even to
// delta^6: below 1 ulp of the result for |delta| <= 0.25; beyond that — a tumbling candidate — the full reduction runs, behind a
// wave-uniform branch).  A full qt_sincos is ~25 instructions on the serial chain, this is 11.
struct TrigBase {
  float a, s, c;      // base angle and its sin / cos
};
__device__ __forceinline__ void stage_sincos(const TrigBase& b, float x, float* s, float* c) {
  const float d = x - b.a;
  if (__builtin_expect(__any(!(fabsf(d) <= 0.25f)), 0)) {
    qt_sincos(x, s, c);
    return;
  }
  const float z = d * d;
  const float sd = d * fmaf(z, fmaf(z, fmaf(z, -1.9841270e-4f, 8.3333333e-3f), -1.6666667e-1f), 1.0f);
  const float cd = fmaf(z, fmaf(z, fmaf(z, -1.3888889e-3f, 4.1666667e-2f), -0.5f), 1.0f);
  *s = fmaf(b.s, cd, b.c * sd);
  *c = fmaf(b.c, cd, -(b.s * sd));
}

template <bool STAGE>
__device__ __forceinline__ void quad_rate(const LaneConst& L, const float* xo, const QuadU& U, float* xd, TrigBase& tb) {
  const float u0 = U.u0, u1 = U.u1, u2 = U.u2, u3 = U.u3;
  const float tm = (u0 + u1 + u2 + u3) * L.inv_mass;
  const float tau = fmaf(L.tc[3], u3, fmaf(L.tc[2], u2, fmaf(L.tc[1], u1, L.tc[0] * u0)));
  float so, co;
  if constexpr (STAGE) {
    stage_sincos(tb, xo[2], &so, &co);
  } else {
    qt_sincos_chain(xo[2], &so, &co);
    tb.a = xo[2]; tb.s = so; tb.c = co;
  }
  const float sph = quad_bcast<0>(so), cph = quad_bcast<0>(co), sth = quad_bcast<1>(so), cth = quad_bcast<1>(co),
              sps = quad_bcast<2>(so), cps = quad_bcast<2>(co);
  const float y = __builtin_amdgcn_rcpf(cth);
  const float sec = fmaf(fmaf(-cth, y, 1.0f), y, y);       // 1 / cos(theta): reciprocal + one Newton step
  const float tth = sth * sec;
  const float wp = quad_bcast<0>(xo[3]), wq = quad_bcast<1>(xo[3]), wr = quad_bcast<2>(xo[3]);
  xd[0] = xo[1];
  // v' = (T/m) [sps sph + cps sth cph, cps sph - sps sth cph, cth cph] - [0, 0, g]
  const float ca = L.sel(sps, cps, 0.0f), cb = L.sel(cps * sth, -(sps * sth), cth);
  xd[1] = fmaf(tm, fmaf(ca, sph, cb * cph), L.gz);
  // Euler-angle rates
  const float mix = fmaf(wq, sph, wr * cph);
  xd[2] = L.sel(fmaf(mix, tth, wp), fmaf(wq, cph, -(wr * sph)), mix * sec);
  // body rates
  xd[3] = fmaf(L.gy, L.sel(wq * wr, wp * wr, wp * wq), tau);
}

template <bool RK4>
__device__ __forceinline__ void quad_step(const LaneConst& L, const float* xo, const QuadU& uo, float* xn) {
  const float dt = L.dt;
  float k1[4];
  TrigBase tb;
  quad_rate<false>(L, xo, uo, k1, tb);
  if constexpr (!RK4) {
#pragma unroll
    for (int g = 0; g < 4; ++g) xn[g] = fmaf(dt, k1[g], xo[g]);
    return;
  }
  float k2[4], k3[4], k4[4], xs[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xs[g] = fmaf(0.5f * dt, k1[g], xo[g]);
  quad_rate<QT_RK4_STAGE_TRIG>(L, xs, uo, k2, tb);
#pragma unroll
  for (int g = 0; g < 4; ++g) xs[g] = fmaf(0.5f * dt, k2[g], xo[g]);
  quad_rate<QT_RK4_STAGE_TRIG>(L, xs, uo, k3, tb);
#pragma unroll
  for (int g = 0; g < 4; ++g) xs[g] = fmaf(dt, k3[g], xo[g]);
  quad_rate<QT_RK4_STAGE_TRIG>(L, xs, uo, k4, tb);
#pragma unroll
  for (int g = 0; g < 4; ++g) xn[g] = xo[g] + (dt / 6.0f) * (k1[g] + 2.0f * k2[g] + 2.0f * k3[g] + k4[g]);
}

// this lane's share of L(x,u): its four state terms, its control's R and barrier terms.  `counted` = lane belongs to a
// rollout whose result is used (the wave-uniform barrier shortcut must not be vetoed by idle quads), as a lane mask the caller
// takes once, outside its loop (__builtin_amdgcn_ballot_w64(counted)).
__device__ __forceinline__ float lane_stage_cost(const quattro_model_params& p, const LaneConst& L, const float* xo,
                                                 float uo, unsigned long long counted) {
  float c = 0.0f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float d = xo[g] - L.xr[g];
    c = fmaf(L.qw[g] * d, d, c);
  }
  c = fmaf(L.rj * uo, uo, c);
  if (p.barrier_alpha != 0.0f) {
    // exact shortcut, per lane (see qt_stage_cost): with beta*u > 20 the lane's barrier term is below
    // alpha * 4.25e-18 / beta^2; when that is under half an ulp of c the fmaf below returns c unchanged
    const float ib = 1.0f / p.barrier_beta;
    // (the wave-wide test as lane masks of the two raw compares combined on the scalar unit: `__all(a && b)` makes the
    // compiler materialise the conjunction as a 0/1 VGPR and compare it again)
    const unsigned long long big = __builtin_amdgcn_ballot_w64(p.barrier_beta * uo > 20.0f);
    const unsigned long long tiny = __builtin_amdgcn_ballot_w64(fabsf(p.barrier_alpha) * 4.25e-18f * ib * ib < c * 2.9e-8f);
    if ((counted & ~(big & tiny)) != 0ull) {
      const float sp = qt_softplus(-uo, p.barrier_beta);
      c = fmaf(p.barrier_alpha, sp * sp, c);
    }
  }
  return c;
}

__device__ __forceinline__ float lane_final_cost(const LaneConst& L, const float* xo) {
  float c = 0.0f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float d = xo[g] - L.xr[g];
    c = fmaf(L.qfw[g] * d, d, c);
  }
  return c;
}

// Where a wave's nominal data lives: the arrays of the wave's trajectories as wave-uniform buffer resources plus this lane's
// loop-invariant 32-bit byte offsets; the step is a SCALAR offset.  (Round 4: as 64-bit pointers per lane the four address
// computations were ~8 of a step's ~165 vector instructions, in a kernel that is bound by exactly those.)
// `wb` = the wave's first trajectory (wave-uniform), `li` = this lane's trajectory minus wb (0 for a lane that only runs along),
// `nwt` = trajectories a wave spans.
struct NomSrc {
  __amdgpu_buffer_rsrc_t rx, ru, rK, rk;
  int vx, vu, vK;
  __device__ __forceinline__ NomSrc(const LaneConst& L, const float* x_nom, const float* u_nom, const float* K, const float* k,
                                    int N, int wb, int li, int nwt) {
    const size_t w = (size_t)wb;
    rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x_nom) + w * (N + 1) * NX, 0, nwt * (N + 1) * NX * 4, 0x00020000);
    ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(u_nom) + w * N * NU, 0, nwt * N * NU * 4, 0x00020000);
    rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K) + w * N * NU * NX, 0, nwt * N * NU * NX * 4, 0x00020000);
    rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k) + w * N * NU, 0, nwt * N * NU * 4, 0x00020000);
    vx = 4 * (li * (N + 1) * NX + L.a);
    vu = 4 * (li * N * NU + L.j);
    vK = 4 * NX * (li * N * NU + L.j);
  }
};

// nominal data of one step as this lane needs it: own states, own control, own gain row
struct NomLane {
  float x[4], u, k;
  float4 K[3];
  __device__ __forceinline__ void load(const NomSrc& s, int t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) x[g] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(s.rx, s.vx + 12 * g, t * (NX * 4), 0));
    u = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(s.ru, s.vu, t * (NU * 4), 0));
    k = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(s.rk, s.vu, t * (NU * 4), 0));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const auto q = __builtin_amdgcn_raw_buffer_load_b128(s.rK, s.vK + 16 * i, t * (NU * NX * 4), 0);
      K[i] = make_float4(__int_as_float(q[0]), __int_as_float(q[1]), __int_as_float(q[2]), __int_as_float(q[3]));
    }
  }
};

// Stores.  One 16-byte store per lane per step: the x' row (12 floats, state 3g+a in lane a / register g) is transposed
// inside the quad so that lane j < 3 holds x'[4j .. 4j+3] and lane 3 holds (u'_0..u'_3).  Per-state dword stores were
// 5 requests of 12-16 bytes per quad and step; at 2048 waves that request rate, not bandwidth, cost 45 us per line
// search once the caches were cold.  Stores are predicated by `en`, never branched around a rollout: the DPP exchanges
// and the wave-uniform barrier shortcut need every quad of the wave on the same path, so idle quads (alpha slot >=
// n_alpha, inactive trajectory) run along on valid data and simply do not store.
__device__ __forceinline__ float4 gather_quarter(const LaneConst& L, const QuadU& U, const float* xn) {
  const float t0 = L.sel(xn[0], xn[1], xn[2]);                                       // own lane: states 0, 4, 8
  const float t1 = quad_perm<QT_QP(1, 2, 0, 3)>(L.sel(xn[3], xn[0], xn[1]));         // states 1, 5, 9
  const float t2 = quad_perm<QT_QP(2, 0, 1, 3)>(L.sel(xn[2], xn[3], xn[0]));         // states 2, 6, 10
  const float t3 = L.sel(xn[1], xn[2], xn[3]);                                       // own lane: states 3, 7, 11
  const bool l3 = L.j == 3;
  return make_float4(l3 ? U.u0 : t0, l3 ? U.u1 : t1, l3 ? U.u2 : t2, l3 ? U.u3 : t3);
}

struct NoStore {
  __device__ __forceinline__ void operator()(const LaneConst&, int, const QuadU&, const float*) const {}
};
struct ArrayStore {     // x_new [N+1][12], u_new [N][4]
  float* base;          // lanes 0..2: x_new + 12 + 4j (row t+1 at + 12 t); lane 3: u_new (row t at + 4 t)
  int stride;
  bool en;
  __device__ __forceinline__ ArrayStore(const LaneConst& L, float* xo, float* uo, bool en_)
      : base(L.j < 3 ? xo + NX + 4 * L.j : uo), stride(L.j < 3 ? NX : NU), en(en_) {}
  __device__ __forceinline__ void operator()(const LaneConst& L, int t, const QuadU& U, const float* xnext) const {
    const float4 v = gather_quarter(L, U, xnext);
    if (en) *reinterpret_cast<float4*>(base + (size_t)t * stride) = v;
  }
};
// Candidate records of the fused line search [t][CS], AXIS-major: lane a < 3 writes its own four states as they sit in
// its registers — floats 4a .. 4a+3 = x'[a], x'[3+a], x'[6+a], x'[9+a] — and lane 3 the four controls: no transposition
// inside the quad on the serial chain (it was ~16 of a step's ~175 vector instructions, for all six candidates); the
// one candidate that is accepted is put back into natural order by the copy that commits it.
struct ScratchStore {
  float* s;
  bool en, l3;
  __device__ __forceinline__ ScratchStore(const LaneConst& L, float* rec, bool en_) : s(rec + 4 * L.j), en(en_), l3(L.j == 3) {}
  __device__ __forceinline__ void operator()(const LaneConst&, int t, const QuadU& U, const float* xnext) const {
    const float4 v = make_float4(l3 ? U.u0 : xnext[0], l3 ? U.u1 : xnext[1], l3 ? U.u2 : xnext[2], l3 ? U.u3 : xnext[3]);
    if (en) *reinterpret_cast<float4*>(s + (size_t)t * CS) = v;
  }
};

// 12-term dot product K_j,: dx with dx spread over the axis lanes: state 3g+a lives in lane a, register g.
// Written as v_fmac_f32_dpp (the broadcast is the DPP modifier of the multiply-add's first source): the compiler does
// not fold a quad_perm move into v_fmac, and 12 extra moves plus their hazard no-ops were 9 % of a step.  The leading
// s_nop covers the VALU-write -> DPP-read hazard for dx, which the hazard recogniser cannot see inside inline asm.
__device__ __forceinline__ float gain_dot(const float4* K, const float* dx, float acc) {
#define QT_FD(k, d, l) "v_fmac_f32_dpp %0, " d ", " k " quad_perm:[" l "," l "," l "," l "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  asm volatile("s_nop 1\n\t"
               QT_FD("%5", "%1", "0") QT_FD("%6", "%1", "1") QT_FD("%7", "%1", "2")
               QT_FD("%8", "%2", "0") QT_FD("%9", "%2", "1") QT_FD("%10", "%2", "2")
               QT_FD("%11", "%3", "0") QT_FD("%12", "%3", "1") QT_FD("%13", "%3", "2")
               QT_FD("%14", "%4", "0") QT_FD("%15", "%4", "1") QT_FD("%16", "%4", "2")
               : "+v"(acc)
               : "v"(dx[0]), "v"(dx[1]), "v"(dx[2]), "v"(dx[3]), "v"(K[0].x), "v"(K[0].y), "v"(K[0].z), "v"(K[0].w),
                 "v"(K[1].x), "v"(K[1].y), "v"(K[1].z), "v"(K[1].w), "v"(K[2].x), "v"(K[2].y), "v"(K[2].z), "v"(K[2].w));
#undef QT_FD
  return acc;
}

// One closed-loop rollout by a quad.  Returns this LANE's partial of sum_t L + Lf (fp64); quad_sum() gives the total.
template <bool RK4, int PF, class Store>
__device__ __forceinline__ double quad_rollout_closed(const quattro_model_params& p, const LaneConst& L,
                                                      const NomSrc& src, float alpha, int N, bool counted, Store store) {
  // nominal data is requested PF steps ahead (PF register buffers, loop unrolled by PF): the loads come from HBM / the
  // far cache (K was written by the sweep, 39 MB per 4096 trajectories) and a step is only ~0.5 us of issue.  PF = 4 in
  // the stand-alone kernels; 2 inside the device-resident solve loop, where the gains were written by the same workgroup a
  // moment ago (L2-resident) and the loop shares its 128 registers with the sweep.
  const unsigned long long counted_mask = __builtin_amdgcn_ballot_w64(counted);
  NomLane nb[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) nb[d].load(src, d < N ? d : N - 1);
  float xh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xh[g] = nb[0].x[g];
  double J = 0.0;
  auto step = [&](const NomLane& b, int t) __attribute__((always_inline)) {
#if QT_ROLLOUT_PRIO != 0
    __builtin_amdgcn_s_setprio(QT_ROLLOUT_PRIO);
#endif
    float dx[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) dx[g] = xh[g] - b.x[g];
#if QT_ABLATE_LS != 1
    const float du = gain_dot(b.K, dx, b.k);
#else
    const float du = b.k + dx[0] * b.K[0].x;
#endif
    const float uh = fmaf(alpha, du, b.u);
#if QT_ROLLOUT_PRIO == 0
#if QT_ABLATE_LS != 2
    J += (double)lane_stage_cost(p, L, xh, uh, counted_mask);
#endif
#endif
    float xnext[4];
    const QuadU U(uh);
#if QT_ABLATE_LS != 3
    quad_step<RK4>(L, xh, U, xnext);
#else
    for (int g = 0; g < 4; ++g) xnext[g] = fmaf(L.dt, U.u0 + xh[(g + 1) & 3], xh[g]);
#endif
#if QT_ROLLOUT_PRIO != 0
    // (experiment: the state recurrence at a higher wave priority than the cost, the store and the loads that hang off it)
    __builtin_amdgcn_s_setprio(QT_ROLLOUT_PRIO_LO);
    J += (double)lane_stage_cost(p, L, xh, uh, counted_mask);
#endif
#if QT_ABLATE_LS != 4
    store(L, t, U, xnext);
#else
    if (t == 0) store(L, t, U, xnext);
#endif
#pragma unroll
    for (int g = 0; g < 4; ++g) xh[g] = xnext[g];
  };
  int t = 0;
  for (; t + PF <= N; t += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      step(nb[d], t + d);
#if QT_ABLATE_LS != 5
      nb[d].load(src, t + d + PF < N ? t + d + PF : N - 1);
#endif
    }
  }
#pragma unroll
  for (int d = 0; d < PF - 1; ++d)
    if (t + d < N) step(nb[d], t + d);          // wave-uniform tail (N % PF steps)
  J += (double)lane_final_cost(L, xh);
  return J;
}

// simulate: quad per trajectory, 16 trajectories per wave
// `gid` = 4 * trajectory + lane-in-quad; `live` = this quad has a trajectory to roll out (idle quads run along on
// trajectory 0 without storing: the DPP exchanges and the wave-uniform barrier shortcut need every quad on the same path)
template <bool RK4>
__device__ __forceinline__ void simulate_quad_body(const quattro_model_params& p, const float* __restrict__ x0,
                                                   const float* __restrict__ u, int N, float* __restrict__ x,
                                                   double* __restrict__ cost, const int gid, const bool live) {
  const int b = gid >> 2;
  const size_t bb = live ? b : 0;
  const LaneConst L = lane_const(p, gid & 3);
  const float* ub = u + bb * N * NU + L.j;
  float* xo = x + bb * (N + 1) * NX + L.a;
  float* xrow = x + bb * (N + 1) * NX + 4 * (L.j < 3 ? L.j : 0);      // this lane's 16-byte quarter of a state row
  float xh[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xh[g] = x0[bb * NX + 3 * g + L.a];
  const bool writer = live && L.j < 3;
  const int bw = __builtin_amdgcn_readfirstlane(b);         // the wave's first trajectory (gid grows with the lane)
  const __amdgpu_buffer_rsrc_t rsx =
      __builtin_amdgcn_make_buffer_rsrc(x + (size_t)bw * (N + 1) * NX, 0, 16 * (N + 1) * NX * 4, 0x00020000);
  const int vox = writer ? 4 * ((b - bw) * (N + 1) * NX + L.a) : -1;
  if (writer) {
#pragma unroll
    for (int g = 0; g < 4; ++g) xo[3 * g] = xh[g];
  }
  double J = 0.0;
  const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
  float u0 = ub[0], u1 = ub[(size_t)(N > 1 ? 1 : 0) * NU];
  auto step = [&](float ut, int t) __attribute__((always_inline)) {
#if QT_ABLATE_LS != 2
    J += (double)lane_stage_cost(p, L, xh, ut, live_mask);
#endif
    float xn[4];
    const QuadU U(ut);
#if QT_ABLATE_LS != 3
    quad_step<RK4>(L, xh, U, xn);
#else
    for (int g = 0; g < 4; ++g) xn[g] = fmaf(L.dt, U.u0 + xh[(g + 1) & 3], xh[g]);
#endif
#if QT_ABLATE_LS != 4
#ifdef QT_SIM_GATHER_STORE
    const float4 row = gather_quarter(L, U, xn);
    if (writer) *reinterpret_cast<float4*>(xrow + (size_t)(t + 1) * NX) = row;
#else
    // Round 4: the lane's four states go out as they sit in its registers (x[t+1][3g + a], four dword stores).  The in-quad
    // transposition that made one 16-byte store per lane of them (gather_quarter: two DPP permutes and ten selects) was 27 %
    // of this kernel by the ablation of scripts/ablate_linesearch.sh — a lone wave per SIMD pays for every instruction it issues,
    // and at 256 waves the request rate that made dword stores expensive in the line search (2048 waves) is no issue.
    // ... through buffer stores: the wave's 16 trajectories are one wave-uniform resource, the row a scalar offset, and a lane
    // that writes nothing (the quad's control lane, a lane past the batch) carries an out-of-range offset the range check drops.
    // No exec-mask branch around the stores — which matters beyond the branch: with the stores behind a branch the compiler's
    // wait-count pass must assume they may NOT have been issued, and the wait for the next control (loaded BEFORE them) comes out
    // as vmcnt(1) instead of vmcnt(5) — every step then waits for the previous step's stores to be acknowledged.
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(xn[g]), rsx, vox, (t + 1) * (NX * 4) + 12 * g, 0);
#endif
#else
    if (t == 0 && writer) *reinterpret_cast<float4*>(xrow + NX) = gather_quarter(L, U, xn);
#endif
#pragma unroll
    for (int g = 0; g < 4; ++g) xh[g] = xn[g];
  };
  // Everything loaded so far (the lane's constants, x0, the first two controls) lands HERE, through the builtin the compiler's
  // wait-count pass understands: otherwise the loop header inherits those loads as pending and every step carries the decreasing
  // `s_waitcnt vmcnt(4..1)` the first one needs — which from the second step on wait for the PREVIOUS step's four state stores to
  // be acknowledged (the counter retires in order): a store round trip on the chain of every step.
  // (the empty asm "uses" the two prefetched controls, so their loads cannot sink below the wait)
  asm volatile("" : "+v"(u0), "+v"(u1));
  __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0), nothing else
  int t = 0;
  for (; t + 1 < N; t += 2) {
    step(u0, t);
    u0 = ub[(size_t)(t + 2 < N ? t + 2 : N - 1) * NU];
    step(u1, t + 1);
    u1 = ub[(size_t)(t + 3 < N ? t + 3 : N - 1) * NU];
  }
  if (t < N) step(u0, t);
  J += (double)lane_final_cost(L, xh);
  J = quad_sum(J);
  if (live && L.j == 0 && cost != nullptr) cost[b] = J;
}

// Fused line search: 8 candidate quads = 32 lanes per trajectory, 2 trajectories per wave.  Every candidate leaves its
// (x', u') in the scratch; after the ballot the trajectory's 32 lanes copy the accepted candidate over the nominal.
// `gid` = 32 * trajectory + lane-in-trajectory for this lane (the 64 lanes of a wave hold two consecutive trajectories);
// `force` treats every trajectory as active whatever its flag says (fixed-iteration benchmarking runs).
template <bool RK4, int PF>
__device__ __forceinline__ void linesearch_quad_body(const quattro_model_params& p, float* x_nom, float* u_nom,
                                                     const float* __restrict__ K, const float* __restrict__ k,
                                                     const AlphaList& al, int n_alpha, int B, int N, double tol,
                                                     double* cost, int32_t* __restrict__ alpha_idx, int32_t* active,
                                                     int32_t* iters, float* __restrict__ scratch, const int gid,
                                                     const bool force) {
  const int b = gid >> 5, ai = (gid >> 2) & 7, l32 = gid & 31;
  const bool live = (b < B) && (force || active == nullptr || active[b < B ? b : 0] != 0);
  if (!__any(live)) return;   // both trajectories of the wave converged / out of range: nothing to do (late iterations
                              // of a solve run mostly such waves)
  const bool mine = live && ai < n_alpha;
  const size_t bb = live ? b : 0;
  const int aa = mine ? ai : 0;
  const LaneConst L = lane_const(p, gid & 3);
  float* xn = x_nom + bb * (N + 1) * NX;
  float* un = u_nom + bb * N * NU;
  const int wb = __builtin_amdgcn_readfirstlane(b);    // the wave's first trajectory (gid grows with the lane; wb < B: a lane is live)
  const NomSrc nom(L, x_nom, u_nom, K, k, N, wb, live ? b - wb : 0, 2);
  float* sc = scratch + (bb * 8) * (size_t)N * CS;     // this trajectory's 8 candidate slots
  float alpha = al.a[0];
#pragma unroll
  for (int i = 1; i < QUATTRO_MAX_ALPHAS; ++i) alpha = (aa == i) ? al.a[i] : alpha;
  const double J0 = live ? cost[bb] : 0.0;
  double J = quad_rollout_closed<RK4, PF>(p, L, nom, alpha, N, mine, ScratchStore(L, sc + (size_t)aa * N * CS, mine));
  J = quad_sum(J);
  const bool ok = mine && (J <= J0);     // false for NaN, like the reference's comparison
  // first accepted alpha among this trajectory's 8 quads (bit 4*ai of its 32-bit half of the ballot)
  const unsigned long long bal = __ballot(ok);
  const int lane = threadIdx.x & 63;
  const unsigned grp = (unsigned)((bal >> (lane & 32)) & 0x11111111ull);
  const int first = grp ? ((__ffs((int)grp) - 1) >> 2) : -1;
  if (live && first >= 0) {
    // The candidate records were stored by OTHER lanes of this wave.  For a workgroup-scope release gfx950 emits
    // s_waitcnt lgkmcnt(0) only (the CU is assumed to perform its vector-memory operations in order); the explicit vmcnt(0)
    // makes the hand-over independent of that assumption: every scratch store of this wave is complete (written through to
    // L2) before any lane loads a record (one wait per line search).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // Lane pair (e, e + 16) of the trajectory's 32 lanes takes element e of two consecutive records: e < 12 is state e
    // (natural order) read from its axis-major slot 4 (e % 3) + e / 3, e >= 12 is control e - 12.  Stores are contiguous
    // runs of 12 (4) floats per record; the reads stay inside the record's 64 bytes.
    const float* src = sc + (size_t)first * N * CS;
    const int e = l32 & 15, half = l32 >> 4;
    const bool isx = e < 12;
    const int slot = isx ? 4 * (e % 3) + e / 3 : e;
    float* dst = isx ? xn + NX + e : un + (e - 12);
    const int dstride = isx ? NX : NU;
#if QT_ABLATE_LS != 6
    // COPY_U records per lane in flight: all of a block's loads are issued before its first store (a dword load / wait / store
    // per record, as the plain loop compiles, is a round trip to L2 per four records: 10 % of the kernel by the ablation)
    constexpr int COPY_U = 13;
    for (int t0 = half; t0 < N; t0 += 2 * COPY_U) {
      float v[COPY_U];
#pragma unroll
      for (int q = 0; q < COPY_U; ++q) {
        const int t = t0 + 2 * q;
        v[q] = src[(size_t)(t < N ? t : t0) * CS + slot];
      }
#pragma unroll
      for (int q = 0; q < COPY_U; ++q) {
        const int t = t0 + 2 * q;
        if (t < N) dst[(size_t)t * dstride] = v[q];
      }
    }
#else
    dst[0] = src[slot];
#endif
    if (ai == first && L.j == 0) {
      cost[b] = J;
      if (active != nullptr && fabs(J0 - J) < tol) active[b] = 0;   // converged
    }
  }
  if (live && l32 == 0) {
    if (alpha_idx != nullptr) alpha_idx[b] = first;
    if (iters != nullptr) iters[b] += 1;
    if (first < 0 && active != nullptr) active[b] = 0;            // no improving step
  }
}

}  // namespace
