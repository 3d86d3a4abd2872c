// Body of the quadrotor-shaped Riccati-like sweep, shared by sweep_tile16.hip (one launch per sweep) and solve_quad.hip (the
// device-resident solve loop).  Design notes: sweep_tile16.hip.
#pragma once
#include "models_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct StepRegs {
  float f0, f1, f2;  // F[3r+s][z(c)]
  f32x4 lq;          // state column j: (l_xx[3r..3r+2][j], l_ux[r][j]); control column: l_uu[r][0..3]
  float lz;          // l_z[z(c)]
};

struct LanePtrs {
  const float* pf;
  const float* plq;
  const float* plz;
  bool dynf, dynq;   // TILE16C / TILE16R: this lane's F triple / l_zz quad changes from step to step (else it sits in the header)
};

// COMPACT (TILE16C): a lane's loads point into the per-step compact record (stride Tile16CRec::STRIDE) when what it
// holds depends on (x_t, u_t), and into the constant header record (stride 0) otherwise.  Same three loads per step.
// STRIDE: floats between the per-step records; HDR: constants may sit in a header record (stride 0)
template <int STRIDE, bool HDR>
__device__ __forceinline__ StepRegs load_step(const LanePtrs& lp, int s) {
  StepRegs o;
  const int off = s * STRIDE;
  const int offf = (!HDR || lp.dynf) ? off : 0, offq = (!HDR || lp.dynq) ? off : 0;
  o.f0 = lp.pf[offf + 0];
  o.f1 = lp.pf[offf + 1];
  o.f2 = lp.pf[offf + 2];
  o.lq = *reinterpret_cast<const f32x4*>(lp.plq + offq);
  o.lz = lp.plz[off];
  return o;
}

__device__ __forceinline__ float sel4(int r, float a0, float a1, float a2, float a3) {
  const float lo = (r & 1) ? a1 : a0, hi = (r & 1) ? a3 : a2;
  return (r & 2) ? hi : lo;
}

// value held by lane group P (lanes 16P..16P+15), delivered to the same column of every lane group
// (one ds_bpermute through the LDS crossbar; c4 = 4 * (lane & 15).  Measured alternatives, both slower at B = 4096:
//  the v_permlane16/32_swap pair (~7 issue slots with its register copies and hazard nops: 102 vs 95 us), and an
//  indicator-operand v_mfma_f32_16x16x4_f32 as broadcast / column-sum engine (15 instead of 7 MFMAs per step, no LDS
//  hops in the chain: 117 vs 103 us — an MFMA holds its SIMD for 32 cycles, the time of 8 VALU instructions))
template <int P>
__device__ __forceinline__ float bcast_row(float v, int c4) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(c4 + 64 * P, __float_as_int(v)));
}

// sum over the four 16-lane rows: every lane ends with v(c) + v(c+16) + v(c+32) + v(c+48); a16 / a32 are the byte
// addresses 4 * (lane ^ 16), 4 * (lane ^ 32)
__device__ __forceinline__ float sum_rows(float v, int a16, int a32) {
  const float t = v + __int_as_float(__builtin_amdgcn_ds_bpermute(a16, __float_as_int(v)));
  return t + __int_as_float(__builtin_amdgcn_ds_bpermute(a32, __float_as_int(t)));
}

// value held by column C of each lane group, delivered to all 16 lanes of that group (DPP row_newbcast, folded by the
// compiler into the consuming v_mul_f32_dpp: no LDS crossbar hop in the pivot chain)
template <int C>
__device__ __forceinline__ float bcast_col(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + C, 0xf, 0xf, true));
}

// 1 / pivot: the hardware reciprocal as it is (<= 1 ulp).  A Newton step on top (two dependent FMAs per pivot, eight
// per step, in the middle of the pivot chain) changed K by 4e-9 relative on the golden inputs and cost 2.4 % of the
// kernel.
__device__ __forceinline__ float recip(float x) { return __builtin_amdgcn_rcpf(x); }

// one Gauss-Jordan pivot step on the 4 x 16 matrix R (element [r][c] in lane 16r + c) and the side vector q (q[r] in
// every lane of group r)
template <int P, bool CHECK>
__device__ __forceinline__ void gj_step(float& R, float& q, int r, int c4, float& pivmin, float R0, bool& illc) {
  const float piv = qt_readlane(R, 16 * P + 4 * P + 3);
  pivmin = fminf(pivmin, fabsf(piv));
  if constexpr (CHECK) {
    const float d0 = qt_readlane(R0, 16 * P + 4 * P + 3);   // (Q_uu + reg I)[P][P] before any elimination
    illc = illc || !(piv > 1.0e-6f * fabsf(d0));            // also true for piv <= 0 and NaN
  }
  const float ip = recip(piv);
  const float rowp = bcast_row<P>(R, c4);
  const float colp = bcast_col<4 * P + 3>(R);
  const float qp = qt_readlane(q, 16 * P);
  const float f = colp * ip;
  const bool isp = (r == P);
  R = isp ? rowp * ip : fmaf(-f, rowp, R);
  q = isp ? qp * ip : fmaf(-f, qp, q);
}

constexpr int LD = 20;  // LDS row pitch (floats) of the 16x16 transpose tile: 16-B aligned rows, conflict-free b128 writes

// Diagnostic build only (-DQT_SWEEP_PROFILE, scripts/sweep_profile.sh): s_memtime deltas per phase of a step, summed over
// the sweep by every wave; each stamp first forces the phase's result (a readfirstlane on it), so the deltas follow the
// dependency chain.  The shipped library is built without it.
#ifdef QT_SWEEP_PROFILE
#define QT_SWEEP_DBG_PARAM , unsigned long long* __restrict__ dbg
#define QT_PH(i, v)                                                          \
  do {                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                       \
    const int force_ = __builtin_amdgcn_readfirstlane(__float_as_int(v));    \
    asm volatile("" ::"s"(force_));                                          \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
    ph[i] += now_ - last;                                                    \
    last = now_;                                                             \
    __builtin_amdgcn_sched_barrier(0);                                       \
  } while (0)
#else
#define QT_SWEEP_DBG_PARAM
#define QT_PH(i, v)
#endif

// MODE_FUSED (Euler quadrotor): no record buffer at all.  The 76 state-dependent floats of a TILE16C record are functions
// of (x_t, u_t) only, not of the value function, so the wave linearises its own trajectory ahead of the chain: 16 lanes
// produce the records of 16 steps at a time into an LDS stage (the SAME fill_const / fill_state code as
// linearize_compact_kernel: bit-identical records), the constants of the problem sit once in an LDS header record, and
// the recursion reads both exactly as the TILE16C kernel reads its record buffer.  Per step 64 B (x_t, u_t) come from
// HBM instead of 304 B, and the separate linearisation launch (and its 62 MB of record writes) is gone; the terminal
// pair V_x(N) = 2 Qf (x_N - x_ref), V_xx(N) = 2 Qf is formed in registers.
constexpr int MODE_TILE16 = 0, MODE_COMPACT = 1, MODE_FUSED = 2, MODE_DENSEF = 3;   // DENSEF: TILE16R records
constexpr int FUSED_BATCH = 25;   // 2 refills for N = 50; 9.8 KB of LDS per wave still keeps 16 workgroups on a CU

struct FusedArgs {
  quattro_model_params p;
  const float* x;   // [B][N+1][12]
  const float* u;   // [B][N][4]
  int N, t_start;
  int B;            // trajectories (the grid is ceil(B / WPB) workgroups)
};

#ifndef QT_SWEEP_WPB
#define QT_SWEEP_WPB 2
#endif
constexpr int WPB = QT_SWEEP_WPB;   // trajectories (waves) per workgroup; the waves of a workgroup never synchronise
__device__ __forceinline__ void wave_sync() {
  if constexpr (WPB == 1) {
    __syncthreads();   // single-wave workgroup: orders this wave's LDS traffic, no s_barrier is emitted
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// The recursion for ONE trajectory `b`, run by one wavefront (`lane` = lane id); s_t (16 * LD floats), s_vx (16) and s_lin
// (sweep_lin_floats<MODE>()) are this wave's private LDS slices.  Called by sweep_tile16_kernel (one launch per sweep) and by
// the device-resident solve loop (solve_quad.hip), which runs it once per iLQR iteration inside one persistent launch.
template <int MODE>
constexpr int sweep_lin_floats() { return MODE == MODE_FUSED ? Tile16Rec::STRIDE + FUSED_BATCH * Tile16FRec::STRIDE : 4; }

template <int MODE>
__device__ __forceinline__ void sweep_tile16_body(const float* __restrict__ rec, const float* __restrict__ VxN,
                                                  const float* __restrict__ VxxN, int S, float reg,
                                                  float* __restrict__ Kout, float* __restrict__ kout,
                                                  int32_t* __restrict__ status, const FusedArgs& fa, const int b,
                                                  const int lane, float* s_t, float* s_vx, float* s_lin QT_SWEEP_DBG_PARAM) {
  constexpr bool COMPACT = MODE != MODE_TILE16;              // constants of the problem in a header record
  constexpr int REC_STRIDE = MODE == MODE_TILE16 ? Tile16Rec::STRIDE : MODE == MODE_DENSEF ? Tile16RRec::STRIDE : Tile16CRec::STRIDE;
  constexpr bool FUSED = MODE == MODE_FUSED;
  const int r = lane >> 4, c = lane & 15, g = c >> 2, sp = c & 3;
  const bool ucol = (sp == 3);
  const int xj = 3 * g + (ucol ? 0 : sp);  // state index of this lane's tile column (unused for control columns)
  const float regadd = (c == 4 * r + 3) ? reg : 0.0f;   // this lane holds Q_uu[r][r]
  const int c4 = 4 * c, a16 = 4 * (lane ^ 16), a32 = 4 * (lane ^ 32);   // ds_bpermute byte addresses

  // terminal values: A-operand layout of V_xx is lane(r,c) = V[x_j][3r+s]; used as given (not symmetrised)
  float vA0 = 0.0f, vA1 = 0.0f, vA2 = 0.0f;
  float vx0, vx1, vx2;
  if constexpr (FUSED) {
    const float* xN = fa.x + ((size_t)b * (fa.N + 1) + fa.N) * 12;
    if (!ucol) {
      vA0 = (xj == 3 * r + 0) ? 2.0f * fa.p.qf[xj] : 0.0f;
      vA1 = (xj == 3 * r + 1) ? 2.0f * fa.p.qf[xj] : 0.0f;
      vA2 = (xj == 3 * r + 2) ? 2.0f * fa.p.qf[xj] : 0.0f;
    }
    vx0 = 2.0f * fa.p.qf[3 * r + 0] * (xN[3 * r + 0] - fa.p.x_ref[3 * r + 0]);
    vx1 = 2.0f * fa.p.qf[3 * r + 1] * (xN[3 * r + 1] - fa.p.x_ref[3 * r + 1]);
    vx2 = 2.0f * fa.p.qf[3 * r + 2] * (xN[3 * r + 2] - fa.p.x_ref[3 * r + 2]);
    // the constants of the problem, once
    for (int i = lane; i < Tile16Rec::STRIDE; i += QT_WAVE) s_lin[i] = 0.0f;
    wave_sync();
    if (lane == 0) EulerRecord<QUATTRO_MODEL_QUADROTOR, Tile16Rec>::fill_const(s_lin, fa.p);
    wave_sync();
  } else {
    if (!ucol) {
      const float* pv = VxxN + (size_t)b * 144 + xj * 12 + 3 * r;
      vA0 = pv[0];
      vA1 = pv[1];
      vA2 = pv[2];
    }
    vx0 = VxN[(size_t)b * 12 + 3 * r + 0];
    vx1 = VxN[(size_t)b * 12 + 3 * r + 1];
    vx2 = VxN[(size_t)b * 12 + 3 * r + 2];
  }

  LanePtrs lp;
  // MODE_FUSED: float offsets into s_lin of this lane's three loads (+ local step x STRIDE for the dynamic ones)
  int of_f = 0, of_q = 0, of_z = 0;
  f32x4 lqc = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool qsel = ucol && (g == r);
  if constexpr (FUSED) {
    const int d = Tile16CRec::dyn_index(lane);
    lp.dynf = d >= 0;
    lp.dynq = ucol;
    lp.pf = lp.plq = lp.plz = nullptr;
    of_f = lp.dynf ? Tile16Rec::STRIDE + Tile16FRec::F + 3 * d : Tile16Rec::F + 3 * lane;
    of_q = Tile16Rec::STRIDE + Tile16FRec::LUUD + r;
    of_z = Tile16Rec::STRIDE + Tile16FRec::LZ + (ucol ? 12 + g : xj);
    // l_xx = 2Q and l_ux = 0 are constants of the problem: this lane's quad of them stays in registers for the whole
    // sweep; only l_uu[r][r] (one float, control-column lane g == r) changes from step to step
    if (!ucol) lqc = *reinterpret_cast<const f32x4*>(&s_lin[Tile16Rec::LXB + 4 * (12 * r + xj)]);
  } else if constexpr (MODE == MODE_DENSEF) {
    const float* hdr = rec;                                                         // constant TILE16 record
    const float* base = rec + Tile16RRec::HEADER + (size_t)b * S * Tile16RRec::STRIDE;
    lp.dynf = true;
    lp.dynq = ucol;
    lp.pf = base + Tile16RRec::F + 12 * c + 3 * r;             // column-major F: (F[3r][z(c)], F[3r+1][z(c)], F[3r+2][z(c)])
    lp.plq = ucol ? base + Tile16RRec::LUU + 4 * r : hdr + Tile16Rec::LXB + 4 * (12 * r + xj);
    lp.plz = base + Tile16RRec::LZ + (ucol ? 12 + g : xj);
  } else if constexpr (COMPACT) {
    const float* hdr = rec;                                                         // constant TILE16 record
    const float* base = rec + Tile16CRec::HEADER + (size_t)b * S * Tile16CRec::STRIDE;
    const int d = Tile16CRec::dyn_index(lane);
    lp.dynf = d >= 0;
    lp.dynq = ucol;
    lp.pf = lp.dynf ? base + Tile16CRec::F + 3 * d : hdr + Tile16Rec::F + 3 * lane;
    lp.plq = ucol ? base + Tile16CRec::LUU + 4 * r : hdr + Tile16Rec::LXB + 4 * (12 * r + xj);
    lp.plz = base + Tile16CRec::LZ + (ucol ? 12 + g : xj);
  } else {
    const float* base = rec + (size_t)b * S * Tile16Rec::STRIDE;
    lp.dynf = lp.dynq = true;
    lp.pf = base + Tile16Rec::F + 3 * lane;
    lp.plq = base + (ucol ? Tile16Rec::LUU + 4 * r : Tile16Rec::LXB + 4 * (12 * r + xj));
    lp.plz = base + Tile16Rec::LZ + (ucol ? 12 + g : xj);
  }
  float* pK = Kout + ((size_t)b * S) * 48 + r * 12 + xj;
  float* pk = kout + ((size_t)b * S) * 4 + r;

  bool bad = false, illc = false;
  float pivmin = 3.0e38f;   // smallest |pivot| seen: 0 (or NaN-poisoned gains) marks a singular Q_uu + reg I

#ifdef QT_SWEEP_PROFILE
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long last = __builtin_amdgcn_s_memtime();
  const unsigned long long t_begin = last;
#endif
  // one step of the recursion on the record held in `cur`
  auto step = [&](const StepRegs& cur, int s) __attribute__((always_inline)) {
    // P = V_xx F   (tile rows 4r+s <-> x_{3r+s}; the control slot s = 3 contributes nothing)
    f32x4 P = {0.0f, 0.0f, 0.0f, 0.0f};
    P = __builtin_amdgcn_mfma_f32_16x16x4f32(vA0, cur.f0, P, 0, 0, 0);
    P = __builtin_amdgcn_mfma_f32_16x16x4f32(vA1, cur.f1, P, 0, 0, 0);
    P = __builtin_amdgcn_mfma_f32_16x16x4f32(vA2, cur.f2, P, 0, 0, 0);
    // Q = L_zz + F^T P
    f32x4 Q;
    if constexpr (FUSED) {
      Q = lqc;
      Q[3] = qsel ? cur.lq[0] : lqc[3];   // l_uu[r][g]: its diagonal from the step's record, zero elsewhere
    } else {
      Q = cur.lq;
      if (ucol) Q = f32x4{0.0f, 0.0f, 0.0f, sel4(g, cur.lq[0], cur.lq[1], cur.lq[2], cur.lq[3])};
    }
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f0, P[0], Q, 0, 0, 0);
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f1, P[1], Q, 0, 0, 0);
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f2, P[2], Q, 0, 0, 0);
    // q_z = l_z + F^T V_x   (valid in every lane group after the row sum)
    QT_PH(0, Q[3]);
    const float qz = cur.lz + sum_rows(fmaf(cur.f0, vx0, fmaf(cur.f1, vx1, cur.f2 * vx2)), a16, a32);
    QT_PH(1, qz);

    // (Q_uu + reg I)^-1 [Q_ux | Q_u] by Gauss-Jordan on the control rows
    const float q3 = Q[3];
    float R = q3 + regadd;              // reg on the Q_uu diagonal only (regadd = diag ? reg : 0, hoisted)
    float qu = __shfl(qz, 4 * r + 3);   // Q_u[r]
    const float R0 = R;
    gj_step<0, MODE == MODE_TILE16>(R, qu, r, c4, pivmin, R0, illc);
    gj_step<1, MODE == MODE_TILE16>(R, qu, r, c4, pivmin, R0, illc);
    gj_step<2, MODE == MODE_TILE16>(R, qu, r, c4, pivmin, R0, illc);
    gj_step<3, MODE == MODE_TILE16>(R, qu, r, c4, pivmin, R0, illc);
    // In the control columns of the tile the three values below are meaningless (they hold -I, Q_uu - reg I): they
    // are left as they are.  Every product that follows only ever combines state-column lanes with state-row
    // registers into the state-state entries that survive, so nothing is spent on zeroing the rest.
    QT_PH(2, R);
    const float Kv = -R;                                  // K[r][j]
    const float kr = -qu;                                 // k[r]
    const float E = fmaf(-reg, Kv, q3);                   // (Q_ux - reg K)[r][j]
    bad = bad || !qt_finite(Kv) || !qt_finite(kr);

    // outputs: K [m][n] row-major, k [m]
    if (!ucol) pK[s * 48] = Kv;
    if (c == 3) pk[s * 4] = kr;

    // V_xx' = Q_xx + E^T K ; V_x' = Q_x + E^T k
    f32x4 Vn = __builtin_amdgcn_mfma_f32_16x16x4f32(E, Kv, Q, 0, 0, 0);
    const float vxn = qz + sum_rows(E * kr, a16, a32);
    // (control rows / columns of V_xx' hold leftovers of Q_xu, Q_uz: never read as state-state data below)

    QT_PH(3, Vn[0] + vxn);
    // symmetrise through LDS: write the tile transposed, read it back in place.  (Needed: without the 1/2 (V + V^T) the
    // fp32 recursion is unstable — K off by 5 % after 50 steps.  Measured alternative: V'^T from four more MFMAs with the
    // operand roles swapped, no data movement at all — 101 vs 86 us, the MFMA pipe is the contended resource at 4 waves
    // per SIMD.)
    wave_sync();
    *reinterpret_cast<f32x4*>(&s_t[c * LD + 4 * r]) = Vn;
    if (r == 0) s_vx[c] = vxn;
    wave_sync();
    const float t0 = s_t[(4 * r + 0) * LD + c], t1 = s_t[(4 * r + 1) * LD + c], t2 = s_t[(4 * r + 2) * LD + c];
    const f32x4 vxq = *reinterpret_cast<const f32x4*>(&s_vx[4 * r]);
    vA0 = 0.5f * (Vn[0] + t0);
    vA1 = 0.5f * (Vn[1] + t1);
    vA2 = 0.5f * (Vn[2] + t2);
    vx0 = vxq[0];
    vx1 = vxq[1];
    vx2 = vxq[2];
    QT_PH(4, vA0 + vx0);
  };

  // record of local step `ls` (global step base + ls)
  auto load = [&](int ls) __attribute__((always_inline)) {
    if constexpr (FUSED) {
      StepRegs o;
      const int off = ls * Tile16FRec::STRIDE;
      const int a = of_f + (lp.dynf ? off : 0);
      o.f0 = s_lin[a + 0];
      o.f1 = s_lin[a + 1];
      o.f2 = s_lin[a + 2];
      o.lq = f32x4{s_lin[of_q + off], 0.0f, 0.0f, 0.0f};
      o.lz = s_lin[of_z + off];
      return o;
    } else {
      return load_step<REC_STRIDE, COMPACT>(lp, ls);
    }
  };
  // Steps base + cnt - 1 ... base.  Three record buffers rotate through an unrolled-by-3 loop, so a record is requested
  // three steps before it is consumed and no register copies (which would force the loads to land early) are needed.
  auto run = [&](int cnt, int base) __attribute__((always_inline)) {
    StepRegs b0, b1, b2;
    b0 = load(cnt - 1);
    b1 = load(cnt > 1 ? cnt - 2 : 0);
    b2 = load(cnt > 2 ? cnt - 3 : 0);
    int s = cnt - 1;
    for (; s >= 2; s -= 3) {
      step(b0, base + s);
      b0 = load(s >= 3 ? s - 3 : 0);
      step(b1, base + s - 1);
      b1 = load(s >= 4 ? s - 4 : 0);
      step(b2, base + s - 2);
      b2 = load(s >= 5 ? s - 5 : 0);
    }
    if (s >= 0) step(b0, base + s);
    if (s >= 1) step(b1, base + s - 1);
  };
  if constexpr (FUSED) {
    float* stage = s_lin + Tile16Rec::STRIDE;
    // (x_t, u_t) of a batch's steps, one step per lane: requested a whole batch ahead (the loads of batch j - 1 fly while
    // the 16 steps of batch j run; waiting for them at the refill would expose an HBM round trip four times per sweep)
    float4 xa, xb, xc, ua;
    auto fetch = [&](int base) __attribute__((always_inline)) {
      const int cnt = S - base < FUSED_BATCH ? S - base : FUSED_BATCH;
      const int t = fa.t_start + base + (lane < cnt ? lane : 0);
      const float4* px = reinterpret_cast<const float4*>(fa.x + ((size_t)b * (fa.N + 1) + t) * 12);
      xa = px[0];
      xb = px[1];
      xc = px[2];
      ua = *reinterpret_cast<const float4*>(fa.u + ((size_t)b * fa.N + t) * 4);
    };
    const int top = ((S - 1) / FUSED_BATCH) * FUSED_BATCH;
    fetch(top);
    for (int base = top; base >= 0; base -= FUSED_BATCH) {
      const int cnt = S - base < FUSED_BATCH ? S - base : FUSED_BATCH;
      wave_sync();                                       // the previous batch's records are no longer read
      if (lane < cnt) {
        const float xs[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
        const float us[4] = {ua.x, ua.y, ua.z, ua.w};
        float* mine = stage + lane * Tile16FRec::STRIDE;
        // exactly linearize_compact_kernel's sequence (a dynamic lane's triple may hold constants and structural zeros)
#pragma unroll
        for (int i = 0; i < Tile16FRec::SIZE / 4; ++i) reinterpret_cast<float4*>(mine)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        EulerRecord<QUATTRO_MODEL_QUADROTOR, Tile16FRec>::fill_const(mine, fa.p);
        EulerRecord<QUATTRO_MODEL_QUADROTOR, Tile16FRec>::fill_state(mine, fa.p, xs, us);
      }
      if (base > 0) fetch(base - FUSED_BATCH);
      wave_sync();
      run(cnt, base);
    }
  } else {
    run(S, 0);
  }
#ifdef QT_SWEEP_PROFILE
  if (lane == 0) {
    for (int i = 0; i < 5; ++i) dbg[(size_t)b * 8 + i] = ph[i];
    dbg[(size_t)b * 8 + 5] = __builtin_amdgcn_s_memtime() - t_begin;
  }
#endif
  const bool singular = !(pivmin > 0.0f);
  if (status != nullptr) {
    const bool any_bad = __any(bad);
    if (lane == 0)
      status[b] = (any_bad ? QUATTRO_TRAJ_NONFINITE : 0) | (singular ? QUATTRO_TRAJ_SINGULAR : 0) |
                  (illc ? QUATTRO_TRAJ_ILLCOND : 0);
  }
}

}  // namespace
