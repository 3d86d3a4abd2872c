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

// One Gauss-Jordan pivot step on the 4 x 16 matrix R (element [r][c] in lane 16r + c).  The pivot and the multiplier column are read
// from `S`: R itself, except for the first pivot, whose column (tile column 3) has been handed to Q_u by then (see step()) and
// survives in a copy.  The right-hand side Q_u needs no instructions of its own: it rides in tile column 3 of R.
template <int P, bool CHECK>
__device__ __forceinline__ void gj_step(float& R, const float S, int r, int c4, float& pivmin, float R0, bool& illc) {
  const float piv = qt_readlane(S, 16 * P + 4 * P + 3);
  pivmin = fminf(pivmin, fabsf(piv));
  if constexpr (CHECK) {
    const float d0 = qt_readlane(R0, 16 * P + 4 * P + 3);   // (Q_uu + reg I)[P][P] before any elimination
    illc = illc || !(piv > 1.0e-6f * fabsf(d0));            // also true for piv <= 0 and NaN
  }
  const float ip = recip(piv);
  const float rowp = bcast_row<P>(R, c4);
  const float colp = bcast_col<4 * P + 3>(S);
  const float f = colp * ip;
  R = (r == P) ? rowp * ip : fmaf(-f, rowp, R);
}

// Diagnostic build only (-DQT_ABLATE=n, scripts/ablate_sweep.sh): one segment of the step is left out (the numbers that come out are
// wrong on purpose) so that the segment's REAL share of the time shows — at full load and for a lone wave — without stamps that
// serialise what the hardware overlaps.  1: the four pivots; 2: the LDS transposition; 3: the three Q MFMAs; 4: the V' MFMA and the
// V_x' row sum; 5: the stores of K, k.  The shipped library is built without it.
// Wave priority along a step of the recursion (round 4).  A SIMD arbitrates vector issue between its resident waves by priority, then
// age (MI355X_MICROARCH.md, "Two waves per SIMD"); the four waves it holds here sit at different points of their steps, and the
// one inside the elimination — a chain of dependent cross-lane hops, reciprocals and row operations where every issue delay is
// latency — should not queue behind another wave's run of MFMAs.  QT_SWEEP_PRIO = 1 (default): the four pivots at priority 2, the
// rest of the step's chain (E, stores, V' product, row sum, LDS transposition, symmetrisation) at 1, the P / Q MFMA block at 0:
// 64.9 -> 59.8 us at B = 4096, persistent loop 97.4 -> 90.8 us per iteration (A/B on one box; a lone wave pays ~1 us per
// iteration for the three s_setprio per step).  Other placements measured: pivots only 60.7, pivots .. row sum 60.6, pivots .. end
// of step 60.5, the MFMA block raised instead 64.0, three levels (pivots 3, transposition 2, rest 1) 60.1 us.  0 = none.
#ifndef QT_SWEEP_PRIO
#define QT_SWEEP_PRIO 1
#endif
#ifndef QT_ABLATE
#define QT_ABLATE 0
#endif

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
// MODE_FUSED_RK4 (RK4 quadrotor): the sweep's own wave also linearises its trajectory, off the value-function chain.
// (1) Once per sweep, one LANE per step evaluates the four RK4 stage points of its step and leaves the 28 non-trivial entries
// of each stage's rate Jacobian, premultiplied (rk4_stage_entries), plus l_z and diag(l_uu), in a global scratch record of 528 B
// per step (written and read back by the same wave: L2 / Infinity Cache traffic, never a second kernel's; 12 steps' worth at a
// time is copied into LDS).  (2) The discrete Jacobian [A | B] = I + dt/6 (D1 + 2 D2 + 2 D3 + D4), D_{s+1} = M_{s+1} (I + c_s D_s),
// is a chain of three 16 x 16 x 16 products per step — forward mode for all 16 unit directions of z = (x, u) at once — and those
// run on the MATRIX pipe (exact-fp32 MFMA, 12 per step): a lane fetches its four entries of M from the step's coefficient
// table by index (plus a per-lane constant for the structural entries), the accumulator of one product is the B operand of
// the next, and the result comes out in exactly the registers the recursion reads [A | B] from (C-layout rows 4r + s = x_{3r+s}).
// The products of step t - 3 are issued while the recursion works on step t: they are independent of the value function and
// fill the matrix pipe in the shadow of the chain's dependency stalls.  Per lane the whole linearisation state is 3 tiles of
// 4 registers.  (A first version pushed directions through the stages on the vector ALUs, 4 steps x 16 directions per pass
// through an LDS stage: with three 12-vectors per lane next to the recursion's own state it spilled ~100 registers, and the
// scratch round trips made it 4x slower than linearising in a kernel of its own.)
// The stand-alone RK4 path (linearize_rk4_quad_kernel -> 187 MB of TILE16R records -> sweep) costs 63 + 75 us at B = 4096.
constexpr int MODE_FUSED_RK4 = 4;
// MODE_ROWPAD: ROWMAJOR records of ANY problem with n <= 12, m <= 4 (a user-compiled model's, or foreign ones) on the same 16 x 16
// tile recursion: the kernel pads inside — a lane whose tile row / column has no state or control of the problem behind it
// holds constants (zero; one on the diagonal of l_uu, so that the padded controls are unit pivots of the elimination and get
// zero gains) instead of loading — so the records stay as small as the problem (432 B per step for n = 6, m = 2 instead of a
// padded TILE16 record's 1,664 B) and nothing upstream changes.  Eight scalar loads + selects per step instead of five
// loads; no pivoting, like every tile mode (QUATTRO_TRAJ_ILLCOND reports a pivot that needed it).
constexpr int MODE_ROWPAD = 5;
constexpr int RK4_BATCH = 12;                      // steps whose coefficient tables sit in LDS at a time
struct Rk4Tab {                                    // LDS image of a batch: [step][stage][32] coefficient tables, then [step][20] cost entries
  static constexpr int STAGE = 32, STEP = 4 * STAGE, ZERO = 28, COST = RK4_BATCH * STEP, COST_STEP = 20, LUUD = 16,
                       FLOATS = COST + RK4_BATCH * COST_STEP;
};
struct Rk4Coef {                                   // the global scratch record of one step: 33 float4 pieces
  static constexpr int PER_STAGE = 28, LZ = 112, LUUD = 128, SIZE = 132, STRIDE = 132, PIECES = 33;
  // Piece q of step ls of a trajectory with S steps sits at float4 index q * S + ls of the trajectory's block: the producing
  // lanes (one step each) store piece q as ONE contiguous run of 16-byte elements.  (Record-major — lane ls writing its own
  // 528 bytes — is 64 separate 16-byte requests per store instruction to a write-through L2: 2.6x the whole sweep's time.)
};
constexpr int FUSED_BATCH = 25;   // 2 refills for N = 50; 9.8 KB of LDS per wave still keeps 16 workgroups on a CU

struct FusedArgs {
  quattro_model_params p;
  const float* x;   // [B][N+1][12]
  const float* u;   // [B][N][4]
  int N, t_start;
  int B;            // trajectories (the grid is ceil(B / WPB) workgroups)
  float* coef;      // MODE_FUSED_RK4: global scratch, [B][N - t_start][Rk4Coef::STRIDE] floats
  int k_rows;       // rows per trajectory of the gain arrays the fused modes write: 0 = N - t_start (index t - t_start);
                    // N = the full stacks, step t written in place at row t (quattro_linearize_sweep_rows_f32)
  int rn, rm;       // MODE_ROWPAD: the problem's own dimensions (records, terminal pair and gains are laid out for them)
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

// The recursion for ONE trajectory `b`, run by one wavefront (`lane` = lane id); s_t (16 * LD floats), s_vx (64) and s_lin
// (sweep_lin_floats<MODE>()) are this wave's private LDS slices.  Called by sweep_tile16_kernel (one launch per sweep) and by
// the device-resident solve loop (solve_quad.hip), which runs it once per iLQR iteration inside one persistent launch.
template <int MODE>
constexpr int sweep_lin_floats() {
  return MODE == MODE_FUSED ? Tile16Rec::STRIDE + FUSED_BATCH * Tile16FRec::STRIDE
                            : (MODE == MODE_FUSED_RK4 ? Rk4Tab::FLOATS : 4);
}

// ---- MODE_FUSED_RK4, stage (1): the four stage points of one step -> Rk4Coef record (132 floats)
__device__ __forceinline__ void rk4_stage_entries(const QuadStage& s, const quattro_model_params& p, float* c) {
  const QuadTrig& t = s.t;
  const float inv_mass = 1.0f / p.phys[0];
  const float c1 = (p.phys[2] - p.phys[3]) / p.phys[1], c2 = (p.phys[3] - p.phys[1]) / p.phys[2], c3 = (p.phys[1] - p.phys[2]) / p.phys[3];
  c[0] = s.tm * (t.sps * t.cph - t.cps * t.sth * t.sph);     // d v_x' / d phi, theta, psi, thrust
  c[1] = s.tm * (t.cps * t.cth * t.cph);
  c[2] = s.tm * s.ry;
  c[3] = s.rx * inv_mass;
  c[4] = s.tm * (t.cps * t.cph + t.sps * t.sth * t.sph);     // v_y'
  c[5] = -(s.tm * (t.sps * t.cth * t.cph));
  c[6] = -(s.tm * s.rx);
  c[7] = s.ry * inv_mass;
  c[8] = -(s.tm * (t.cth * t.sph));                          // v_z'
  c[9] = -(s.tm * (t.sth * t.cph));
  c[10] = s.rz * inv_mass;
  c[11] = s.dmix * t.tth;                                    // phi'  : d / d phi, theta, q, r  (d / d p = 1)
  c[12] = s.mix * s.sec2;
  c[13] = t.sph * t.tth;
  c[14] = t.cph * t.tth;
  c[15] = -s.mix;                                            // theta': d / d phi, q, r
  c[16] = t.cph;
  c[17] = -t.sph;
  c[18] = s.dmix * t.sec;                                    // psi'  : d / d phi, theta, q, r
  c[19] = s.mix * t.sth * s.sec2;
  c[20] = t.sph * t.sec;
  c[21] = t.cph * t.sec;
  c[22] = c1 * s.wr;                                         // p', q', r' : the gyroscopic terms
  c[23] = c1 * s.wq;
  c[24] = c2 * s.wr;
  c[25] = c2 * s.wp;
  c[26] = c3 * s.wq;
  c[27] = c3 * s.wp;
}

template <class Emit>
__device__ __forceinline__ void rk4_step_coefs(const quattro_model_params& p, const float* xs, const float* us, Emit emit) {
  constexpr int NX = 12;
  const float dt = p.dt;
  float k[NX], xst[NX], c[Rk4Coef::PER_STAGE];
  auto flush = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < Rk4Coef::PER_STAGE / 4; ++q) emit(stage * (Rk4Coef::PER_STAGE / 4) + q, make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]));
  };
  const QuadStage s1 = quad_stage(p, xs, us);
  rk4_stage_entries(s1, p, c);
  flush(0);
  quad_rate_at(s1, p, xs, us, k);
#pragma unroll
  for (int i = 0; i < NX; ++i) xst[i] = fmaf(0.5f * dt, k[i], xs[i]);
  const QuadStage s2 = quad_stage(p, xst, us);
  rk4_stage_entries(s2, p, c);
  flush(1);
  quad_rate_at(s2, p, xst, us, k);
#pragma unroll
  for (int i = 0; i < NX; ++i) xst[i] = fmaf(0.5f * dt, k[i], xs[i]);
  const QuadStage s3 = quad_stage(p, xst, us);
  rk4_stage_entries(s3, p, c);
  flush(2);
  quad_rate_at(s3, p, xst, us, k);
#pragma unroll
  for (int i = 0; i < NX; ++i) xst[i] = fmaf(dt, k[i], xs[i]);
  const QuadStage s4 = quad_stage(p, xst, us);
  rk4_stage_entries(s4, p, c);
  flush(3);
  // cost entries (fill_cost_entries' expressions): l_z = (l_x, l_u), diag(l_uu)
  float lz[16], luud[4];
#pragma unroll
  for (int i = 0; i < NX; ++i) lz[i] = 2.0f * p.q[i] * (xs[i] - p.x_ref[i]);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float lu = 2.0f * p.r[a] * us[a], luu = 2.0f * p.r[a];
    if (p.barrier_alpha != 0.0f) {
      const float sp = qt_softplus(-us[a], p.barrier_beta), sg = qt_sigmoid(-p.barrier_beta * us[a]);
      lu = fmaf(p.barrier_alpha, -2.0f * sp * sg, lu);
      luu = fmaf(p.barrier_alpha, 2.0f * sg * sg + 2.0f * sp * p.barrier_beta * sg * (1.0f - sg), luu);
    }
    lz[NX + a] = lu;
    luud[a] = luu;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) emit(Rk4Coef::LZ / 4 + q, make_float4(lz[4 * q], lz[4 * q + 1], lz[4 * q + 2], lz[4 * q + 3]));
  emit(Rk4Coef::LUUD / 4, make_float4(luud[0], luud[1], luud[2], luud[3]));
}

// ---- MODE_FUSED_RK4, stage (2): where entry [row tile][column tile] of a stage's rate Jacobian M = [f_x | f_u] comes from: index
// into the stage's coefficient table (Rk4Tab::ZERO = none) and a constant to add (structural ones, the torque rows).  Tile
// index 4 g + s' is x_{3g+s'} for s' < 3 and u_g for s' = 3; rows of M for the controls are zero (du' = 0).
struct Rk4Entry {
  int idx;
  float cst;
};
__device__ __forceinline__ Rk4Entry rk4_m_entry(const quattro_model_params& p, int row_tile, int col_tile) {
  Rk4Entry e{Rk4Tab::ZERO, 0.0f};
  if ((row_tile & 3) == 3) return e;
  const int i = 3 * (row_tile >> 2) + (row_tile & 3);
  const bool uc = (col_tile & 3) == 3;
  const int g = col_tile >> 2, j = 3 * g + (col_tile & 3);
  if (uc) {                                                  // d / d u_g: thrust rows and torque rows
    if (i == 3) e.idx = 3;
    else if (i == 4) e.idx = 7;
    else if (i == 5) e.idx = 10;
    else if (i == 9) e.cst = ((g == 1 || g == 2) ? 1.0f : -1.0f) * (p.phys[4] / p.phys[1]);
    else if (i == 10) e.cst = (g < 2 ? 1.0f : -1.0f) * (p.phys[4] / p.phys[2]);
    else if (i == 11) e.cst = ((g & 1) ? -1.0f : 1.0f) * (p.phys[6] / p.phys[3]);
    return e;
  }
  if (i < 3) { if (j == i + 3) e.cst = 1.0f; return e; }
  if (i == 3) { if (j == 6) e.idx = 0; else if (j == 7) e.idx = 1; else if (j == 8) e.idx = 2; return e; }
  if (i == 4) { if (j == 6) e.idx = 4; else if (j == 7) e.idx = 5; else if (j == 8) e.idx = 6; return e; }
  if (i == 5) { if (j == 6) e.idx = 8; else if (j == 7) e.idx = 9; return e; }
  if (i == 6) { if (j == 6) e.idx = 11; else if (j == 7) e.idx = 12; else if (j == 9) e.cst = 1.0f; else if (j == 10) e.idx = 13; else if (j == 11) e.idx = 14; return e; }
  if (i == 7) { if (j == 6) e.idx = 15; else if (j == 10) e.idx = 16; else if (j == 11) e.idx = 17; return e; }
  if (i == 8) { if (j == 6) e.idx = 18; else if (j == 7) e.idx = 19; else if (j == 10) e.idx = 20; else if (j == 11) e.idx = 21; return e; }
  if (i == 9) { if (j == 10) e.idx = 22; else if (j == 11) e.idx = 23; return e; }
  if (i == 10) { if (j == 9) e.idx = 24; else if (j == 11) e.idx = 25; return e; }
  if (j == 9) e.idx = 26; else if (j == 10) e.idx = 27;     // i == 11
  return e;
}

template <int MODE>
__device__ __forceinline__ void sweep_tile16_body(const float* __restrict__ rec, const float* __restrict__ VxN,
                                                  const float* __restrict__ VxxN, int S, float reg,
                                                  float* __restrict__ Kout, float* __restrict__ kout,
                                                  int32_t* __restrict__ status, const FusedArgs& fa, const int b,
                                                  const int lane, float* s_t, float* s_vx, float* s_lin QT_SWEEP_DBG_PARAM) {
  constexpr bool ROWPAD = MODE == MODE_ROWPAD;
  constexpr bool COMPACT = MODE != MODE_TILE16 && !ROWPAD;   // constants of the problem in a header record
  constexpr int REC_STRIDE = MODE == MODE_TILE16 ? Tile16Rec::STRIDE : MODE == MODE_DENSEF ? Tile16RRec::STRIDE : Tile16CRec::STRIDE;
  constexpr bool FUSED = MODE == MODE_FUSED;
  constexpr bool RK4F = MODE == MODE_FUSED_RK4;
  constexpr bool CHECK_PIVOTS = MODE == MODE_TILE16 || ROWPAD;   // records from anywhere: report pivots that needed pivoting
  const int pn = ROWPAD ? fa.rn : 12, pm = ROWPAD ? fa.rm : 4;   // the problem's dimensions (layout of VxN, VxxN, K, k)
  const int r = lane >> 4, c = lane & 15, g = c >> 2, sp = c & 3;
  const bool ucol = (sp == 3);
  const int xj = 3 * g + (ucol ? 0 : sp);  // state index of this lane's tile column (unused for control columns)
  const float regadd = (c == 4 * r + 3) ? reg : 0.0f;   // this lane holds Q_uu[r][r]
  const int c4 = 4 * c, a16 = 4 * (lane ^ 16), a32 = 4 * (lane ^ 32);   // ds_bpermute byte addresses

  // terminal values: A-operand layout of V_xx is lane(r,c) = V[x_j][3r+s]; used as given (not symmetrised)
  float vA0 = 0.0f, vA1 = 0.0f, vA2 = 0.0f;
  float vx0, vx1, vx2;
  if constexpr (FUSED || RK4F) {
    const float* xN = fa.x + ((size_t)b * (fa.N + 1) + fa.N) * 12;
    if (!ucol) {
      vA0 = (xj == 3 * r + 0) ? 2.0f * fa.p.qf[xj] : 0.0f;
      vA1 = (xj == 3 * r + 1) ? 2.0f * fa.p.qf[xj] : 0.0f;
      vA2 = (xj == 3 * r + 2) ? 2.0f * fa.p.qf[xj] : 0.0f;
    }
    vx0 = 2.0f * fa.p.qf[3 * r + 0] * (xN[3 * r + 0] - fa.p.x_ref[3 * r + 0]);
    vx1 = 2.0f * fa.p.qf[3 * r + 1] * (xN[3 * r + 1] - fa.p.x_ref[3 * r + 1]);
    vx2 = 2.0f * fa.p.qf[3 * r + 2] * (xN[3 * r + 2] - fa.p.x_ref[3 * r + 2]);
    if constexpr (FUSED) {
      // the constants of the problem, once
      for (int i = lane; i < Tile16Rec::STRIDE; i += QT_WAVE) s_lin[i] = 0.0f;
      wave_sync();
      if (lane == 0) EulerRecord<QUATTRO_MODEL_QUADROTOR, Tile16Rec>::fill_const(s_lin, fa.p);
      wave_sync();
    }
  } else if constexpr (ROWPAD) {
    // V_xx(N) [n][n], V_x(N) [n] of the problem; rows / columns beyond n are zero (nothing depends on a padded state)
    const float* pv = VxxN + (size_t)b * pn * pn;
    const float* pg = VxN + (size_t)b * pn;
    const bool cj = !ucol && xj < pn;
    const int i0 = 3 * r, xjc = cj ? xj : 0;
    vA0 = (cj && i0 + 0 < pn) ? pv[xjc * pn + (i0 + 0 < pn ? i0 + 0 : 0)] : 0.0f;
    vA1 = (cj && i0 + 1 < pn) ? pv[xjc * pn + (i0 + 1 < pn ? i0 + 1 : 0)] : 0.0f;
    vA2 = (cj && i0 + 2 < pn) ? pv[xjc * pn + (i0 + 2 < pn ? i0 + 2 : 0)] : 0.0f;
    vx0 = i0 + 0 < pn ? pg[i0 + 0] : 0.0f;
    vx1 = i0 + 1 < pn ? pg[i0 + 1 < pn ? i0 + 1 : 0] : 0.0f;
    vx2 = i0 + 2 < pn ? pg[i0 + 2 < pn ? i0 + 2 : 0] : 0.0f;
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

  if (ucol) {            // control-slot columns of the A operand carry V_x (q_z comes out of the P product: see step())
    vA0 = vx0;
    vA1 = vx1;
    vA2 = vx2;
  }

  LanePtrs lp;
  int pad_of[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pad_stride = 0;     // MODE_ROWPAD (see below)
  const float* pad_base = nullptr;
  // MODE_FUSED: float offsets into s_lin of this lane's three loads (+ local step x STRIDE for the dynamic ones)
  int of_f = 0, of_q = 0, of_z = 0;
  f32x4 lqc = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool qsel = ucol && (g == r);
  if constexpr (RK4F) {
    lp.dynf = true;
    lp.dynq = ucol;
    lp.pf = lp.plq = lp.plz = nullptr;
    of_q = Rk4Tab::COST + Rk4Tab::LUUD + r;
    of_z = Rk4Tab::COST + (ucol ? 12 + g : xj);
    // l_xx = 2Q (diagonal), l_ux = 0: this lane's quad (l_xx[3r..3r+2][x_j], l_ux[r][x_j]) straight from the parameters
    if (!ucol) {
      const float q2 = 2.0f * fa.p.q[xj];
      lqc = f32x4{xj == 3 * r + 0 ? q2 : 0.0f, xj == 3 * r + 1 ? q2 : 0.0f, xj == 3 * r + 2 ? q2 : 0.0f, 0.0f};
    }
  } else if constexpr (FUSED) {
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
  } else if constexpr (ROWPAD) {
    // ROWMAJOR record of the problem's own (n, m): [A (n n) | B (n m) | l_xx (n n) | l_ux (m n) | l_uu (m m) | l_x | l_u],
    // stride padded to 4 floats (RowMajorRec).  This lane's eight entries as float offsets into a record, -1 = padded
    const int oB = pn * pn, oLXX = oB + pn * pm, oLUX = oLXX + pn * pn, oLUU = oLUX + pm * pn, oLX = oLUU + pm * pm, oLU = oLX + pn;
    pad_stride = (oLU + pm + 3) / 4 * 4;
    const bool colreal = ucol ? g < pm : xj < pn;                 // this lane's tile column stands for a real direction
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = 3 * r + q;
      pad_of[q] = (colreal && i < pn) ? (ucol ? oB + i * pm + g : i * pn + xj) : -1;                  // F[i][z(c)]
      pad_of[3 + q] = ucol ? ((r < pm && q < pm) ? oLUU + r * pm + q : -1)                            // l_uu[r][q]
                           : ((colreal && i < pn) ? oLXX + i * pn + xj : -1);                         // l_xx[i][x_j]
    }
    pad_of[6] = ucol ? ((r < pm && 3 < pm) ? oLUU + r * pm + 3 : -1) : ((colreal && r < pm) ? oLUX + r * pn + xj : -1);
    pad_of[7] = colreal ? (ucol ? oLU + g : oLX + xj) : -1;                                           // l_z[z(c)]
    pad_base = rec + (size_t)b * S * pad_stride;
    lp.dynf = lp.dynq = true;
    lp.pf = lp.plq = lp.plz = nullptr;
  } else if constexpr (MODE == MODE_DENSEF) {
    const float* hdr = rec;                                                         // constant TILE16 record
    const float* base = rec + Tile16RRec::HEADER + (size_t)b * S * Tile16RRec::STRIDE;
    const int d = Tile16RRec::col_index(c);                    // -1: a position / velocity direction, constant -> header record
    lp.dynf = d >= 0;
    lp.dynq = ucol;
    lp.pf = lp.dynf ? base + Tile16RRec::F + 12 * d + 3 * r    // column-major F: (F[3r][z(c)], F[3r+1][z(c)], F[3r+2][z(c)])
                    : hdr + Tile16Rec::F + 3 * lane;
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
  int krows = S;                                            // rows per trajectory of Kout / kout; local step s sits at row krows - S + s
  if constexpr (FUSED || RK4F) krows = fa.k_rows > 0 ? fa.k_rows : S;
  const int kK = pm * pn;                                   // floats of one K_t / k_t (48 / 4 unless MODE_ROWPAD)
  const bool storeK = !ucol && (!ROWPAD || (r < pm && xj < pn)), storek = c == 3 && (!ROWPAD || r < pm);
  // Gain rows leave through buffer stores (round 4): the trajectory's block is a wave-uniform buffer resource, the step's row a
  // SCALAR offset and the lane's element a loop-invariant 32-bit offset — no per-step 64-bit address arithmetic in vector
  // registers — and lanes that hold no gain entry carry an out-of-range offset, which the hardware's range check drops: no
  // exec-mask branch around the store either.
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(Kout + ((size_t)b * krows + (krows - S)) * kK, 0, S * kK * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(kout + ((size_t)b * krows + (krows - S)) * pm, 0, S * pm * 4, 0x00020000);
  const int voK = storeK ? 4 * (r * pn + xj) : -1, vok = storek ? 4 * r : -1;

  // MODE_FUSED_RK4: this lane's four entries of a stage Jacobian in A layout (lane (r, c): M[tile row c][tile column 4r + q]) and
  // in C layout (M[tile row 4r + q][tile column c]) as table offsets + constants; the identity tile in C layout
  int rk_aidx[4] = {0, 0, 0, 0}, rk_cidx[4] = {0, 0, 0, 0};
  float rk_acst[4] = {0.0f, 0.0f, 0.0f, 0.0f}, rk_ccst[4] = {0.0f, 0.0f, 0.0f, 0.0f}, rk_eye[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float rk_dt = 0.0f;
  if constexpr (RK4F) {
    rk_dt = fa.p.dt;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const Rk4Entry ea = rk4_m_entry(fa.p, c, 4 * r + q), ec = rk4_m_entry(fa.p, 4 * r + q, c);
      rk_aidx[q] = ea.idx;
      rk_acst[q] = ea.cst;
      rk_cidx[q] = ec.idx;
      rk_ccst[q] = ec.cst;
      rk_eye[q] = (4 * r + q == c) ? 1.0f : 0.0f;
    }
  }
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
    if constexpr (FUSED || RK4F) {
      Q = lqc;
      Q[3] = qsel ? cur.lq[0] : lqc[3];   // l_uu[r][g]: its diagonal from the step's record, zero elsewhere
    } else {
      Q = cur.lq;
      if (ucol) Q = f32x4{0.0f, 0.0f, 0.0f, sel4(g, cur.lq[0], cur.lq[1], cur.lq[2], cur.lq[3])};
    }
#if QT_ABLATE != 3
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f0, P[0], Q, 0, 0, 0);
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f1, P[1], Q, 0, 0, 0);
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f2, P[2], Q, 0, 0, 0);
#else
    Q[0] += P[0]; Q[1] += P[1]; Q[2] += P[2]; Q[3] += P[3];
#endif
    // q_z = l_z + F^T V_x comes out of the P product for free (round 4): the four tile rows 4r + 3 are the control slots of the
    // x' index, whose rows of V_xx are not data — the A operand of the lanes that own them (tile column c % 4 == 3) carries V_x
    // instead (vA_s = V_x[x_{3r+s}], see the symmetrisation below), so P[4r' + 3][c] = sum_i V_x[i] F[i][z(c)] lands in register 3
    // of EVERY lane group.  Those rows of P never enter Q (its k-steps use P[0..2] only).  This removes the separate
    // F^T V_x (3 multiply-adds and a 4-row sum through the LDS crossbar: 185 of a lone wave's 1 730 cycles per step).
    QT_PH(0, Q[3]);
    const float qz = cur.lz + P[3];
    QT_PH(1, qz);

    // (Q_uu + reg I)^-1 [Q_ux | Q_u] by Gauss-Jordan on the control rows
    const float q3 = Q[3];
    float R = q3 + regadd;              // reg on the Q_uu diagonal only (regadd = diag ? reg : 0, hoisted)
    const float qu = __shfl(qz, 4 * r + 3);   // Q_u[r]
    const float R0 = R;
    // Round 4: the right-hand side Q_u rides in tile column 3 of R.  That column is the first pivot's own (Q_uu[:][0]): pivot 0
    // reads its pivot and multipliers from the copy R0, every row operation then updates Q_u along with the other fifteen columns,
    // and after the last pivot lane (r, 3) holds (Q_uu + reg I)^-1 Q_u.  Same operations on Q_u as the separate register got
    // (bit-identical k), sixteen vector instructions fewer per step — the elimination is 23 % of the kernel.
    R = (c == 3) ? qu : R;
#if QT_ABLATE != 1
#if QT_SWEEP_PRIO
    __builtin_amdgcn_s_setprio(2);          // the pivots
#endif
    gj_step<0, CHECK_PIVOTS>(R, R0, r, c4, pivmin, R0, illc);
    gj_step<1, CHECK_PIVOTS>(R, R, r, c4, pivmin, R0, illc);
    gj_step<2, CHECK_PIVOTS>(R, R, r, c4, pivmin, R0, illc);
    gj_step<3, CHECK_PIVOTS>(R, R, r, c4, pivmin, R0, illc);
#else
    R *= 0.01f;
#endif
    // In the control columns of the tile the three values below are meaningless (they hold -I, Q_uu - reg I): they
    // are left as they are.  Every product that follows only ever combines state-column lanes with state-row
    // registers into the state-state entries that survive, so nothing is spent on zeroing the rest.
#if QT_SWEEP_PRIO
    __builtin_amdgcn_s_setprio(1);          // the rest of the step's chain
#endif
    QT_PH(2, R);
    const float Kv = -R;                                  // K[r][j]; in tile column 3: k[r]
    const float kr = bcast_col<3>(Kv);                    // k[r] in every lane of group r
    const float E = fmaf(-reg, Kv, q3);                   // (Q_ux - reg K)[r][j]
    bad = bad || !qt_finite(Kv);                          // (covers k: it is column 3 of the same register)

    // outputs: K [m][n] row-major, k [m]
#if QT_ABLATE != 5
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(Kv), rsK, voK, 4 * s * kK, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(Kv), rsk, vok, 4 * s * pm, 0);      // (tile column 3 of Kv is k)
#else
    if (s == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(Kv), rsK, voK, 0, 0);
#endif

    // V_xx' = Q_xx + E^T K ; V_x' = Q_x + E^T k
#if QT_ABLATE != 4
    f32x4 Vn = __builtin_amdgcn_mfma_f32_16x16x4f32(E, Kv, Q, 0, 0, 0);
    const float vxn = qz + sum_rows(E * kr, a16, a32);
#else
    f32x4 Vn = Q;
    Vn[0] += E;
    const float vxn = qz + E * kr;
#endif
    // (control rows / columns of V_xx' hold leftovers of Q_xu, Q_uz: never read as state-state data below)

    QT_PH(3, Vn[0] + vxn);
    // symmetrise through LDS: write the tile transposed, read it back in place.  (Needed: without the 1/2 (V + V^T) the
    // fp32 recursion is unstable — K off by 5 % after 50 steps.  Measured alternative: V'^T from four more MFMAs with the
    // operand roles swapped, no data movement at all — 101 vs 86 us, the MFMA pipe is the contended resource at 4 waves
    // per SIMD.)
#if QT_ABLATE != 2
    wave_sync();
    *reinterpret_cast<f32x4*>(&s_t[c * LD + 4 * r]) = Vn;
    s_vx[lane] = vxn;        // (every lane into a slot of its own — s_vx has 64 floats, the first 16 are read: a store behind
                             //  `if (r == 0)` is an exec-mask branch per step)
    wave_sync();
    float t0 = s_t[(4 * r + 0) * LD + c], t1 = s_t[(4 * r + 1) * LD + c], t2 = s_t[(4 * r + 2) * LD + c];
    const f32x4 vxq = *reinterpret_cast<const f32x4*>(&s_vx[4 * r]);
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));      // (all three loads unconditional: left alone, the compiler sinks one of
                                                          //  them into an exec-masked block around the select below: a branch per step)
#else
    const float t0 = Vn[1], t1 = Vn[2], t2 = Vn[0];
    const f32x4 vxq = f32x4{vxn, vxn * 0.5f, vxn * 0.25f, 0.0f};
#endif
    // (control-slot columns: the A operand carries V_x there — it becomes the row of P that is q_z - l_z, see above)
    vA0 = ucol ? vxq[0] : 0.5f * (Vn[0] + t0);
    vA1 = ucol ? vxq[1] : 0.5f * (Vn[1] + t1);
    vA2 = ucol ? vxq[2] : 0.5f * (Vn[2] + t2);
#if QT_SWEEP_PRIO
    __builtin_amdgcn_s_setprio(0);          // the next step's P / Q MFMA block (and the loads around it)
#endif
    QT_PH(4, vA0);
  };

  // record of local step `ls` (global step base + ls)
  auto load = [&](int ls) __attribute__((always_inline)) {
    if constexpr (RK4F) {
      // [A | B] of local step ls by forward mode on the matrix pipe (see MODE_FUSED_RK4 above); tiles in C layout: register s
      // of lane (r, c) = entry [tile row 4r + s][tile column c]
      StepRegs o;
      const float* tab = s_lin + ls * Rk4Tab::STEP;
      const float hdt = 0.5f * rk_dt;
      f32x4 D, T, acc;
      // every table entry of the step up front (16 LDS reads in flight at once), then the products: two accumulators per
      // stage (k = {s 0, 2} and {s 1, 3}) halve the dependent-MFMA latency of a stage
      float a2[4], a3[4], a4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        D[q] = tab[rk_cidx[q]];
        a2[q] = tab[Rk4Tab::STAGE + rk_aidx[q]];
        a3[q] = tab[2 * Rk4Tab::STAGE + rk_aidx[q]];
        a4[q] = tab[3 * Rk4Tab::STAGE + rk_aidx[q]];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {                                        // D1 = M1 (C layout)
        D[q] += rk_ccst[q];
        acc[q] = D[q];
        T[q] = fmaf(hdt, D[q], rk_eye[q]);
        a2[q] += rk_acst[q];
        a3[q] += rk_acst[q];
        a4[q] += rk_acst[q];
      }
      auto product = [&](const float* A) __attribute__((always_inline)) {
        f32x4 Da = {0.0f, 0.0f, 0.0f, 0.0f}, Db = {0.0f, 0.0f, 0.0f, 0.0f};
        Da = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0], T[0], Da, 0, 0, 0);
        Db = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1], T[1], Db, 0, 0, 0);
        Da = __builtin_amdgcn_mfma_f32_16x16x4f32(A[2], T[2], Da, 0, 0, 0);
        Db = __builtin_amdgcn_mfma_f32_16x16x4f32(A[3], T[3], Db, 0, 0, 0);
        return Da + Db;
      };
      f32x4 Dn = product(a2);
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[q] = fmaf(2.0f, Dn[q], acc[q]); T[q] = fmaf(hdt, Dn[q], rk_eye[q]); }
      Dn = product(a3);
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[q] = fmaf(2.0f, Dn[q], acc[q]); T[q] = fmaf(rk_dt, Dn[q], rk_eye[q]); }
      D = product(a4);
      const float sixth = rk_dt / 6.0f;
      o.f0 = fmaf(sixth, acc[0] + D[0], rk_eye[0]);
      o.f1 = fmaf(sixth, acc[1] + D[1], rk_eye[1]);
      o.f2 = fmaf(sixth, acc[2] + D[2], rk_eye[2]);
      o.lq = f32x4{s_lin[of_q + ls * Rk4Tab::COST_STEP], 0.0f, 0.0f, 0.0f};
      o.lz = s_lin[of_z + ls * Rk4Tab::COST_STEP];
      return o;
    } else if constexpr (FUSED) {
      StepRegs o;
      const int off = ls * Tile16FRec::STRIDE;
      const int a = of_f + (lp.dynf ? off : 0);
      o.f0 = s_lin[a + 0];
      o.f1 = s_lin[a + 1];
      o.f2 = s_lin[a + 2];
      o.lq = f32x4{s_lin[of_q + off], 0.0f, 0.0f, 0.0f};
      o.lz = s_lin[of_z + off];
      return o;
    } else if constexpr (ROWPAD) {
      // every entry unconditionally from a clamped offset (a conditional load is a branch), constants selected afterwards:
      // zero, and one on the diagonal of l_uu for a padded control (a unit pivot: zero gains, nothing else touched)
      StepRegs o;
      const float* rp = pad_base + (size_t)ls * pad_stride;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = rp[pad_of[q] < 0 ? 0 : pad_of[q]];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = pad_of[q] < 0 ? 0.0f : v[q];
      if (ucol && r >= pm) {                                  // l_uu row of a padded control: e_r
        v[3] = r == 0 ? 1.0f : 0.0f;
        v[4] = r == 1 ? 1.0f : 0.0f;
        v[5] = r == 2 ? 1.0f : 0.0f;
        v[6] = r == 3 ? 1.0f : 0.0f;
      }
      o.f0 = v[0];
      o.f1 = v[1];
      o.f2 = v[2];
      o.lq = f32x4{v[3], v[4], v[5], v[6]};
      o.lz = v[7];
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
  if constexpr (RK4F) {
    float* coef = fa.coef + (size_t)b * S * Rk4Coef::STRIDE;
    // (1) one lane per step: stage points -> coefficient records in the global scratch
#ifndef QT_X_NO_PHASE1
    for (int c0 = 0; c0 < S; c0 += QT_WAVE) {
      const int ls = c0 + lane;
      if (ls < S) {
        const int t = fa.t_start + ls;
        const float4* px = reinterpret_cast<const float4*>(fa.x + ((size_t)b * (fa.N + 1) + t) * 12);
        const float4 xa = px[0], xb = px[1], xc = px[2];
        const float4 ua = *reinterpret_cast<const float4*>(fa.u + ((size_t)b * fa.N + t) * 4);
        const float xs[12] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w, xc.x, xc.y, xc.z, xc.w};
        const float us[4] = {ua.x, ua.y, ua.z, ua.w};
        float4* dst = reinterpret_cast<float4*>(coef) + ls;
        rk4_step_coefs(fa.p, xs, us, [&](int q, float4 v) __attribute__((always_inline)) { dst[(size_t)q * S] = v; });
      }
    }
#endif
    // the records are read back by other lanes of this same wave: complete (written through to L2) before any is loaded
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // (2) batches of RK4_BATCH steps: the batch's coefficient records -> LDS tables (the recursion's load() does the rest)
    const float4* coef4 = reinterpret_cast<const float4*>(coef);
    for (int i = lane; i < Rk4Tab::FLOATS; i += QT_WAVE) s_lin[i] = 0.0f;      // (the ZERO slots stay zero for good)
    const int top = ((S - 1) / RK4_BATCH) * RK4_BATCH;
    for (int base = top; base >= 0; base -= RK4_BATCH) {
      const int cnt = S - base < RK4_BATCH ? S - base : RK4_BATCH;
      wave_sync();                                       // the previous batch's tables are no longer read
      for (int e = lane; e < cnt * Rk4Coef::PIECES; e += QT_WAVE) {
        const int sl = e / Rk4Coef::PIECES, q = e - sl * Rk4Coef::PIECES;
        const float4 v = coef4[(size_t)q * S + base + sl];
        float* dst = q < 28 ? s_lin + sl * Rk4Tab::STEP + (q / 7) * Rk4Tab::STAGE + 4 * (q % 7)
                            : s_lin + Rk4Tab::COST + sl * Rk4Tab::COST_STEP + 4 * (q - 28);
        *reinterpret_cast<float4*>(dst) = v;
      }
      wave_sync();
      run(cnt, base);
    }
  } else if constexpr (FUSED) {
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
