// Shared device-side definitions for libquattro_hip (gfx950 only).
//   - derivative-record layouts (ROWMAJOR, TILE16) as constexpr offset maps
//   - wave-level helpers (64-lane wavefronts)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/quattro_hip.h"

#define QT_WAVE 64

#define QT_HD __host__ __device__ __forceinline__

// ----------------------------------------------------------------------------------------------
// ROWMAJOR record: [A | B | l_xx | l_ux | l_uu | l_x | l_u], stride padded to a multiple of 4 floats
// so every record starts 16-byte aligned.
// ----------------------------------------------------------------------------------------------
template <int NX, int NU>
struct RowMajorRec {
  static constexpr int NZ = NX + NU;
  static constexpr int A = 0;
  static constexpr int B = A + NX * NX;
  static constexpr int LXX = B + NX * NU;
  static constexpr int LUX = LXX + NX * NX;
  static constexpr int LUU = LUX + NU * NX;
  static constexpr int LX = LUU + NU * NU;
  static constexpr int LU = LX + NX;
  static constexpr int SIZE = LU + NU;
  static constexpr int STRIDE = (SIZE + 3) / 4 * 4;
  static QT_HD int a(int i, int j) { return A + i * NX + j; }
  static QT_HD int b(int i, int a_) { return B + i * NU + a_; }
  static QT_HD int lxx(int i, int j) { return LXX + i * NX + j; }
  static QT_HD int lux(int a_, int j) { return LUX + a_ * NX + j; }
  static QT_HD int luu(int a_, int b_) { return LUU + a_ * NU + b_; }
  static QT_HD int lx(int i) { return LX + i; }
  static QT_HD int lu(int a_) { return LU + a_; }
};

// ----------------------------------------------------------------------------------------------
// TILE16 record (n = 12, m = 4): the 416 floats in the order the 64 lanes of the sweep wave consume them.
//
// The sweep works on 16x16 tiles indexed by the augmented variable z = (x, u), n + m = 16, one tile column
// per lane-in-row c = lane & 15 and four tile rows per lane group r = lane >> 4 (the C/D layout of
// v_mfma_f32_16x16x4_f32: element [4r + s][c] sits in accumulator register s of lane 16r + c).
// Tile index q = 4g + s' holds x_{3g+s'} for s' < 3 and u_g for s' = 3, so every lane group owns three
// state rows and one control row.
//
//   [  0,192)  F = [A | B] (12 x 16):  lane l = 16r + c keeps F[3r + s][z(c)], s = 0..2, at 3l + s
//   [192,384)  lane (r, x-column j):   { l_xx[3r][j], l_xx[3r+1][j], l_xx[3r+2][j], l_ux[r][j] } at 192 + 4(12r + j)
//   [384,400)  l_uu row-major:         l_uu[r][a] at 384 + 4r + a
//   [400,416)  l_z = (l_x, l_u) in natural z order
// ----------------------------------------------------------------------------------------------
struct Tile16Rec {
  static constexpr int NX = 12, NU = 4;
  static constexpr int F = 0, LXB = 192, LUU = 384, LZ = 400, SIZE = 416, STRIDE = 416;
  // natural z index (0..11 = x, 12..15 = u) -> tile column
  static QT_HD int zcol(int z) { return z < 12 ? (z / 3) * 4 + z % 3 : 4 * (z - 12) + 3; }
  static QT_HD int f(int i, int z) { return F + 3 * (16 * (i / 3) + zcol(z)) + i % 3; }
  static QT_HD int a(int i, int j) { return f(i, j); }
  static QT_HD int b(int i, int a_) { return f(i, 12 + a_); }
  static QT_HD int lxx(int i, int j) { return LXB + 4 * (12 * (i / 3) + j) + i % 3; }
  static QT_HD int lux(int a_, int j) { return LXB + 4 * (12 * a_ + j) + 3; }
  static QT_HD int luu(int a_, int b_) { return LUU + 4 * a_ + b_; }
  static QT_HD int lx(int i) { return LZ + i; }
  static QT_HD int lu(int a_) { return LZ + 12 + a_; }
};

// ----------------------------------------------------------------------------------------------
// TILE16C record: the TILE16 record of the Euler-discretised quadrotor with everything that does not depend on
// (x, u) taken out.  Of the 416 floats only those in 14 of the 64 F lanes, l_uu and l_z change from step to step
// (v' reads the angles and the thrust, the Euler-angle rates read phi, theta and omega, omega' reads omega; A's
// identity and dt entries, B's torque rows, l_xx = 2Q and l_ux = 0 are constants of the problem).  The constants
// live ONCE in a header record (a plain TILE16 record, HEADER floats at the start of the buffer, L2-resident for the
// whole sweep); per (b,t) only 76 floats = 304 B are written by the linearisation and streamed by the sweep instead
// of 1,664 B.  The sweep's arithmetic is unchanged: a lane's three loads per step simply point either into the
// header (stride 0) or into the compact record.
//
//   [ 0,42)  F entries of the dynamic lanes, 3 per lane in the order of DYN_LANE   (lane l = 16r + c <-> F[3r+s][z(c)])
//   [42,44)  padding; 43 doubles as the sink for writes of constant entries (fill_const / fill_state are shared code)
//   [44,60)  l_uu row-major
//   [60,76)  l_z = (l_x, l_u)
// ----------------------------------------------------------------------------------------------
struct Tile16CRec {
  static constexpr int NX = 12, NU = 4;
  static constexpr int F = 0, NDYN = 14, DUMP = 43, LUU = 44, LZ = 60, SIZE = 76, STRIDE = 76;
  static constexpr int HEADER = Tile16Rec::STRIDE;
  // rows 3-5 x (phi, theta, psi, u0..u3); rows 6-8 x (phi, theta, q, r); rows 9-11 x (p, q, r)
  static QT_HD int dyn_index(int lane) {
    switch (lane) {
      case 19: return 0;  case 23: return 1;  case 24: return 2;  case 25: return 3;  case 26: return 4;
      case 27: return 5;  case 31: return 6;  case 40: return 7;  case 41: return 8;  case 45: return 9;
      case 46: return 10; case 60: return 11; case 61: return 12; case 62: return 13;
      default: return -1;
    }
  }
  static QT_HD int f(int i, int z) {
    const int d = dyn_index(16 * (i / 3) + Tile16Rec::zcol(z));
    return d < 0 ? DUMP : F + 3 * d + i % 3;
  }
  static QT_HD int a(int i, int j) { return f(i, j); }
  static QT_HD int b(int i, int a_) { return f(i, 12 + a_); }
  static QT_HD int lxx(int, int) { return DUMP; }
  static QT_HD int lux(int, int) { return DUMP; }
  static QT_HD int luu(int a_, int b_) { return LUU + 4 * a_ + b_; }
  static QT_HD int lx(int i) { return LZ + i; }
  static QT_HD int lu(int a_) { return LZ + 12 + a_; }
};

// ----------------------------------------------------------------------------------------------
// TILE16F record: the TILE16C record as the fused linearise+sweep kernel keeps it in LDS (never in HBM).  l_uu is
// diagonal for the built-in cost, so only its diagonal is kept: 64 floats of payload, pitch 68 (16-B aligned records
// whose starts fall on 16 different LDS banks), which is what lets 25 steps' records + the header fit the 10 KB a
// wave may use with 16 workgroups on a CU.
//   [ 0,42)  F entries of the dynamic lanes (as TILE16C)   43  sink   [44,48)  diag(l_uu)   [48,64)  l_z
// ----------------------------------------------------------------------------------------------
struct Tile16FRec {
  static constexpr int NX = 12, NU = 4;
  static constexpr int F = 0, DUMP = 43, LUUD = 44, LZ = 48, SIZE = 64, STRIDE = 68;
  static QT_HD int f(int i, int z) {
    const int d = Tile16CRec::dyn_index(16 * (i / 3) + Tile16Rec::zcol(z));
    return d < 0 ? DUMP : F + 3 * d + i % 3;
  }
  static QT_HD int a(int i, int j) { return f(i, j); }
  static QT_HD int b(int i, int a_) { return f(i, 12 + a_); }
  static QT_HD int lxx(int, int) { return DUMP; }
  static QT_HD int lux(int, int) { return DUMP; }
  static QT_HD int luu(int a_, int b_) { return a_ == b_ ? LUUD + a_ : DUMP; }
  static QT_HD int lx(int i) { return LZ + i; }
  static QT_HD int lu(int a_) { return LZ + 12 + a_; }
};

// ----------------------------------------------------------------------------------------------
// TILE16R record: the TILE16 record of the quadrotor under RK4 with everything that does not change from step to step
// taken out.  The rate function is p' = v, v' = f(angles, u), angles' = g(angles, omega), omega' = h(omega, u), so through
// all four stages the position directions of z never reach anything (their columns of [A | B] stay unit vectors) and a
// velocity direction reaches only the position it integrates into (column = e_v + dt e_p): 6 of the 16 columns are constants
// of the problem and live in the header record (a plain TILE16 record, with l_xx = 2Q and l_ux = 0 of the built-in cost);
// the other 10 — angles, body rates, controls — are dense and change every step.  Per (b,t): those 10 columns, COLUMN by
// column (the producer finishes one direction at a time and stores it as three 16-byte pieces; sweep lane 16r + c finds its
// triple F[3r..3r+2][z(c)] as three consecutive floats), l_uu, l_z and a sink slot for the shared record-filling code:
// 156 floats = 624 B (912 B with all 16 columns stored, rounds 2-3a; 1,664 B for a plain TILE16 record).
//   [  0,120)  F columns of (angle 0..2, rate 0..2, control 0..3), 12 floats each   [120,136)  l_uu row-major
//   [136,152)  l_z = (l_x, l_u)            [152,156)  sink / padding
// ----------------------------------------------------------------------------------------------
struct Tile16RRec {
  static constexpr int NX = 12, NU = 4;
  static constexpr int F = 0, NCOL = 10, LUU = 120, LZ = 136, DUMP = 152, SIZE = 156, STRIDE = 156;
  static constexpr int HEADER = Tile16Rec::STRIDE;
  // tile column c (Tile16Rec::zcol) -> stored column, or -1 for a position / velocity direction (constant: header record)
  static QT_HD int col_index(int c) {
    const int g = c >> 2, sp = c & 3;
    return sp == 3 ? 6 + g : (g >= 2 ? 3 * (g - 2) + sp : -1);
  }
  static QT_HD int f(int i, int z) {
    const int d = col_index(Tile16Rec::zcol(z));
    return d < 0 ? DUMP : F + 12 * d + i;
  }
  static QT_HD int a(int i, int j) { return f(i, j); }
  static QT_HD int b(int i, int a_) { return f(i, 12 + a_); }
  static QT_HD int lxx(int, int) { return DUMP; }
  static QT_HD int lux(int, int) { return DUMP; }
  static QT_HD int luu(int a_, int b_) { return LUU + 4 * a_ + b_; }
  static QT_HD int lx(int i) { return LZ + i; }
  static QT_HD int lu(int a_) { return LZ + 12 + a_; }
};

// ----------------------------------------------------------------------------------------------
// wave helpers
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float qt_readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// sum over the four 16-lane rows of a wave: every lane ends with v(c) + v(c+16) + v(c+32) + v(c+48).
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange whole rows, no LDS involved.
__device__ __forceinline__ float qt_sum_rows(float v) {
  int vi = __float_as_int(v);
  auto s16 = __builtin_amdgcn_permlane16_swap(vi, vi, false, false);
  float a = __int_as_float(s16[0]) + __int_as_float(s16[1]);
  int ai = __float_as_int(a);
  auto s32 = __builtin_amdgcn_permlane32_swap(ai, ai, false, false);
  return __int_as_float(s32[0]) + __int_as_float(s32[1]);
}

__device__ __forceinline__ bool qt_finite(float v) { return fabsf(v) <= 3.0e38f; }
