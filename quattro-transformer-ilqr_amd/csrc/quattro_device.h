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
