// Device-side body of the generic Riccati-like sweep (one wavefront per trajectory, blocks staged in LDS), shared by
// sweep_generic.hip (one launch per call) and solve_user.hip (the device-resident loop of a user-compiled model).
// Design notes: sweep_generic.hip.
#pragma once
#include "quattro_device.h"

namespace {


// in-register inverse of an M x M matrix by Gauss-Jordan with partial pivoting; every lane holds the same
// matrix, so control flow is wave-uniform and all indices are compile-time constants after unrolling.
template <int M>
__device__ __forceinline__ bool invert_gj(float (&a)[M][M], float (&w)[M][M]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < M; ++i)
#pragma unroll
    for (int j = 0; j < M; ++j) w[i][j] = (i == j) ? 1.0f : 0.0f;
#pragma unroll
  for (int p = 0; p < M; ++p) {
    // bring the largest |a[r][p]|, r >= p, into row p (bubble it up: static indices only)
#pragma unroll
    for (int r = p + 1; r < M; ++r) {
      const bool sw = fabsf(a[r][p]) > fabsf(a[p][p]);
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const float ap = a[p][j], ar = a[r][j], wp = w[p][j], wr = w[r][j];
        a[p][j] = sw ? ar : ap;
        a[r][j] = sw ? ap : ar;
        w[p][j] = sw ? wr : wp;
        w[r][j] = sw ? wp : wr;
      }
    }
    const float piv = a[p][p];
    ok = ok && (piv != 0.0f) && qt_finite(piv);
    const float ip = 1.0f / piv;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      a[p][j] *= ip;
      w[p][j] *= ip;
    }
#pragma unroll
    for (int r = 0; r < M; ++r) {
      if (r == p) continue;
      const float f = a[r][p];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        a[r][j] -= f * a[p][j];
        w[r][j] -= f * w[p][j];
      }
    }
  }
  return ok;
}

// trajectory b by the 64 lanes of ONE wave (a workgroup of its own: the __syncthreads below are that wave's)
template <int NX, int NU>
__device__ __forceinline__ void sweep_generic_body(const float* __restrict__ rec, const float* __restrict__ VxN,
                                                   const float* __restrict__ VxxN, int S, float reg,
                                                   float* __restrict__ Kout, float* __restrict__ kout,
                                                   int32_t* __restrict__ status, const int b, const int lane) {
  using R = RowMajorRec<NX, NU>;
  constexpr int NZ = NX + NU;
  constexpr int NC = NX + 1;                      // columns of [K | k]
  constexpr int CHUNKS = R::STRIDE / 4;           // float4 chunks per record
  constexpr int CPL = (CHUNKS + QT_WAVE - 1) / QT_WAVE;

  __shared__ __attribute__((aligned(16))) float s_rec[R::STRIDE];
  __shared__ float s_V[NX * NX], s_P[NX * NZ], s_Q[NZ * NZ], s_qz[NZ], s_vx[NX];
  __shared__ float s_Kk[NU * NC], s_G[NU * NC], s_Vn[NX * NX];

  for (int o = lane; o < NX * NX; o += QT_WAVE) s_V[o] = VxxN[(size_t)b * NX * NX + o];
  if (lane < NX) s_vx[lane] = VxN[(size_t)b * NX + lane];

  const float4* rec4 = reinterpret_cast<const float4*>(rec + (size_t)b * S * R::STRIDE);
  float4 pre[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int ch = lane + i * QT_WAVE;
    if (ch < CHUNKS) pre[i] = rec4[(size_t)(S - 1) * CHUNKS + ch];
  }
  bool bad = false, singular = false;

  auto F = [&](int k, int j) -> float { return j < NX ? s_rec[R::a(k, j)] : s_rec[R::b(k, j - NX)]; };

  for (int s = S - 1; s >= 0; --s) {
    // 1. record -> LDS, prefetch the next (earlier) step
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int ch = lane + i * QT_WAVE;
      if (ch < CHUNKS) reinterpret_cast<float4*>(s_rec)[ch] = pre[i];
    }
    if (s > 0) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int ch = lane + i * QT_WAVE;
        if (ch < CHUNKS) pre[i] = rec4[(size_t)(s - 1) * CHUNKS + ch];
      }
    }
    __syncthreads();
    // 2. P = V_xx [A|B]
    for (int o = lane; o < NX * NZ; o += QT_WAVE) {
      const int i = o / NZ, j = o % NZ;
      float acc = 0.0f;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) acc = fmaf(s_V[i * NX + kk], F(kk, j), acc);
      s_P[o] = acc;
    }
    __syncthreads();
    // 3. Q = L_zz + [A|B]^T P (the x-row / u-column block is never used), q_z = l_z + [A|B]^T V_x
    for (int o = lane; o < NZ * NZ + NZ; o += QT_WAVE) {
      if (o < NZ * NZ) {
        const int i = o / NZ, j = o % NZ;
        if (i < NX && j >= NX) continue;
        float acc = (i < NX) ? s_rec[R::lxx(i, j)] : (j < NX ? s_rec[R::lux(i - NX, j)] : s_rec[R::luu(i - NX, j - NX)]);
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) acc = fmaf(F(kk, i), s_P[kk * NZ + j], acc);
        s_Q[o] = acc;
      } else {
        const int j = o - NZ * NZ;
        float acc = (j < NX) ? s_rec[R::lx(j)] : s_rec[R::lu(j - NX)];
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) acc = fmaf(F(kk, j), s_vx[kk], acc);
        s_qz[j] = acc;
      }
    }
    __syncthreads();
    // 4. W = inv(Q_uu + reg I) in registers (same in every lane); lanes 0..NX own one column of [K | k]
    float mreg[NU][NU], w[NU][NU];
#pragma unroll
    for (int a = 0; a < NU; ++a)
#pragma unroll
      for (int c = 0; c < NU; ++c) mreg[a][c] = s_Q[(NX + a) * NZ + NX + c] + (a == c ? reg : 0.0f);
    singular = singular || !invert_gj<NU>(mreg, w);
    if (lane < NC) {
      float q[NU];
#pragma unroll
      for (int c = 0; c < NU; ++c) q[c] = (lane < NX) ? s_Q[(NX + c) * NZ + lane] : s_qz[NX + c];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < NU; ++c) acc = fmaf(w[a][c], q[c], acc);
        acc = -acc;
        bad = bad || !qt_finite(acc);
        s_Kk[a * NC + lane] = acc;
        if (lane < NX)
          Kout[((size_t)b * S + s) * NU * NX + a * NX + lane] = acc;
        else
          kout[((size_t)b * S + s) * NU + a] = acc;
      }
    }
    __syncthreads();
    // 5. G = Q_uu [K|k] + [Q_ux | Q_u]
    for (int o = lane; o < NU * NC; o += QT_WAVE) {
      const int a = o / NC, j = o % NC;
      float acc = (j < NX) ? s_Q[(NX + a) * NZ + j] : s_qz[NX + a];
#pragma unroll
      for (int c = 0; c < NU; ++c) acc = fmaf(s_Q[(NX + a) * NZ + NX + c], s_Kk[c * NC + j], acc);
      s_G[o] = acc;
    }
    __syncthreads();
    // 6. V_xx' = Q_xx + K^T G + Q_ux^T K ; V_x' = Q_x + K^T g + Q_ux^T k
    for (int o = lane; o < NX * NX + NX; o += QT_WAVE) {
      if (o < NX * NX) {
        const int i = o / NX, j = o % NX;
        float acc = s_Q[i * NZ + j];
#pragma unroll
        for (int a = 0; a < NU; ++a) acc = fmaf(s_Kk[a * NC + i], s_G[a * NC + j], acc);
#pragma unroll
        for (int a = 0; a < NU; ++a) acc = fmaf(s_Q[(NX + a) * NZ + i], s_Kk[a * NC + j], acc);
        s_Vn[o] = acc;
      } else {
        const int i = o - NX * NX;
        float acc = s_qz[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) acc = fmaf(s_Kk[a * NC + i], s_G[a * NC + NX], acc);
#pragma unroll
        for (int a = 0; a < NU; ++a) acc = fmaf(s_Q[(NX + a) * NZ + i], s_Kk[a * NC + NX], acc);
        s_vx[i] = acc;   // s_vx was last read in phase 3
      }
    }
    __syncthreads();
    // 7. symmetrise
    for (int o = lane; o < NX * NX; o += QT_WAVE) {
      const int i = o / NX, j = o % NX;
      s_V[o] = 0.5f * (s_Vn[o] + s_Vn[j * NX + i]);
    }
    __syncthreads();
  }
  if (status != nullptr) {
    const bool any_bad = __any(bad);
    if (lane == 0) status[b] = (any_bad ? QUATTRO_TRAJ_NONFINITE : 0) | (singular ? QUATTRO_TRAJ_SINGULAR : 0);
  }
}

}  // namespace
