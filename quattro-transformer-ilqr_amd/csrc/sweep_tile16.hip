// Quadrotor-shaped Riccati-like sweep (n = 12, m = 4, n + m = 16): one wavefront per trajectory, the whole
// recursion kept in the wave's registers as 16x16 tiles.
//
// Arithmetic replaced: iLQR_TF.backward_pass / backward_pass_segment of the reference
// (quattro_ilqr_tf/quattro_ilqr_tf.py:297-317, :343-364).
//
// Tile layout (the C/D layout of v_mfma_f32_16x16x4_f32): lane l = 16r + c holds elements [4r + s][c], s = 0..3,
// of a 16x16 matrix over the augmented index z = (x, u); tile index 4g + s' is x_{3g+s'} for s' < 3 and u_g for
// s' = 3 (Tile16Rec in quattro_device.h).  With that interleaving
//   * [A|B] (12 x 16) is three registers per lane, read straight from the TILE16 record (coalesced, no LDS),
//   * P = V_xx [A|B] and Q = L_zz + [A|B]^T P are 3 + 3 exact-fp32 MFMAs (v_mfma_f32_16x16x4_f32 is a k-ordered
//     fmaf chain, bit-identical to VALU fp32): the accumulator of the first product is already the B operand of
//     the second, and the symmetric V_xx in accumulator layout is already the A operand of the next step,
//   * control row u_r of Q (Q_ux | Q_uu) sits in accumulator register 3 of lane group r, which is exactly the
//     one-element-per-lane operand layout of the rank-4 update V_xx' = Q_xx + (Q_ux - reg K)^T K (1 MFMA).
// Cross-lane work: row sums by v_permlane16/32_swap, Q_uu gathered with v_readlane, one LDS round trip per step
// for the V_xx' transpose (symmetrisation) and the V_x redistribution.
//
// V update: the reference computes Q_xx + K^T Q_uu K + K^T Q_ux + Q_ux^T K with K = -(Q_uu + reg I)^-1 Q_ux.
// Since (Q_uu + reg I) K = -Q_ux exactly, Q_uu K + Q_ux = -reg K, so the same quantity is
// Q_xx + (Q_ux - reg K)^T K (and V_x' = Q_x + (Q_ux - reg K)^T k): algebraically identical, one product
// instead of three, and free of the fp32 cancellation in Q_uu K + Q_ux.  Parity vs the fp64 oracle that
// evaluates the reference's 4-term form: tests/test_sweep_gpu.py.
#include "quattro_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct StepRegs {
  float f0, f1, f2;  // F[3r+s][z(c)]
  f32x4 lq;          // C-input of Q: (l_xx[3r..3r+2][j], l_ux[r][j]) or (0,0,0,l_uu[r][g]) for control columns
  float lz;          // l_z[z(c)]
};

__device__ __forceinline__ StepRegs load_step(const float* __restrict__ p, int lane, int r, int g, bool ucol, int xj) {
  StepRegs o;
  const float* pf = p + Tile16Rec::F + 3 * lane;
  o.f0 = pf[0];
  o.f1 = pf[1];
  o.f2 = pf[2];
  if (!ucol) {
    o.lq = *reinterpret_cast<const f32x4*>(p + Tile16Rec::LXB + 4 * (12 * r + xj));
    o.lz = p[Tile16Rec::LZ + xj];
  } else {
    o.lq = f32x4{0.0f, 0.0f, 0.0f, p[Tile16Rec::LUU + 4 * r + g]};
    o.lz = p[Tile16Rec::LZ + 12 + g];
  }
  return o;
}

// general 4x4 inverse by 2x2 minors (no symmetry assumed: the reference inverts Q_uu + reg I as it is).
// All operands are wave-uniform.  Returns the determinant.
__device__ __forceinline__ float inverse4(const float (&a)[4][4], float (&w)[4][4]) {
  const float s0 = a[0][0] * a[1][1] - a[1][0] * a[0][1];
  const float s1 = a[0][0] * a[1][2] - a[1][0] * a[0][2];
  const float s2 = a[0][0] * a[1][3] - a[1][0] * a[0][3];
  const float s3 = a[0][1] * a[1][2] - a[1][1] * a[0][2];
  const float s4 = a[0][1] * a[1][3] - a[1][1] * a[0][3];
  const float s5 = a[0][2] * a[1][3] - a[1][2] * a[0][3];
  const float c5 = a[2][2] * a[3][3] - a[3][2] * a[2][3];
  const float c4 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
  const float c3 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
  const float c2 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
  const float c1 = a[2][0] * a[3][2] - a[3][0] * a[2][2];
  const float c0 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
  const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  const float id = 1.0f / det;
  w[0][0] = (a[1][1] * c5 - a[1][2] * c4 + a[1][3] * c3) * id;
  w[0][1] = (-a[0][1] * c5 + a[0][2] * c4 - a[0][3] * c3) * id;
  w[0][2] = (a[3][1] * s5 - a[3][2] * s4 + a[3][3] * s3) * id;
  w[0][3] = (-a[2][1] * s5 + a[2][2] * s4 - a[2][3] * s3) * id;
  w[1][0] = (-a[1][0] * c5 + a[1][2] * c2 - a[1][3] * c1) * id;
  w[1][1] = (a[0][0] * c5 - a[0][2] * c2 + a[0][3] * c1) * id;
  w[1][2] = (-a[3][0] * s5 + a[3][2] * s2 - a[3][3] * s1) * id;
  w[1][3] = (a[2][0] * s5 - a[2][2] * s2 + a[2][3] * s1) * id;
  w[2][0] = (a[1][0] * c4 - a[1][1] * c2 + a[1][3] * c0) * id;
  w[2][1] = (-a[0][0] * c4 + a[0][1] * c2 - a[0][3] * c0) * id;
  w[2][2] = (a[3][0] * s4 - a[3][1] * s2 + a[3][3] * s0) * id;
  w[2][3] = (-a[2][0] * s4 + a[2][1] * s2 - a[2][3] * s0) * id;
  w[3][0] = (-a[1][0] * c3 + a[1][1] * c1 - a[1][2] * c0) * id;
  w[3][1] = (a[0][0] * c3 - a[0][1] * c1 + a[0][2] * c0) * id;
  w[3][2] = (-a[3][0] * s3 + a[3][1] * s1 - a[3][2] * s0) * id;
  w[3][3] = (a[2][0] * s3 - a[2][1] * s1 + a[2][2] * s0) * id;
  return det;
}

__device__ __forceinline__ float sel4(int r, float a0, float a1, float a2, float a3) {
  return r == 0 ? a0 : (r == 1 ? a1 : (r == 2 ? a2 : a3));
}

constexpr int LD = 20;  // LDS row pitch (floats) of the 16x16 transpose tile: 16-B aligned rows, conflict-free b128 writes

__global__ __launch_bounds__(QT_WAVE) void sweep_tile16_kernel(const float* __restrict__ rec,
                                                               const float* __restrict__ VxN,
                                                               const float* __restrict__ VxxN, int S, float reg,
                                                               float* __restrict__ Kout, float* __restrict__ kout,
                                                               int32_t* __restrict__ status,
                                                               const int32_t* __restrict__ active) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  if (active != nullptr && active[b] == 0) return;
  const int r = lane >> 4, c = lane & 15, g = c >> 2, sp = c & 3;
  const bool ucol = (sp == 3);
  const int xj = 3 * g + (ucol ? 0 : sp);  // state index of this lane's tile column (unused for control columns)

  __shared__ __attribute__((aligned(16))) float s_t[16 * LD];
  __shared__ __attribute__((aligned(16))) float s_vx[16];

  // terminal values: A-operand layout of V_xx is lane(r,c) = V[x_j][3r+s]; used as given (not symmetrised)
  float vA0 = 0.0f, vA1 = 0.0f, vA2 = 0.0f;
  if (!ucol) {
    const float* pv = VxxN + (size_t)b * 144 + xj * 12 + 3 * r;
    vA0 = pv[0];
    vA1 = pv[1];
    vA2 = pv[2];
  }
  float vx0 = VxN[(size_t)b * 12 + 3 * r + 0], vx1 = VxN[(size_t)b * 12 + 3 * r + 1], vx2 = VxN[(size_t)b * 12 + 3 * r + 2];

  const float* base = rec + (size_t)b * S * Tile16Rec::STRIDE;
  StepRegs cur = load_step(base + (size_t)(S - 1) * Tile16Rec::STRIDE, lane, r, g, ucol, xj);
  bool bad = false, singular = false;

  for (int s = S - 1; s >= 0; --s) {
    StepRegs nxt = cur;
    if (s > 0) nxt = load_step(base + (size_t)(s - 1) * Tile16Rec::STRIDE, lane, r, g, ucol, xj);

    // P = V_xx F   (tile rows 4r+s <-> x_{3r+s}; the control slot s = 3 contributes nothing)
    f32x4 P = {0.0f, 0.0f, 0.0f, 0.0f};
    P = __builtin_amdgcn_mfma_f32_16x16x4f32(vA0, cur.f0, P, 0, 0, 0);
    P = __builtin_amdgcn_mfma_f32_16x16x4f32(vA1, cur.f1, P, 0, 0, 0);
    P = __builtin_amdgcn_mfma_f32_16x16x4f32(vA2, cur.f2, P, 0, 0, 0);
    // Q = L_zz + F^T P
    f32x4 Q = cur.lq;
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f0, P[0], Q, 0, 0, 0);
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f1, P[1], Q, 0, 0, 0);
    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.f2, P[2], Q, 0, 0, 0);
    // q_z = l_z + F^T V_x   (valid in every lane group after the row sum)
    const float qz = cur.lz + qt_sum_rows(fmaf(cur.f0, vx0, fmaf(cur.f1, vx1, cur.f2 * vx2)));

    // Q_uu (+ reg on the diagonal) and Q_u to wave-uniform values
    const float q3 = Q[3];
    float M[4][4], W[4][4], qu[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int e = 0; e < 4; ++e) M[a][e] = qt_readlane(q3, 16 * a + 4 * e + 3) + (a == e ? reg : 0.0f);
      qu[a] = qt_readlane(qz, 4 * a + 3);
    }
    const float det = inverse4(M, W);
    singular = singular || !(det != 0.0f) || !qt_finite(det);

    // feed-forward k = -W Q_u (uniform); this lane group's element k_r
    float kv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) kv[a] = -(W[a][0] * qu[0] + W[a][1] * qu[1] + W[a][2] * qu[2] + W[a][3] * qu[3]);
    const float kr = sel4(r, kv[0], kv[1], kv[2], kv[3]);

    // feedback K[r][j] = -sum_a W[r][a] Q_ux[a][j]: column j of Q_ux is spread over the four lane groups
    const float qa0 = __shfl(q3, c), qa1 = __shfl(q3, 16 + c), qa2 = __shfl(q3, 32 + c), qa3 = __shfl(q3, 48 + c);
    const float w0 = sel4(r, W[0][0], W[1][0], W[2][0], W[3][0]);
    const float w1 = sel4(r, W[0][1], W[1][1], W[2][1], W[3][1]);
    const float w2 = sel4(r, W[0][2], W[1][2], W[2][2], W[3][2]);
    const float w3 = sel4(r, W[0][3], W[1][3], W[2][3], W[3][3]);
    float Kv = -(w0 * qa0 + w1 * qa1 + w2 * qa2 + w3 * qa3);
    Kv = ucol ? 0.0f : Kv;
    const float E = ucol ? 0.0f : fmaf(-reg, Kv, q3);   // (Q_ux - reg K)[r][j]
    bad = bad || !qt_finite(Kv) || !qt_finite(kr);

    // outputs: K [m][n] row-major, k [m]
    const size_t o = (size_t)b * S + s;
    if (!ucol) Kout[o * 48 + r * 12 + xj] = Kv;
    if (c == 3) kout[o * 4 + r] = kr;

    // V_xx' = Q_xx + E^T K ; V_x' = Q_x + E^T k
    f32x4 Vn = __builtin_amdgcn_mfma_f32_16x16x4f32(E, Kv, Q, 0, 0, 0);
    const float vxn = qz + qt_sum_rows(E * kr);
    // drop the control rows / columns of the tile
    Vn[3] = 0.0f;
    if (ucol) Vn = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // symmetrise through LDS: write the tile transposed, read it back in place
    __syncthreads();  // previous step's reads are done
    *reinterpret_cast<f32x4*>(&s_t[c * LD + 4 * r]) = Vn;
    if (r == 0) s_vx[c] = vxn;
    __syncthreads();
    const float t0 = s_t[(4 * r + 0) * LD + c], t1 = s_t[(4 * r + 1) * LD + c], t2 = s_t[(4 * r + 2) * LD + c];
    const f32x4 vxq = *reinterpret_cast<const f32x4*>(&s_vx[4 * r]);
    vA0 = 0.5f * (Vn[0] + t0);
    vA1 = 0.5f * (Vn[1] + t1);
    vA2 = 0.5f * (Vn[2] + t2);
    vx0 = vxq[0];
    vx1 = vxq[1];
    vx2 = vxq[2];
    cur = nxt;
  }
  if (status != nullptr) {
    const bool any_bad = __any(bad);
    if (lane == 0) status[b] = (any_bad ? QUATTRO_TRAJ_NONFINITE : 0) | (singular ? QUATTRO_TRAJ_SINGULAR : 0);
  }
}

}  // namespace

int quattro_launch_sweep_tile16(const float* rec, const float* VxN, const float* VxxN, int B, int S, float reg,
                                float* K, float* k, int32_t* status, const int32_t* active, hipStream_t stream) {
  hipLaunchKernelGGL(sweep_tile16_kernel, dim3(B), dim3(QT_WAVE), 0, stream, rec, VxN, VxxN, S, reg, K, k, status,
                     active);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
