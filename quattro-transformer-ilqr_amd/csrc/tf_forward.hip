// Transformer gain predictor, fused forward for a batch of sequences, bf16 MFMA with fp32 accumulation.
//
// Arithmetic replaced (reference): TransformerPredictor.forward quattro_ilqr_tf/transformer_model.py:122-138
// (embeddings :125-131, PositionalEncoding :77-80, 3 x nn.TransformerEncoderLayer as configured at :106-113:
// batch_first, post-LayerNorm, ReLU, eps 1e-5, causal mask :135, output_linear :138) and the normalise /
// de-normalise steps of TransformerILQR.predict quattro_ilqr_tf/transformer_ilqr.py:312-324.
//
// MI355X design: the model is tiny (0.6 M parameters, L <= 128 tokens, d = 128), so ONE workgroup (4 waves) runs the
// WHOLE forward of one sequence with every activation resident on the CU: the fp32 residual stream lives in
// accumulator registers, its bf16 image (the MFMA operand) in LDS, and only weights (L2-resident, 1.2 MB) and the
// 12.8 KB of per-sequence input/output touch memory.  Layer-by-layer GEMM kernels would move ~2 GB per layer
// through HBM at B = 4096.
//
// Decomposition: wave w owns feature slice [32w, 32w+32) of every d-wide activation for ALL tokens, and head w of the
// attention.  Every projection is computed TRANSPOSED (features x tokens) with v_mfma_f32_32x32x16_bf16:
//   A = weight rows, read straight from global in PyTorch's [out][in] layout (16 B per lane, each wave reads only
//       its own rows: no weight byte is fetched twice by a workgroup),
//   B = token rows of the bf16 activation image in LDS (ds_read_b128, XOR-swizzled 256-B rows).
// The accumulator then has one token per lane and 16 features per lane in registers, which is what LayerNorm
// (per-token statistics), the residual add, bias add and the 8-byte LDS write-back all want.
// Attention never leaves registers: with Q^T, K^T (hd x tokens) and V (tokens x hd) accumulators,
//   S^T = K Q^T, softmax over keys = over registers (+ one v_permlane32_swap), O^T = V^T P^T
// all take the previous accumulator as the next operand without any lane movement (each product sums over the
// accumulator's ROW index; cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand").
#include "quattro_device.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int D = 128;        // d_model (4 waves x 32 features)
constexpr int HD = 32;        // head dim
constexpr int ROWB = D * 2;   // bytes per token row of a bf16 [token][128] LDS image

// row of a 32x32 accumulator held by register `reg` of lane half `half`
__device__ __forceinline__ constexpr int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// accumulator registers 8s..8s+7 as the 8 k-elements of the next product's operand (k order permuted identically
// for every accumulator, so two accumulator-derived operands always agree)
template <int S>
__device__ __forceinline__ bf16x8 pack8(const f32x16& a) {
  bf16x8 p;
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = (__bf16)a[8 * S + j];
  return p;
}

// weight fragment: 8 consecutive k of row n of a row-major [rows][ld] bf16 matrix
__device__ __forceinline__ bf16x8 gw_frag(const uint16_t* __restrict__ Wm, int ld, int n, int ks, int half) {
  return *reinterpret_cast<const bf16x8*>(Wm + (size_t)n * ld + 16 * ks + 8 * half);
}

// one 128-deep K panel (8 k-steps) of row n, starting at k-step ks0
struct W8 {
  bf16x8 f[8];
};
__device__ __forceinline__ W8 load_w8(const uint16_t* __restrict__ Wm, int ld, int n, int ks0, int half) {
  W8 o;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) o.f[ks] = gw_frag(Wm, ld, n, ks0 + ks, half);
  return o;
}

// activation fragment: 8 consecutive features (k-step ks, lane half) of token row `tok` of a swizzled LDS image
__device__ __forceinline__ bf16x8 lds_frag(const char* img, int tok, int ks, int half) {
  const int gran = (2 * ks + half) ^ (tok & 15);
  return *reinterpret_cast<const bf16x8*>(img + tok * ROWB + gran * 16);
}

// write one transposed accumulator tile (rows = features 32*fslice + acc_row, cols = tokens 32*tt + lc) into a
// [token][128] bf16 image: four 8-byte pieces per lane
__device__ __forceinline__ void lds_store_tile(char* img, const f32x16& a, int tt, int fslice, int lc, int half) {
  const int tok = 32 * tt + lc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bf16x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (__bf16)a[4 * q + j];
    const int gran = (4 * fslice + q) ^ (tok & 15);
    *reinterpret_cast<bf16x4*>(img + tok * ROWB + gran * 16 + 8 * half) = v;
  }
}

// acc[tt] += (weight panel) x (token rows of an LDS image) over one 128-deep K panel.  The LDS fragments of k-step
// ks+1 are requested before the MFMAs of k-step ks issue: at one wave per SIMD an LDS read waited for right before
// its MFMA costs its full latency every time (measured: 4.6 k cycles per 32-MFMA panel instead of 1 k).
template <int TT, bool W_IS_A>
__device__ __forceinline__ void gemm_panel(f32x16 (&acc)[TT], const W8& wp, const char* img, int lc, int half) {
  bf16x8 fr[2][TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) fr[0][tt] = lds_frag(img, 32 * tt + lc, 0, half);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if (ks + 1 < 8) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) fr[(ks + 1) & 1][tt] = lds_frag(img, 32 * tt + lc, ks + 1, half);
    }
    // keep the reads above the MFMAs: left alone, the scheduler (register file full) sinks every read to just
    // before its MFMA and reuses one register quad, i.e. waits out the LDS latency 32 times per panel
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      acc[tt] = W_IS_A ? mfma(wp.f[ks], fr[ks & 1][tt], acc[tt]) : mfma(fr[ks & 1][tt], wp.f[ks], acc[tt]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// x + (same lane of the other half-wave)
__device__ __forceinline__ float add_halves(float v) {
  const int vi = __float_as_int(v);
  auto s = __builtin_amdgcn_permlane32_swap(vi, vi, false, false);
  return __int_as_float(s[0]) + __int_as_float(s[1]);
}
__device__ __forceinline__ float max_halves(float v) {
  const int vi = __float_as_int(v);
  auto s = __builtin_amdgcn_permlane32_swap(vi, vi, false, false);
  return fmaxf(__int_as_float(s[0]), __int_as_float(s[1]));
}

// per-layer small vectors staged in LDS: [b_qkv 3d | b_o d | ln1_g d | ln1_b d | b_1 ff | b_2 d | ln2_g d | ln2_b d]
constexpr int FF_MAX = 1024;
constexpr int P_BQKV = 0, P_BO = 3 * D, P_LN1G = 4 * D, P_LN1B = 5 * D, P_B1 = 6 * D;
constexpr int PAR_FLOATS = 9 * D + FF_MAX;
__device__ __forceinline__ int p_b2(int ff) { return 6 * D + ff; }

template <int TT>
struct TfSmem {
  static constexpr int NTOK = 32 * TT;
  alignas(16) char x[NTOK * ROWB];     // bf16 image of the residual stream (MFMA operand)
  alignas(16) char s[NTOK * ROWB];     // attention output / FFN hidden chunk (even) / embedding weight staging
  alignas(16) char s2[NTOK * ROWB];    // FFN hidden chunk (odd): one barrier per chunk instead of two
  float stat[2][4][NTOK];              // per-wave LayerNorm partial sums: [0] sums, [1] squared deviations
  float par[2][PAR_FLOATS];            // bias / LayerNorm vectors of the current and the next layer (one round trip per
                                       // layer instead of ~8 exposed global-load latencies)
};

// post-LayerNorm of z (fp32, transposed tiles of this wave's 32 features) over all 128 features of each token.
// Each wave reduces its 32 features to (sum, squared deviations from its own mean) in registers; the four partial
// results per token are combined after ONE workgroup barrier with the pairwise (Chan et al.) update, which is as
// stable as the two-pass form: M2 = sum_w [ M2_w + 32 (mean_w - mean)^2 ].
template <int TT>
__device__ __forceinline__ void layer_norm(f32x16 (&z)[TT], TfSmem<TT>& sm, const float* __restrict__ g,
                                           const float* __restrict__ bt, int w, int lc, int half) {
  float gv[16], bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    gv[r] = g[32 * w + acc_row(r, half)];
    bv[r] = bt[32 * w + acc_row(r, half)];
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float sacc = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += z[tt][r];
    sacc = add_halves(sacc);
    const float mw = sacc * (1.0f / 32.0f);
    float m2 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dlt = z[tt][r] - mw;
      m2 = fmaf(dlt, dlt, m2);
    }
    m2 = add_halves(m2);
    if (half == 0) {
      sm.stat[0][w][32 * tt + lc] = sacc;
      sm.stat[1][w][32 * tt + lc] = m2;
    }
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = 32 * tt + lc;
    const float s0 = sm.stat[0][0][tok], s1 = sm.stat[0][1][tok], s2 = sm.stat[0][2][tok], s3 = sm.stat[0][3][tok];
    const float mean = ((s0 + s1) + (s2 + s3)) * (1.0f / D);
    float m2 = (sm.stat[1][0][tok] + sm.stat[1][1][tok]) + (sm.stat[1][2][tok] + sm.stat[1][3][tok]);
    const float d0 = s0 * (1.0f / 32.0f) - mean, d1 = s1 * (1.0f / 32.0f) - mean;
    const float d2 = s2 * (1.0f / 32.0f) - mean, d3 = s3 * (1.0f / 32.0f) - mean;
    m2 = fmaf(32.0f, (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3), m2);
    const float rs = rsqrtf(m2 * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int r = 0; r < 16; ++r) z[tt][r] = fmaf((z[tt][r] - mean) * rs, gv[r], bv[r]);
  }
  // stat[] is rewritten only after the caller's next barrier (every call site stores the image and syncs)
}

__device__ __forceinline__ void stage_layer_params(float* dst, const quattro_tf_weights& W, int layer, int tid) {
  const int ff = W.d_ff;
  for (int i = tid; i < 3 * D; i += 256) dst[P_BQKV + i] = W.b_qkv[layer][i];
  for (int i = tid; i < ff; i += 256) dst[P_B1 + i] = W.b_1[layer][i];
  if (tid < D) {
    dst[P_BO + tid] = W.b_o[layer][tid];
    dst[P_LN1G + tid] = W.ln1_g[layer][tid];
    dst[P_LN1B + tid] = W.ln1_b[layer][tid];
    dst[p_b2(ff) + tid] = W.b_2[layer][tid];
    dst[p_b2(ff) + D + tid] = W.ln2_g[layer][tid];
    dst[p_b2(ff) + 2 * D + tid] = W.ln2_b[layer][tid];
  }
}

// Diagnostic build only (-DQT_TF_PROFILE, scripts/tf_profile.sh): s_memtime stamps of wave 0 at phase boundaries go to
// a buffer of their own; the shipped library is built without it and executes no stamp.
#ifdef QT_TF_PROFILE
#define QT_STAMP(i)                                                                     \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (tid == 0) dbg[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime();    \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#define QT_DBG_PARAM , unsigned long long* __restrict__ dbg
#else
#define QT_STAMP(i)
#define QT_DBG_PARAM
#endif

// optional direct output of the prediction into gain stacks K [B][N][m][n], k [B][N][m] (quattro_tf_gains_bf16)
struct TfGainsOut {
  float* K;
  float* k;
  const int32_t* active;
  int N, n, m;
};

template <int TT>
__global__ __launch_bounds__(256, 1) void tf_forward_kernel(const quattro_tf_weights W,
                                                            const float* __restrict__ x_err,
                                                            const float* __restrict__ prompt,
                                                            float* __restrict__ pred, TfGainsOut go QT_DBG_PARAM) {
  __shared__ TfSmem<TT> sm;
  const int b = blockIdx.x;
  if (go.active != nullptr && go.active[b] == 0) return;   // gains mode: converged trajectories keep their gains
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, half = lane >> 5, lc = lane & 31;
  const int NS = W.n_state_tok, P = W.prompt_len, T = W.target_len, L = NS + P + T;
  const int NXI = W.n_x, C = W.c_dim;
  QT_STAMP(0);

  // ------------------------------------------------------------------ embeddings (+ positional / target rows)
  // State tokens: one MFMA k-step per tile.  A = state_embed.weight rows as bf16 [128][16] (zero-padded), B = the
  // normalised state of the lane's own token built in registers (k = 8*half .. 8*half+7), C = bias.  (The fp32 version
  // read its weights back from LDS scalar by scalar, 768 ds_read_b32 per lane.  Doing the prompt rows the same way —
  // four more k-steps on the one tile that holds them, no LDS staging or barrier at all — measured SLOWER, 1.42 vs
  // 1.34 ms: 32 predicated scalar loads per lane in front of the MFMAs.)
  stage_layer_params(sm.par[0], W, 0, tid);
  // control embedding of the P prompt tokens, computed by the whole workgroup into LDS (the prompt rows belong to
  // one or two lanes of the token-per-lane layout; left to them, P*128 dot products of length c run serially)
  float* pn = reinterpret_cast<float*>(sm.s);                 // normalised prompt rows [P][c]
  float* pe_out = pn + P * C;                                 // embedded prompt rows   [P][128]
  float* cw = reinterpret_cast<float*>(sm.x);                 // control_embed.weight [128][c], fp32
  {
    const int n4 = (D * C) / 4;                               // 16-byte loads, several in flight per thread
#pragma unroll 4
    for (int i = tid; i < n4; i += 256)
      reinterpret_cast<float4*>(cw)[i] = reinterpret_cast<const float4*>(W.ctrl_w)[i];
    for (int i = 4 * n4 + tid; i < D * C; i += 256) cw[i] = W.ctrl_w[i];
  }
  for (int i = tid; i < P * C; i += 256) {
    const int k = i % C;
    pn[i] = (prompt[(size_t)b * P * C + i] - W.u_mean[k]) / W.u_std[k];
  }
  f32x16 X[TT];
  {
    const bf16x8 wst = *reinterpret_cast<const bf16x8*>(W.w_state + (size_t)(32 * w + lc) * 16 + 8 * half);
    float sb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sb[r] = W.state_b[32 * w + acc_row(r, half)];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int tok = 32 * tt + lc;
      bf16x8 xf;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * half + j;
        float v = 0.0f;
        if (tok < NS && k < NXI) v = (x_err[((size_t)b * NS + tok) * NXI + k] - W.x_mean[k]) / W.x_std[k];
        xf[j] = (__bf16)v;
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = sb[r];
      X[tt] = mfma(wst, xf, acc);                             // rows = this wave's 32 features, cols = tokens of tile tt
    }
  }
  __syncthreads();
  for (int o = tid; o < P * D; o += 256) {
    const int pi = o / D, f = o % D;
    float acc = W.ctrl_b[f];
    float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;                // four independent chains: the LDS reads of a chain overlap
    int k = 0;
    for (; k + 3 < C; k += 4) {
      acc = fmaf(pn[pi * C + k], cw[f * C + k], acc);
      a1 = fmaf(pn[pi * C + k + 1], cw[f * C + k + 1], a1);
      a2 = fmaf(pn[pi * C + k + 2], cw[f * C + k + 2], a2);
      a3 = fmaf(pn[pi * C + k + 3], cw[f * C + k + 3], a3);
    }
    for (; k < C; ++k) acc = fmaf(pn[pi * C + k], cw[f * C + k], acc);
    pe_out[o] = (acc + a1) + (a2 + a3);
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = 32 * tt + lc;                             // this lane's token (accumulator column)
    if (tok >= NS) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        X[tt][r] = (tok < NS + P) ? pe_out[(tok - NS) * D + 32 * w + acc_row(r, half)] : 0.0f;
    }
    if (tok < L) {
#pragma unroll
      for (int r = 0; r < 16; ++r) X[tt][r] += W.tok_bias[(size_t)tok * D + 32 * w + acc_row(r, half)];
    }
  }
  __syncthreads();   // staging area sm.s is free again
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) lds_store_tile(sm.x, X[tt], tt, w, lc, half);
  __syncthreads();

  QT_STAMP(1);
  const float qscale = 0.17677669529663687f;   // 1/sqrt(32)

  // Weights are requested a whole phase ahead (8 fragments = one 128-deep K panel of this wave's 32 rows): at one
  // wave per SIMD nothing else hides the L2 latency, and the compiler on its own fetches each fragment just in time.
  W8 wq = load_w8(W.w_qkv[0], D, 0 * D + HD * w + lc, 0, half);

  for (int layer = 0; layer < W.n_layers; ++layer) {
    const float* par = sm.par[layer & 1];
    const float* bqkv = par + P_BQKV;
    // -------------------------------------------------------------- Q^T, K^T (hd x tokens), V (tokens x hd) of head w
    // (one projection at a time: its 4 accumulator tiles are packed to bf16 operands before the next one starts;
    //  the K and V panels are requested here and land behind the Q products)
    bf16x8 Qp[TT][2], Kp[TT][2], Vp[TT][2];
    const W8 wk = load_w8(W.w_qkv[layer], D, 1 * D + HD * w + lc, 0, half);
    const W8 wv = load_w8(W.w_qkv[layer], D, 2 * D + HD * w + lc, 0, half);
    {
      f32x16 acc[TT];
      float bias[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = bqkv[0 * D + HD * w + acc_row(r, half)];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = zero16();
      gemm_panel<TT, true>(acc, wq, sm.x, lc, half);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = (acc[tt][r] + bias[r]) * qscale;
        Qp[tt][0] = pack8<0>(acc[tt]); Qp[tt][1] = pack8<1>(acc[tt]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) bias[r] = bqkv[1 * D + HD * w + acc_row(r, half)];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = zero16();
      gemm_panel<TT, true>(acc, wk, sm.x, lc, half);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] += bias[r];
        Kp[tt][0] = pack8<0>(acc[tt]); Kp[tt][1] = pack8<1>(acc[tt]);
      }
      const float bvl = bqkv[2 * D + HD * w + lc];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[tt] = zero16();
      gemm_panel<TT, false>(acc, wv, sm.x, lc, half);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] += bvl;
        Vp[tt][0] = pack8<0>(acc[tt]); Vp[tt][1] = pack8<1>(acc[tt]);
      }
    }
    QT_STAMP(2 + 6 * layer + 0);
    const W8 wo = load_w8(W.w_o[layer], D, 32 * w + lc, 0, half);          // lands during the attention
    // -------------------------------------------------------------- causal attention of head w, one query tile at a time
#pragma unroll
    for (int qt = 0; qt < TT; ++qt) {
      f32x16 S[TT];
      float m = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt <= qt; ++kt) {
        S[kt] = mfma(Kp[kt][0], Qp[qt][0], zero16());
        S[kt] = mfma(Kp[kt][1], Qp[qt][1], S[kt]);          // S^T tile: rows keys, cols queries
        if (kt == qt) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (acc_row(r, half) > lc) S[kt][r] = -3.0e38f;    // key index > query index: masked
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, S[kt][r]);
      }
      m = max_halves(m);
      float lsum = 0.0f;
      f32x16 O = zero16();
#pragma unroll
      for (int kt = 0; kt <= qt; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __expf(S[kt][r] - m);
          S[kt][r] = pv;
          lsum += pv;
        }
        O = mfma(Vp[kt][0], pack8<0>(S[kt]), O);             // O^T tile: rows head features, cols queries
        O = mfma(Vp[kt][1], pack8<1>(S[kt]), O);
      }
      lsum = add_halves(lsum);
      const float inv = 1.0f / lsum;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] *= inv;
      lds_store_tile(sm.s, O, qt, w, lc, half);
    }
    QT_STAMP(2 + 6 * layer + 1);
    const uint16_t* W1 = W.w_1[layer];
    const uint16_t* W2 = W.w_2[layer];
    const int FF = W.d_ff;
    W8 w1 = load_w8(W1, D, 32 * w + lc, 0, half);                          // first FFN chunk: lands during out-proj / LN1
    __syncthreads();
    // -------------------------------------------------------------- output projection + residual + LayerNorm 1
    {
      f32x16 Y[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) Y[tt] = zero16();
      gemm_panel<TT, true>(Y, wo, sm.s, lc, half);
      const float* bo = par + P_BO;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bb = bo[32 * w + acc_row(r, half)];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) X[tt][r] += Y[tt][r] + bb;
      }
    }
    QT_STAMP(2 + 6 * layer + 2);
    W8 w2 = load_w8(W2, FF, 32 * w + lc, 0, half);
    layer_norm<TT>(X, sm, par + P_LN1G, par + P_LN1B, w, lc, half);
    if (layer + 1 < W.n_layers) stage_layer_params(sm.par[(layer + 1) & 1], W, layer + 1, tid);   // visible after the FFN barriers
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) lds_store_tile(sm.x, X[tt], tt, w, lc, half);
    __syncthreads();
    QT_STAMP(2 + 6 * layer + 3);
    // -------------------------------------------------------------- feed-forward in hidden chunks of 128
    {
      const float* b1 = par + P_B1;
      f32x16 Y2[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) Y2[tt] = zero16();
      for (int c0 = 0; c0 < FF; c0 += 128) {
        const bool more = c0 + 128 < FF;
        f32x16 H[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) H[tt] = zero16();
        gemm_panel<TT, true>(H, w1, sm.x, lc, half);
        if (more) {
          w1 = load_w8(W1, D, c0 + 128 + 32 * w + lc, 0, half);           // next chunk's W1 panel
        } else if (layer + 1 < W.n_layers) {                              // next layer's Q panel
          wq = load_w8(W.w_qkv[layer + 1], D, 0 * D + HD * w + lc, 0, half);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float bb = b1[c0 + 32 * w + acc_row(r, half)];
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) H[tt][r] = fmaxf(H[tt][r] + bb, 0.0f);
        }
        char* hbuf = ((c0 >> 7) & 1) ? sm.s2 : sm.s;      // alternate: the other image may still be read by slower waves
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) lds_store_tile(hbuf, H[tt], tt, w, lc, half);
        __syncthreads();
        gemm_panel<TT, true>(Y2, w2, hbuf, lc, half);
        if (more) w2 = load_w8(W2, FF, 32 * w + lc, (c0 + 128) / 16, half);
      }
      const float* b2 = par + p_b2(FF);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bb = b2[32 * w + acc_row(r, half)];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) X[tt][r] += Y2[tt][r] + bb;
      }
    }
    QT_STAMP(2 + 6 * layer + 4);
    layer_norm<TT>(X, sm, par + p_b2(FF) + D, par + p_b2(FF) + 2 * D, w, lc, half);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) lds_store_tile(sm.x, X[tt], tt, w, lc, half);
    __syncthreads();
    QT_STAMP(2 + 6 * layer + 5);
  }

  // ------------------------------------------------------------------ output head on the last T tokens, de-normalised
  if (w < TT) {
    const int tok = 32 * w + lc;
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      if (32 * ft >= C) break;
      f32x16 acc = zero16();
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks)
        acc = mfma(gw_frag(W.w_out, D, 32 * ft + lc, ks, half), lds_frag(sm.x, tok, ks, half), acc);
      if (tok >= L - T && tok < L) {
        const int t = tok - (L - T);
        if (go.K == nullptr) {
          float* dst = pred + ((size_t)b * T + t) * C;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = 32 * ft + acc_row(r, half);
            if (o < C) dst[o] = fmaf(acc[r] + W.b_out[o], W.u_std[o], W.u_mean[o]);
          }
        } else if (t < go.N) {
          // gains mode: row t of the prediction viewed as (m, 1 + n) — column 0 is k_t, the rest K_t
          // (quattro_ilqr_tf.py:510-514) — written straight into the solver's gain stacks; rows >= N of an
          // over-long prediction are dropped like the reference's forward_pass never reads them
          const int n1 = go.n + 1;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = 32 * ft + acc_row(r, half);
            if (o < C) {
              const float v = fmaf(acc[r] + W.b_out[o], W.u_std[o], W.u_mean[o]);
              const int i = o / n1, j = o - i * n1;
              if (j == 0) go.k[((size_t)b * go.N + t) * go.m + i] = v;
              else go.K[(((size_t)b * go.N + t) * go.m + i) * go.n + (j - 1)] = v;
            }
          }
        }
      }
    }
  }
  QT_STAMP(62);
}

}  // namespace

#ifdef QT_TF_PROFILE
#define QT_DBG_ARG , dbg
extern "C" int quattro_tf_forward_profile(const quattro_tf_weights* Wp, const float* x_err, const float* prompt, int B,
                                          float* pred, unsigned long long* dbg, void* stream_) {
  const quattro_tf_weights& W = *Wp;
  hipStream_t stream = (hipStream_t)stream_;
#else
#define QT_DBG_ARG
int quattro_launch_tf_forward(const quattro_tf_weights& W, const float* x_err, const float* prompt, int B, float* pred,
                              float* Kout, float* kout, const int32_t* active, int N, int n, int m, hipStream_t stream) {
#endif
#ifdef QT_TF_PROFILE
  const TfGainsOut go{nullptr, nullptr, nullptr, 0, 0, 0};
#else
  const TfGainsOut go{Kout, kout, active, N, n, m};
#endif
  const int L = W.n_state_tok + W.prompt_len + W.target_len;
  if (W.d_model != D || W.n_head != 4 || W.d_ff <= 0 || W.d_ff % 128 != 0 || W.d_ff > FF_MAX || W.c_dim <= 0 || W.c_dim > 64 ||
      W.n_x <= 0 || W.n_x > QUATTRO_MAX_NX || L > 128 || W.n_layers <= 0 || W.n_layers > QUATTRO_TF_MAX_LAYERS ||
      W.n_state_tok <= 0 || W.prompt_len <= 0 || W.target_len <= 0)
    return QUATTRO_ERR_UNSUPPORTED;
  // embedding staging must fit the two LDS images it borrows
  const size_t img = (size_t)(L <= 64 ? 64 : 128) * ROWB;
  if ((size_t)(W.prompt_len * W.c_dim + W.prompt_len * D) * sizeof(float) > img ||
      (size_t)(D * W.c_dim) * sizeof(float) > img)
    return QUATTRO_ERR_UNSUPPORTED;
  if (L <= 64) {
    hipLaunchKernelGGL((tf_forward_kernel<2>), dim3(B), dim3(256), 0, stream, W, x_err, prompt, pred, go QT_DBG_ARG);
  } else {
    hipLaunchKernelGGL((tf_forward_kernel<4>), dim3(B), dim3(256), 0, stream, W, x_err, prompt, pred, go QT_DBG_ARG);
  }
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
