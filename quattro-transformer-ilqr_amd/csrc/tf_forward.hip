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

template <int TT>
struct TfSmem {
  static constexpr int NTOK = 32 * TT;
  alignas(16) char x[NTOK * ROWB];     // bf16 image of the residual stream (MFMA operand)
  alignas(16) char s[NTOK * ROWB];     // attention output / FFN hidden chunk / embedding weight staging
  float stat[4][NTOK];                 // per-wave LayerNorm partial sums
};

// post-LayerNorm of z (fp32, transposed tiles of this wave's 32 features) over all 128 features of each token
template <int TT>
__device__ __forceinline__ void layer_norm(f32x16 (&z)[TT], TfSmem<TT>& sm, const float* __restrict__ g,
                                           const float* __restrict__ bt, int w, int lc, int half) {
  float gv[16], bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    gv[r] = g[32 * w + acc_row(r, half)];
    bv[r] = bt[32 * w + acc_row(r, half)];
  }
  float mean[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float sacc = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc += z[tt][r];
    sacc = add_halves(sacc);
    if (half == 0) sm.stat[w][32 * tt + lc] = sacc;
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = 32 * tt + lc;
    mean[tt] = (sm.stat[0][tok] + sm.stat[1][tok] + sm.stat[2][tok] + sm.stat[3][tok]) * (1.0f / D);
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    float sacc = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dlt = z[tt][r] - mean[tt];
      sacc = fmaf(dlt, dlt, sacc);
    }
    sacc = add_halves(sacc);
    if (half == 0) sm.stat[w][32 * tt + lc] = sacc;
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = 32 * tt + lc;
    const float var = (sm.stat[0][tok] + sm.stat[1][tok] + sm.stat[2][tok] + sm.stat[3][tok]) * (1.0f / D);
    const float rs = rsqrtf(var + 1e-5f);
#pragma unroll
    for (int r = 0; r < 16; ++r) z[tt][r] = fmaf((z[tt][r] - mean[tt]) * rs, gv[r], bv[r]);
  }
  __syncthreads();   // stat[] may be rewritten by the next LayerNorm
}

template <int TT>
__global__ __launch_bounds__(256, 1) void tf_forward_kernel(const quattro_tf_weights W,
                                                            const float* __restrict__ x_err,
                                                            const float* __restrict__ prompt,
                                                            float* __restrict__ pred) {
  __shared__ TfSmem<TT> sm;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, half = lane >> 5, lc = lane & 31;
  const int NS = W.n_state_tok, P = W.prompt_len, T = W.target_len, L = NS + P + T;
  const int NXI = W.n_x, C = W.c_dim;

  // ------------------------------------------------------------------ embeddings (+ positional / target rows)
  float* sw = reinterpret_cast<float*>(sm.s);                 // state_w [128][n_x] then state_b [128], fp32
  for (int i = tid; i < D * NXI; i += 256) sw[i] = W.state_w[i];
  for (int i = tid; i < D; i += 256) sw[D * NXI + i] = W.state_b[i];
  __syncthreads();
  f32x16 X[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int tok = 32 * tt + lc;
    X[tt] = zero16();
    if (tok < NS) {
      float xn[QUATTRO_MAX_NX];
      for (int k = 0; k < NXI; ++k)
        xn[k] = (x_err[((size_t)b * NS + tok) * NXI + k] - W.x_mean[k]) / W.x_std[k];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 32 * w + acc_row(r, half);
        float acc = sw[D * NXI + f];
        for (int k = 0; k < NXI; ++k) acc = fmaf(xn[k], sw[f * NXI + k], acc);
        X[tt][r] = acc;
      }
    } else if (tok < NS + P) {
      const float* pr = prompt + ((size_t)b * P + (tok - NS)) * C;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = 32 * w + acc_row(r, half);
        float acc = W.ctrl_b[f];
        for (int k = 0; k < C; ++k) acc = fmaf((pr[k] - W.u_mean[k]) / W.u_std[k], W.ctrl_w[(size_t)f * C + k], acc);
        X[tt][r] = acc;
      }
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

  const float qscale = 0.17677669529663687f;   // 1/sqrt(32)

  for (int layer = 0; layer < W.n_layers; ++layer) {
    const uint16_t* Wqkv = W.w_qkv[layer];
    const float* bqkv = W.b_qkv[layer];
    // -------------------------------------------------------------- Q^T, K^T (hd x tokens), V (tokens x hd) of head w
    bf16x8 Qp[TT][2], Kp[TT][2], Vp[TT][2];
    {
      f32x16 aQ[TT], aK[TT], aV[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) { aQ[tt] = zero16(); aK[tt] = zero16(); aV[tt] = zero16(); }
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const bf16x8 wq = gw_frag(Wqkv, D, 0 * D + HD * w + lc, ks, half);
        const bf16x8 wk = gw_frag(Wqkv, D, 1 * D + HD * w + lc, ks, half);
        const bf16x8 wv = gw_frag(Wqkv, D, 2 * D + HD * w + lc, ks, half);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const bf16x8 xf = lds_frag(sm.x, 32 * tt + lc, ks, half);
          aQ[tt] = mfma(wq, xf, aQ[tt]);
          aK[tt] = mfma(wk, xf, aK[tt]);
          aV[tt] = mfma(xf, wv, aV[tt]);
        }
      }
      float bq[16], bk[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        bq[r] = bqkv[0 * D + HD * w + acc_row(r, half)];
        bk[r] = bqkv[1 * D + HD * w + acc_row(r, half)];
      }
      const float bvl = bqkv[2 * D + HD * w + lc];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          aQ[tt][r] = (aQ[tt][r] + bq[r]) * qscale;
          aK[tt][r] += bk[r];
          aV[tt][r] += bvl;
        }
        Qp[tt][0] = pack8<0>(aQ[tt]); Qp[tt][1] = pack8<1>(aQ[tt]);
        Kp[tt][0] = pack8<0>(aK[tt]); Kp[tt][1] = pack8<1>(aK[tt]);
        Vp[tt][0] = pack8<0>(aV[tt]); Vp[tt][1] = pack8<1>(aV[tt]);
      }
    }
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
    __syncthreads();
    // -------------------------------------------------------------- output projection + residual + LayerNorm 1
    {
      const uint16_t* Wo = W.w_o[layer];
      f32x16 Y[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) Y[tt] = zero16();
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const bf16x8 wo = gw_frag(Wo, D, 32 * w + lc, ks, half);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) Y[tt] = mfma(wo, lds_frag(sm.s, 32 * tt + lc, ks, half), Y[tt]);
      }
      const float* bo = W.b_o[layer];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bb = bo[32 * w + acc_row(r, half)];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) X[tt][r] += Y[tt][r] + bb;
      }
    }
    layer_norm<TT>(X, sm, W.ln1_g[layer], W.ln1_b[layer], w, lc, half);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) lds_store_tile(sm.x, X[tt], tt, w, lc, half);
    __syncthreads();
    // -------------------------------------------------------------- feed-forward in hidden chunks of 128
    {
      const uint16_t* W1 = W.w_1[layer];
      const uint16_t* W2 = W.w_2[layer];
      const float* b1 = W.b_1[layer];
      const int FF = W.d_ff;
      f32x16 Y2[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) Y2[tt] = zero16();
      for (int c0 = 0; c0 < FF; c0 += 128) {
        f32x16 H[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) H[tt] = zero16();
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
          const bf16x8 w1 = gw_frag(W1, D, c0 + 32 * w + lc, ks, half);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) H[tt] = mfma(w1, lds_frag(sm.x, 32 * tt + lc, ks, half), H[tt]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float bb = b1[c0 + 32 * w + acc_row(r, half)];
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) H[tt][r] = fmaxf(H[tt][r] + bb, 0.0f);
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) lds_store_tile(sm.s, H[tt], tt, w, lc, half);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 128 / 16; ++ks) {
          const bf16x8 w2 = gw_frag(W2, FF, 32 * w + lc, c0 / 16 + ks, half);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) Y2[tt] = mfma(w2, lds_frag(sm.s, 32 * tt + lc, ks, half), Y2[tt]);
        }
        __syncthreads();
      }
      const float* b2 = W.b_2[layer];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float bb = b2[32 * w + acc_row(r, half)];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) X[tt][r] += Y2[tt][r] + bb;
      }
    }
    layer_norm<TT>(X, sm, W.ln2_g[layer], W.ln2_b[layer], w, lc, half);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) lds_store_tile(sm.x, X[tt], tt, w, lc, half);
    __syncthreads();
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
        float* dst = pred + ((size_t)b * T + (tok - (L - T))) * C;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * ft + acc_row(r, half);
          if (o < C) dst[o] = fmaf(acc[r] + W.b_out[o], W.u_std[o], W.u_mean[o]);
        }
      }
    }
  }
}

}  // namespace

int quattro_launch_tf_forward(const quattro_tf_weights& W, const float* x_err, const float* prompt, int B, float* pred,
                              hipStream_t stream) {
  const int L = W.n_state_tok + W.prompt_len + W.target_len;
  if (W.d_model != D || W.n_head != 4 || W.d_ff <= 0 || W.d_ff % 128 != 0 || W.c_dim <= 0 || W.c_dim > 64 ||
      W.n_x <= 0 || W.n_x > QUATTRO_MAX_NX || L > 128 || W.n_layers <= 0 || W.n_layers > QUATTRO_TF_MAX_LAYERS ||
      W.n_state_tok <= 0 || W.prompt_len <= 0 || W.target_len <= 0)
    return QUATTRO_ERR_UNSUPPORTED;
  if ((size_t)(D * W.n_x + D) * sizeof(float) > (size_t)(L <= 64 ? 64 : 128) * ROWB) return QUATTRO_ERR_UNSUPPORTED;
  if (L <= 64) {
    hipLaunchKernelGGL((tf_forward_kernel<2>), dim3(B), dim3(256), 0, stream, W, x_err, prompt, pred);
  } else {
    hipLaunchKernelGGL((tf_forward_kernel<4>), dim3(B), dim3(256), 0, stream, W, x_err, prompt, pred);
  }
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
