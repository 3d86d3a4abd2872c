// Transformer gain predictor, fused forward for a batch of sequences, bf16 MFMA with fp32 accumulation.
//
// Arithmetic replaced (reference): TransformerPredictor.forward quattro_ilqr_tf/transformer_model.py:122-138
// (embeddings :125-131, PositionalEncoding :77-80, 3 x nn.TransformerEncoderLayer as configured at :106-113:
// batch_first, post-LayerNorm, ReLU, eps 1e-5, causal mask :135, output_linear :138) and the normalise /
// de-normalise steps of TransformerILQR.predict quattro_ilqr_tf/transformer_ilqr.py:312-324.
//
// MI355X design ("token-owning waves, streamed weights").  One workgroup runs the WHOLE forward of one sequence;
// wave w owns the 32 tokens [32w, 32w+32) for ALL 128 features, two workgroups share a CU (two waves per SIMD,
// <= 256 registers, <= 80 KB of LDS each) so one workgroup's VALU stretches (softmax, LayerNorm, packing) run under
// the other's MFMAs.
//   * Every activation of a wave stays in its registers for the whole forward.  All products are computed
//     transposed (features x tokens) with v_mfma_f32_32x32x16_bf16, so a result tile has one token per lane and its
//     features in the 16 accumulator registers — which is (a) the B operand of the next product with no lane
//     movement (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand"; the k order inside a
//     16-deep step is permuted, and the weight stream is stored with the same permutation), and (b) the layout
//     LayerNorm wants: the statistics of a token are a sum over registers plus one v_permlane32_swap — no
//     cross-wave reduction, no barrier.  Residual adds are free: the out-projection and the second FFN product
//     accumulate straight into the fp32 residual tiles.
//   * The only cross-wave data are K and V of the attention (causal: wave w reads the key tiles <= w).  They are
//     exchanged through LDS in MFMA-fragment order (16 B per lane, conflict-free), one head at a time.
//   * Weights are the A operand and every wave needs all of them: they are streamed ONCE per workgroup from L2 into a
//     4-slot LDS ring by LDS-DMA (global_load_lds_dwordx4, no registers, three 8-KB panels in flight behind a counted
//     s_waitcnt vmcnt), in exactly the order they are consumed — the host packs each layer's matrices into that
//     order as 1-KB MFMA fragments (quattro_tf_pack_stream_bf16).  Per MFMA a wave reads ONE 1-KB fragment from
//     LDS; no activation ever goes through LDS.  Biases and LayerNorm vectors of a layer are one more panel of the
//     same stream, copied from its ring slot into a double-buffered LDS block at the layer boundary; they enter
//     as accumulator initial values.
//   * Work nobody reads is not done: attention visits only the key tiles at or below the diagonal; K needs no bias
//     (a per-query constant in the scores cancels in the softmax), and V's bias, which the softmax weights sum to
//     itself, is folded through the out-projection into its bias when the parameters are packed
//     (b_o + W_o b_v).  The out-projection is applied head by head (its k-steps are the heads), so a head's
//     attention output is consumed at once.
#include "quattro_device.h"

#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// 16-bit operand type of the MFMAs: __bf16 (what the north star names) or _Float16 (what the shipped checkpoints and the
// reference's own predict() use, transformer_ilqr.py:317-319: weights exact, 3 more mantissa bits on the activations)
template <class E> struct Vec8;
template <> struct Vec8<__bf16> { typedef bf16x8 type; };
template <> struct Vec8<_Float16> { typedef f16x8 type; };
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int D = 128;           // d_model
constexpr int FRAG_B = 1024;     // one MFMA A/B fragment of a wave: 64 lanes x 16 B
constexpr int FRAG_E = 512;      // ... in bf16 elements
constexpr int PANEL_B = 8 * FRAG_B;
constexpr int RING = 4;          // LDS ring slots (panels); RING - 1 panels are in flight behind the one being read
constexpr int EMB_FRAGS = 20;    // embedding weights: 4 feature tiles x (1 state k-step + 4 control k-steps)

// parameter block of a layer (fp32): offsets in floats; the final block holds [b_out 64 | u_std 64 | u_mean 64]
constexpr int P_BQ = 0, P_BV = 128, P_BO = 256, P_LN1G = 384, P_LN1B = 512, P_B2 = 640, P_LN2G = 768, P_LN2B = 896,
              P_B1 = 1024;

__host__ __device__ constexpr int tf_panels_per_layer(int ff) { return 17 + 2 * (ff / 32); }   // parameters, 4 x (Q K V O), W1/W2 chunks
__host__ __device__ constexpr int tf_out_panels(int c) { return 1 + (c + 31) / 32; }                   // parameters, output rows
__host__ __device__ constexpr int tf_pstride(int ff) { return 1024 + ff; }

// row of a 32x32 accumulator held by register `reg` of lane half `half`
__device__ __forceinline__ constexpr int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
// k offset (0..15) inside a 16-deep step that element j of lane half `half` of an accumulator-derived operand holds
__host__ __device__ constexpr int perm_k(int j, int half) { return 8 * (j >> 2) + 4 * half + (j & 3); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}
// accumulator registers 8s..8s+7 as the 8 k-elements of the next product's operand
template <int S, class E>
__device__ __forceinline__ typename Vec8<E>::type pack8(const f32x16& a) {
  typename Vec8<E>::type p;
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = (E)a[8 * S + j];
  return p;
}
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

// LDS byte address of a __shared__ object
template <class T>
__device__ __forceinline__ uint32_t lds_addr(T* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)p;
}

// One LDS-DMA copy of a wave: 64 lanes x 16 B from (wave-uniform base in SGPRs + 32-bit lane offset) to LDS address
// `lds_dst` + 16 * lane.  Inline asm rather than __builtin_amdgcn_global_load_lds: (1) the SGPR-base form costs one
// shared offset VGPR instead of a 64-bit address pair per copy, and (2) the compiler then sees no LDS-DMA at all — with
// the builtin it orders every LDS read of an object that may alias a copy in flight behind s_waitcnt vmcnt(0), which
// would drain the three panels kept in flight.  Ordering is ours: counted vmcnt waits + barriers in ring_rendezvous,
// whose "memory" clobber also keeps the compiler from moving or caching ring reads across it.
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  uint32_t keep;
  // (s_nop 4 first: an SGPR operand the compiler has just produced with v_readfirstlane needs 5 wait states before a
  //  vector-memory instruction reads it as its base, and nothing inside an asm string is padded for us)
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// own copies landed (all but the N youngest LDS-DMA operations of this wave), own LDS reads retired, workgroup barrier
template <int N>
__device__ __forceinline__ void ring_rendezvous() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// max(x, 0) as ONE instruction: v_max_i32 on the bit pattern (a non-negative float is a non-negative integer, anything with
// the sign bit set becomes +0).  fmaxf on an MFMA result costs a canonicalising v_max first; an inline-asm v_max_f32 is not
// an option: the wait states between an MFMA and a reader inside an asm string are not padded.
__device__ __forceinline__ float relu1(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b > 0 ? b : 0);
}

// Diagnostic build only (-DQT_TF_PROFILE, scripts/tf_profile.sh): every wave accumulates s_memtime deltas per phase
// (and, separately, the cycles spent inside the ring rendezvous: counted wait + barrier) into a buffer of its own; the
// shipped library is built without it and executes no stamp.
#ifdef QT_TF_PROFILE
#define QT_DBG_PARAM , unsigned long long* __restrict__ dbg
#define QT_PH(i)                                               \
  do {                                                         \
    __builtin_amdgcn_sched_barrier(0);                         \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    ph[i] += now_ - last_;                                     \
    last_ = now_;                                              \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)
#else
#define QT_DBG_PARAM
#define QT_PH(i)
#endif

// optional direct output of the prediction into gain stacks K [B][N][m][n], k [B][N][m] (quattro_tf_gains_bf16)
struct TfGainsOut {
  float* K;
  float* k;
  const int32_t* active;
  int N, n, m;
};

template <int NW, int FFMAX>
struct StreamCfg {
  static constexpr int C = (8 + NW - 1) / NW;                     // LDS-DMA instructions per wave per panel
  static constexpr int PSTRIDE_MAX = 1024 + FFMAX;
  static constexpr int XCH_B = 2 * NW * 4 * FRAG_B;               // [head parity][key tile][K0 K1 V0 V1]
};

template <int NW, int FFMAX, class E>
__global__ __launch_bounds__(64 * NW, 2) void tf_stream_kernel(const quattro_tf_weights W,
                                                                const float* __restrict__ x_err,
                                                                const float* __restrict__ prompt,
                                                                float* __restrict__ pred, TfGainsOut go QT_DBG_PARAM) {
  using Cfg = StreamCfg<NW, FFMAX>;
  using ex8 = typename Vec8<E>::type;
  constexpr int C = Cfg::C;
  __shared__ __attribute__((aligned(16))) char s_ring[RING * PANEL_B];
  __shared__ __attribute__((aligned(16))) char s_xch[Cfg::XCH_B];
  __shared__ __attribute__((aligned(16))) float s_par[2 * Cfg::PSTRIDE_MAX];

  const int b = blockIdx.x;
  if (go.active != nullptr && go.active[b] == 0) return;   // gains mode: converged trajectories keep their gains
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, lc = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NS = W.n_state_tok, P = W.prompt_len, T = W.target_len, L = NS + P + T;
  const int NXI = W.n_x, CD = W.c_dim, FF = W.d_ff;
  const int tok = 32 * w + lc;                              // this lane's token (accumulator column)
  const int n_panels = W.n_layers * tf_panels_per_layer(FF) + tf_out_panels(CD);
  const int pstride = tf_pstride(FF);

#ifdef QT_TF_PROFILE
  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime(), rdv_ = 0;
  const unsigned long long t_begin_ = last_;
#endif
  const char* gw = reinterpret_cast<const char*>(W.w_stream) + (size_t)EMB_FRAGS * FRAG_B;   // panel 0
  const uint32_t ring_lds = lds_addr(s_ring);                // LDS byte address of the ring (LDS-DMA destination)
  const char* ring0 = s_ring + lane * 16;                    // this lane's 16 bytes of fragment 0 of slot 0

  // ------------------------------------------------------------------ weight / parameter streaming
  // panel q -> ring slot q % RING; this wave issues C of the panel's 8 fragment copies (1 KB each, lane-linear on both
  // sides).  Past the end of the stream the last panel is copied again: the vmcnt arithmetic below relies on every wave
  // issuing exactly C copies per step, always.
  auto ring_issue = [&](unsigned q) {
    const unsigned qs = q < (unsigned)n_panels ? q : (unsigned)n_panels - 1;
    const char* src = gw + (size_t)qs * PANEL_B;             // wave-uniform
    const uint32_t dst = ring_lds + (q & (RING - 1)) * PANEL_B;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const unsigned f = (w * C + i) < 8 ? (w * C + i) : 7;
      glds16(lane * 16u, src + f * FRAG_B, dst + f * FRAG_B);
    }
  };
  // ------------------------------------------------------------------ embeddings (+ positional / target rows)
  // X^T tiles: XT[ft] rows = features 32 ft + acc_row(reg, half), columns = this wave's tokens.  Initial value: the
  // token-bias table (positional encoding + the embedding bias of the token's kind or its target embedding, folded
  // by the host, stored transposed [128][128] so a half-wave reads 128 contiguous bytes).  State and prompt tokens
  // add W_embed x (normalised input) as MFMA k-steps; the fp32 input is split hi + lo into two bf16 operands, so
  // only the weights are rounded to bf16.  Every ordinary global load of the kernel is requested HERE, ahead of the
  // first LDS-DMA copy: with copies in flight the compiler retires ordinary loads one at a time.
  f32x16 XT[4];
  const bool is_state = tok < NS, is_prompt = tok >= NS && tok < NS + P;
  const float* emb = W.p_stream;                             // [x_istd 16 | u_mean 64 | u_istd 64]
  float xin[8], pin[4][8];
  {
    const float* tb = W.tok_bias_t + tok;                    // zero-padded to 128 tokens: no bounds to check
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) XT[ft][r] = tb[(32 * ft + acc_row(r, half)) * 128];
    // unconditional loads at clamped indices (a conditional load is a branch and a wait of its own), selected afterwards;
    // the inverse standard deviations are zero past the last input dimension
    const float* xr = x_err + ((size_t)b * NS + (tok < NS ? tok : NS - 1)) * NXI;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * half + j, kc = k < NXI ? k : NXI - 1;
      xin[j] = (xr[kc] - W.x_mean[kc]) * emb[k];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) xin[j] = is_state ? xin[j] : 0.0f;
    if (__any(is_prompt)) {
      const int pt = tok - NS, ptc = pt < 0 ? 0 : pt < P ? pt : P - 1;
      // the prompt row [k (m) | K.flat (m n)] (quattro_ilqr_tf.py:498-502): from the caller's array, or — prompt == nullptr,
      // gains mode — straight from rows N - P + pt of the gain stacks, where the tail sweep has just written it
      const float* pr = prompt != nullptr ? prompt + ((size_t)b * P + ptc) * CD : nullptr;
      const size_t grow = (size_t)b * go.N + (go.N - P + ptc);
      const float* gk = go.k + grow * go.m;
      const float* gK = go.K + grow * go.m * go.n;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * s + 8 * half + j, kc = k < CD ? k : CD - 1;
          const float* src = prompt != nullptr ? pr + kc : (kc < go.m ? gk + kc : gK + (kc - go.m));
          pin[s][j] = (*src - emb[16 + k]) * emb[80 + k];
        }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pin[s][j] = is_prompt ? pin[s][j] : 0.0f;
    }
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = 0; q < RING; ++q) ring_issue(q);
  {
    const ex8* wemb = reinterpret_cast<const ex8*>(W.w_stream) + lane;
    auto split = [](const float (&v)[8], ex8& hi, ex8& lo) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        hi[j] = (E)v[j];
        lo[j] = (E)(v[j] - (float)hi[j]);
      }
    };
    if (__any(is_state)) {
      ex8 hi, lo;
      split(xin, hi, lo);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const ex8 a = wemb[(ft * 5 + 0) * 64];
        XT[ft] = mfma(a, hi, XT[ft]);
        XT[ft] = mfma(a, lo, XT[ft]);
      }
    }
    if (__any(is_prompt)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (16 * s < CD) {
          ex8 hi, lo;
          split(pin[s], hi, lo);
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) {
            const ex8 a = wemb[(ft * 5 + 1 + s) * 64];
            XT[ft] = mfma(a, hi, XT[ft]);
            XT[ft] = mfma(a, lo, XT[ft]);
          }
        }
      }
    }
  }

  ex8 Xb[8];                                              // bf16 operand image of XT: k-step 2 ft + s
  auto pack_x = [&]() {
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      Xb[2 * ft + 0] = pack8<0, E>(XT[ft]);
      Xb[2 * ft + 1] = pack8<1, E>(XT[ft]);
    }
  };
  pack_x();

  // ------------------------------------------------------------------ the ring: one step = one 8-fragment panel
  // fa = fragments 0..3 of the current panel (already requested).  A step requests 4..7, runs the first four MFMAs,
  // waits for ITS OWN copies of the next panel (all but the C (RING - 2) youngest), joins the workgroup barrier (every
  // wave's copies of that panel have then landed, and nobody reads the current panel's slot any more), requests
  // fragments 0..3 of the next panel, runs the last four MFMAs and refills the slot just freed.
  QT_PH(0);                                                  // prologue: inputs, first copies issued, embeddings
  unsigned p = 0;                                            // panel being consumed
  ex8 fa[4];
  ring_rendezvous<C * (RING - 1)>();

  auto frag = [&](const char* slot, int f) { return *reinterpret_cast<const ex8*>(slot + f * FRAG_B); };
  // `mid` runs once the first four MFMAs are in the pipe: the place for the PREVIOUS step's dependent epilogue (packing,
  // ReLU, LDS writes), which then executes in their shadow instead of stalling on its own chain's latency.
  auto ring_step = [&](auto&& hook, auto&& mf, auto&& mid, bool prefetch = true) __attribute__((always_inline)) {
    const char* cur = ring0 + (p & (RING - 1)) * PANEL_B;
    const char* nxt = ring0 + ((p + 1) & (RING - 1)) * PANEL_B;
    hook();                                                  // accumulator initial values
    ex8 fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = frag(cur, 4 + i);
    __builtin_amdgcn_sched_barrier(0);   // requested NOW: sunk to just before the rendezvous, their latency is a stall of its own
    static_for<0, 4>([&](auto ic) { mf(ic, fa[decltype(ic)::value]); });
    mid();
#ifdef QT_TF_PROFILE
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long r0_ = __builtin_amdgcn_s_memtime();
#endif
    ring_rendezvous<C * (RING - 2)>();
#ifdef QT_TF_PROFILE
    rdv_ += __builtin_amdgcn_s_memtime() - r0_;
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (prefetch) {     // not ahead of a register-hungry phase (attention, LayerNorm): ring_load_fa() after it instead
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag(nxt, i);
    }
    static_for<0, 4>([&](auto ic) { mf(std::integral_constant<int, decltype(ic)::value + 4>{}, fb[decltype(ic)::value]); });
    ring_issue(p + RING);
    ++p;
  };
  auto no_hook = [] {};
  auto no_mid = [] {};
  auto ring_load_fa_of = [&](unsigned q) __attribute__((always_inline)) {  // fragments 0..3 of a landed panel
    const char* cur = ring0 + (q & (RING - 1)) * PANEL_B;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag(cur, i);
  };
  auto ring_load_fa = [&]() __attribute__((always_inline)) { ring_load_fa_of(p); };
  // A parameter panel (2048 floats: biases and LayerNorm vectors of a layer, or the output head's): copied from its ring
  // slot into parameter buffer `blk & 1` — whose previous readers are a whole layer and many barriers behind — one
  // 1-KB fragment per wave and trip; the rendezvous of the step makes it visible to every wave.
  auto param_step = [&](int blk) __attribute__((always_inline)) {
    const char* cur = ring0 + (p & (RING - 1)) * PANEL_B;
    float* dst = s_par + (blk & 1) * Cfg::PSTRIDE_MAX + lane * 4;
#pragma unroll
    for (int i = 0; i < C; ++i) {                            // fragments w, w + NW, ...
      const int f = w + i * NW;
      if (f < 8 && f * 256 < pstride) *reinterpret_cast<ex8*>(dst + f * 256) = frag(cur, f);
    }
    ring_rendezvous<C * (RING - 2)>();
    ring_load_fa_of(p + 1);
    ring_issue(p + RING);
    ++p;
  };
  // 16 parameters of this lane's accumulator rows (entries 8q + 4 half + 0..3 of a 32-entry group): 4 x ds_read_b128
  auto par_rows = [&](const float* base) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + 8 * q + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[4 * q + j] = v[j];
    }
    return a;
  };
  // post-LayerNorm over the 128 features of each token (lane): registers + the other half-wave, two-pass; the bf16
  // operand image Xb of the normalised tiles is packed before `extra` (the bias of the NEXT residual update: b_2 after
  // LayerNorm 1) is added
  auto layer_norm = [&](const float* g, const float* bt, const float* extra) {
    float s = 0.0f;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += XT[ft][r];
    const float mean = add_halves(s) * (1.0f / D);
    float m2 = 0.0f;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        XT[ft][r] -= mean;                                   // centred IN PLACE: a second copy of the tiles does not fit
        m2 = fmaf(XT[ft][r], XT[ft][r], m2);
      }
    const float rs = rsqrtf(add_halves(m2) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x16 gv = par_rows(g + 32 * ft), bv = par_rows(bt + 32 * ft);
#pragma unroll
      for (int r = 0; r < 16; ++r) XT[ft][r] = fmaf(XT[ft][r] * rs, gv[r], bv[r]);
      Xb[2 * ft + 0] = pack8<0, E>(XT[ft]);
      Xb[2 * ft + 1] = pack8<1, E>(XT[ft]);
      if (extra != nullptr) {
        const f32x16 ev = par_rows(extra + 32 * ft);
#pragma unroll
        for (int r = 0; r < 16; ++r) XT[ft][r] += ev[r];
      }
    }
  };

  const float sc = 0.17677669529663687f * 1.4426950408889634f;   // log2(e) / sqrt(32): softmax in base 2 on raw scores
  char* xch_w = s_xch + w * 4 * FRAG_B + lane * 16;                // this wave's tile of the exchange buffer

  for (int layer = 0; layer < W.n_layers; ++layer) {
    const float* par = s_par + (layer & 1) * Cfg::PSTRIDE_MAX;
    param_step(layer);
    QT_PH(1);
    // Last-layer pruning: an experiment, NOT in the shipped build (-DQT_TF_PRUNE_LAST_LAYER enables it).  Measured
    // on one MI355X, B = 4096, three alternating runs each: 719 / 718 / 723 us with it against 700 / 699 / 700 us without —
    // 2.7 % SLOWER (213 instead of 196 registers, a scalar branch around every MFMA group), results unchanged.  The output head reads only
    // the T target tokens, so in the LAST layer a wave none of whose 32 tokens is a target (wave 0 of the shipped shapes:
    // tokens 0..31 are states) only has to contribute its K and V tiles: its Q projection, attention, out-projection,
    // LayerNorms and feed-forward are skipped.  It still walks every ring step (the weight stream and its barriers are the
    // workgroup's), so the workgroup's own critical path does not move; what is freed is matrix / vector pipe time on that
    // wave's SIMD for the co-resident workgroup.
#ifdef QT_TF_PRUNE_LAST_LAYER
    const bool idle = (layer == W.n_layers - 1) && (32 * (w + 1) <= L - T);
#else
    constexpr bool idle = false;
#endif
    static_for<0, 4>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      char* xw = xch_w + (h & 1) * NW * 4 * FRAG_B;
      ex8 Qp0, Qp1;
      f32x16 qa, ka, va;
      // Q^T of head h (hd x tokens): W_q rows x X^T, bias as initial value
      ring_step([&] { qa = par_rows(par + P_BQ + 32 * h); },
                [&](auto ic, ex8 f) { if (!idle) qa = mfma(f, Xb[decltype(ic)::value], qa); }, no_mid);
      // K^T of head h, no bias; leaves as the A operand of S^T = K Q^T.  (Q is packed under K's first MFMAs.)
      ka = zero16();
      ring_step(no_hook, [&](auto ic, ex8 f) { ka = mfma(f, Xb[decltype(ic)::value], ka); },
                [&] {
                  Qp0 = pack8<0, E>(qa);
                  Qp1 = pack8<1, E>(qa);
                });
      // V of head h (tokens x hd), bias folded into the out-projection's; leaves as the A operand of O^T = V^T P^T
      va = zero16();
      ring_step(no_hook, [&](auto ic, ex8 f) { va = mfma(Xb[decltype(ic)::value], f, va); },
                [&] {
                  *reinterpret_cast<ex8*>(xw + 0 * FRAG_B) = pack8<0, E>(ka);
                  *reinterpret_cast<ex8*>(xw + 1 * FRAG_B) = pack8<1, E>(ka);
                },
                false);
      *reinterpret_cast<ex8*>(xw + 2 * FRAG_B) = pack8<0, E>(va);
      *reinterpret_cast<ex8*>(xw + 3 * FRAG_B) = pack8<1, E>(va);
      QT_PH(2);                                              // Q, K, V steps
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      QT_PH(3);                                              // exchange barrier
      // causal attention of this wave's 32 queries against the key tiles kt <= w, online softmax in base 2.  The scores
      // of tile kt + 1 are requested (two MFMAs) BEFORE the softmax arithmetic of tile kt, which then runs in their shadow.
      f32x16 O = zero16();
      float m = -3.0e38f, l = 0.0f;
      auto xch_tile = [&](int kt) { return s_xch + ((h & 1) * NW + kt) * 4 * FRAG_B + lane * 16; };
      auto scores = [&](int kt) {
        const char* xr = xch_tile(kt);
        const ex8 K0 = *reinterpret_cast<const ex8*>(xr + 0 * FRAG_B);
        const ex8 K1 = *reinterpret_cast<const ex8*>(xr + 1 * FRAG_B);
        f32x16 S = mfma(K0, Qp0, zero16());
        return mfma(K1, Qp1, S);                             // S^T tile: rows keys, columns queries
      };
      f32x16 Sn = zero16();
      if (!idle) Sn = scores(0);
      static_for<0, NW>([&](auto kc) {
        constexpr int kt = decltype(kc)::value;
        if (kt <= w && !idle) {
          f32x16 S = Sn;
          const char* xr = xch_tile(kt);
          const ex8 V0 = *reinterpret_cast<const ex8*>(xr + 2 * FRAG_B);
          const ex8 V1 = *reinterpret_cast<const ex8*>(xr + 3 * FRAG_B);
          if (kt + 1 < NW && kt + 1 <= w) Sn = scores(kt + 1);
          if (kt == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (acc_row(r, half) > lc) S[r] = -3.0e38f;    // key index > query index: masked
          }
          float tm = S[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) tm = fmaxf(tm, S[r]);
          const float mn = fmaxf(m, max_halves(tm));
          const float mc = mn * sc;
          float ls = 0.0f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            S[r] = __builtin_amdgcn_exp2f(fmaf(S[r], sc, -mc));
            ls += S[r];
          }
          if (kt > 0) {
            // rescale what has been accumulated — unless no query of the wave saw a new maximum (exact: corr == 1)
            if (!__all(mn == m)) {
              const float corr = __builtin_amdgcn_exp2f((m - mn) * sc);
              l *= corr;
#pragma unroll
              for (int r = 0; r < 16; ++r) O[r] *= corr;
            }
          }
          l += ls;
          m = mn;
          O = mfma(V0, pack8<0, E>(S), O);                      // O^T tile: rows head features, columns queries
          O = mfma(V1, pack8<1, E>(S), O);
        }
      });
      const float inv = idle ? 0.0f : 1.0f / add_halves(l);
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] *= inv;
      const ex8 Oh0 = pack8<0, E>(O), Oh1 = pack8<1, E>(O);
      QT_PH(4);                                              // attention
      ring_load_fa();
      // this head's two k-steps of the out-projection, accumulated straight into the residual tiles; the step of
      // head h also adds the (folded) out-projection bias of feature tile h
      ring_step(
          [&] {
            const f32x16 bb = par_rows(par + P_BO + 32 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) XT[h][r] += bb[r];
          },
          [&](auto ic, ex8 f) {
            constexpr int i = decltype(ic)::value;
            if (!idle) XT[i >> 1] = mfma(f, (i & 1) ? Oh1 : Oh0, XT[i >> 1]);
          },
          no_mid, h < 3);
      QT_PH(5);                                              // out-projection step
    });
    if (!idle) layer_norm(par + P_LN1G, par + P_LN1B, par + P_B2);      // XT = LN1(..) + b_2, Xb = bf16(LN1(..))
    QT_PH(6);
    ring_load_fa();

    // -------------------------------------------------------------- feed-forward in hidden chunks of 32, LayerNorm 2
    // Stream order W1_0, (W1_1, W2_0), (W1_2, W2_1), ..., W2_last: the ReLU / packing of chunk c runs under the MFMAs
    // of W1_{c+1}, and W2_c follows them into the pipe without waiting for anything.
    {
      f32x16 Ha, Hb;                                         // chunk accumulators, ping-pong
      ex8 H0, H1;
      auto h_pack = [&](const f32x16& Hx) {                  // ReLU -> the two bf16 k-steps of W2's operand
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = relu1(Hx[r]);
        H0 = pack8<0, E>(t);
        H1 = pack8<1, E>(t);
      };
      // W1 step of a chunk into `Hn` (whose initial value, the bias rows, was requested a step earlier: no LDS latency
      // at the head of the MFMA chain), packing the previous chunk `Hp` under its first MFMAs; then W2 of the previous
      // chunk, during which the bias rows of the chunk after `Hn`'s are requested into `Hp`'s registers.
      auto pair = [&](f32x16& Hn, f32x16& Hp, int c_next_bias, bool last) {
        ring_step(no_hook, [&](auto ic, ex8 f) { if (!idle) Hn = mfma(f, Xb[decltype(ic)::value], Hn); },
                  [&] { if (!idle) h_pack(Hp); });
        ring_step(no_hook,
                  [&](auto ic, ex8 f) {
                    constexpr int i = decltype(ic)::value;
                    if (!idle) XT[i >> 1] = mfma(f, (i & 1) ? H1 : H0, XT[i >> 1]);
                  },
                  [&] {
                    if (!last) Hp = par_rows(par + P_B1 + c_next_bias);
                  });
      };
      Ha = par_rows(par + P_B1);
      ring_step(no_hook, [&](auto ic, ex8 f) { if (!idle) Ha = mfma(f, Xb[decltype(ic)::value], Ha); },
                [&] { Hb = par_rows(par + P_B1 + 32); });
      // FF is a multiple of 64: an odd number (FF / 32 - 1) of further chunks; two per trip, the last one peeled
      for (int c0 = 32; c0 + 32 < FF; c0 += 64) {
        pair(Hb, Ha, c0 + 32, false);                        // chunk c0 into Hb; pack chunk c0 - 32 (Ha); Ha <- bias of c0 + 32
        pair(Ha, Hb, c0 + 64, false);                        // chunk c0 + 32 into Ha; pack chunk c0 (Hb); Hb <- bias of c0 + 64
      }
      pair(Hb, Ha, 0, true);                                 // last chunk FF - 32 into Hb; pack chunk FF - 64 (Ha)
      if (!idle) h_pack(Hb);
      ring_step(no_hook,
                [&](auto ic, ex8 f) {
                  constexpr int i = decltype(ic)::value;
                  if (!idle) XT[i >> 1] = mfma(f, (i & 1) ? H1 : H0, XT[i >> 1]);
                },
                no_mid, false);
    }
    QT_PH(7);                                                // feed-forward steps
    if (!idle) layer_norm(par + P_LN2G, par + P_LN2B, nullptr);         // (the parameter step that follows does not use fa)
    QT_PH(8);
  }

  // ------------------------------------------------------------------ output head on the last T tokens, de-normalised
  {
    // (the token index is re-derived behind an opaque asm: the address arithmetic of the ~50 output stores per lane is
    // loop-invariant, and hoisted above the layer loop it would sit in registers the loop needs)
    int tok_o = tok;
    asm volatile("" : "+v"(tok_o));
    const float* par = s_par + (W.n_layers & 1) * Cfg::PSTRIDE_MAX;
    param_step(W.n_layers);
    static_for<0, 2>([&](auto rc) {
      constexpr int rt = decltype(rc)::value;
      if (32 * rt < CD) {
        f32x16 acc;
        ring_step([&] { acc = par_rows(par + 32 * rt); },
                  [&](auto ic, ex8 f) { acc = mfma(f, Xb[decltype(ic)::value], acc); }, no_mid);
        const f32x16 us = par_rows(par + 64 + 32 * rt), um = par_rows(par + 128 + 32 * rt);
        if (tok_o >= L - T && tok_o < L) {
          const int t = tok_o - (L - T);
          if (go.K == nullptr) {
            float* dst = pred + ((size_t)b * T + t) * CD;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int o = 32 * rt + acc_row(r, half);
              if (o < CD) dst[o] = fmaf(acc[r], us[r], um[r]);
            }
          } else if (t < go.N) {
            // gains mode: row t of the prediction viewed as (m, 1 + n) — column 0 is k_t, the rest K_t
            // (quattro_ilqr_tf.py:510-514) — written straight into the solver's gain stacks; rows >= N of an
            // over-long prediction are dropped like the reference's forward_pass never reads them
            const int n1 = go.n + 1;
            const unsigned magic = (65536u + n1 - 1) / n1;      // o / n1 == (o * magic) >> 16 for o < 64 (a true integer
#pragma unroll                                                 // division is ~20 instructions, 52 of them per lane)
            for (int r = 0; r < 16; ++r) {
              const int o = 32 * rt + acc_row(r, half);
              if (o < CD) {
                const float v = fmaf(acc[r], us[r], um[r]);
                const int i = (int)(((unsigned)o * magic) >> 16), j = o - i * n1;
                if (j == 0) go.k[((size_t)b * go.N + t) * go.m + i] = v;
                else go.K[(((size_t)b * go.N + t) * go.m + i) * go.n + (j - 1)] = v;
              }
            }
          }
        }
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // trailing (redundant) ring copies land before the LDS is released
#ifdef QT_TF_PROFILE
  QT_PH(9);                                                  // output head + stores
  if (lane == 0) {
    unsigned long long* d = dbg + ((size_t)blockIdx.x * NW + w) * 16;
    for (int i = 0; i < 10; ++i) d[i] = ph[i];
    d[10] = rdv_;
    d[11] = __builtin_amdgcn_s_memtime() - t_begin_;
    d[12] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ------------------------------------------------------------------------------------------------ stream packing
// One thread per bf16 element of the weight stream / per float of the parameter stream.
__global__ void tf_pack_weights_kernel(const quattro_tf_weights W, uint16_t* __restrict__ ws, long long total) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int FF = W.d_ff, CD = W.c_dim, NXI = W.n_x;
  const long long frag = gid / FRAG_E;
  const int e = (int)(gid % FRAG_E), lane = e >> 3, j = e & 7, half = lane >> 5, r = lane & 31;
  const bool f16 = W.precision == QUATTRO_TF_PRECISION_F16;
  auto bf = [f16](float v) {
    return f16 ? __builtin_bit_cast(uint16_t, (_Float16)v) : __builtin_bit_cast(uint16_t, (__bf16)v);
  };
  if (frag < EMB_FRAGS) {   // embedding: natural k order (the B operand is built by the lane itself)
    const int ft = (int)frag / 5, s = (int)frag % 5, row = 32 * ft + r, k = 8 * half + j;
    uint16_t v = 0;
    if (s == 0) v = W.w_state[row * 16 + k];
    else if (16 * (s - 1) + k < CD) v = bf(W.ctrl_w[row * CD + 16 * (s - 1) + k]);
    ws[gid] = v;
    return;
  }
  const long long pf = frag - EMB_FRAGS;
  const int ppl = tf_panels_per_layer(FF);
  const int panel = (int)(pf / 8), i = (int)(pf % 8);       // i: fragment of the panel, in consumption order
  const int layer = panel / ppl, q = panel % ppl - 1;       // q = -1: the layer's parameter panel (tf_pack_params_kernel)
  const int pk = perm_k(j, half);                            // accumulator-derived operands hold k in this order
  uint16_t v;
  if (layer >= W.n_layers) {                                 // output head: [parameters] rows 32 rt.., k-step i
    const int rt = panel - W.n_layers * ppl - 1;
    if (rt < 0) return;
    v = W.w_out[(32 * rt + r) * D + 16 * i + pk];            // w_out is zero-padded to 64 rows
  } else if (q < 0) {
    return;
  } else if (q < 16) {                                       // head q / 4: Q_h, K_h, V_h, then its slice of the out-projection
    const int h = q >> 2, which = q & 3;
    if (which < 3) v = W.w_qkv[layer][(size_t)(which * D + 32 * h + r) * D + 16 * i + pk];
    else v = W.w_o[layer][(size_t)(32 * (i >> 1) + r) * D + 32 * h + 16 * (i & 1) + pk];   // rows 32 ft.., the head's k-steps
  } else {                                                   // feed-forward, order W1_0, (W1_1, W2_0), ..., W2_last
    const int t = q - 16, nch = FF / 32;
    bool first;
    int c;
    if (t == 0) { first = true; c = 0; }
    else if (t == 2 * nch - 1) { first = false; c = nch - 1; }
    else { first = ((t - 1) & 1) == 0; c = first ? (t - 1) / 2 + 1 : (t - 1) / 2; }
    if (first) v = W.w_1[layer][(size_t)(32 * c + r) * D + 16 * i + pk];                                      // W1 chunk c
    else v = W.w_2[layer][(size_t)(32 * (i >> 1) + r) * FF + 32 * c + 16 * (i & 1) + pk];                    // W2 columns of chunk c
  }
  ws[gid] = v;
}

// parameter panels (2048 fp32 each, inside the weight stream) + the embedding normalisation block (p_stream)
__global__ void tf_pack_params_kernel(const quattro_tf_weights W, uint16_t* __restrict__ ws, float* __restrict__ ps, int total) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int FF = W.d_ff, cd = W.c_dim;
  const int blk = gid / 2048, o = gid % 2048;
  float v = 0.0f;
  if (blk > W.n_layers) {                                    // embedding normalisation: [x_istd 16 | u_mean 64 | u_istd 64]
    if (o >= 256) return;
    if (o < 16) v = o < W.n_x ? 1.0f / W.x_std[o] : 0.0f;
    else if (o < 80) v = (o - 16) < cd ? W.u_mean[o - 16] : 0.0f;
    else if (o < 144) v = (o - 80) < cd ? 1.0f / W.u_std[o - 80] : 0.0f;
    ps[o] = v;
    return;
  }
  if (blk < W.n_layers) {
    const int sec = o >> 7, i = o & 127;
    if (o >= P_B1) v = (o - P_B1) < FF ? W.b_1[blk][o - P_B1] : 0.0f;
    else if (sec == 0) v = W.b_qkv[blk][i];
    else if (sec == 1) v = 0.0f;                             // (V's bias lives in the next section)
    else if (sec == 2) {                                     // b_o + W_o b_v
      v = W.b_o[blk][i];
      for (int j = 0; j < D; ++j)
        v = fmaf(W.precision == QUATTRO_TF_PRECISION_F16 ? (float)__builtin_bit_cast(_Float16, W.w_o[blk][(size_t)i * D + j])
                                                          : (float)__builtin_bit_cast(__bf16, W.w_o[blk][(size_t)i * D + j]),
                 W.b_qkv[blk][2 * D + j], v);
    } else if (sec == 3) v = W.ln1_g[blk][i];
    else if (sec == 4) v = W.ln1_b[blk][i];
    else if (sec == 5) v = W.b_2[blk][i];
    else if (sec == 6) v = W.ln2_g[blk][i];
    else v = W.ln2_b[blk][i];
  } else {
    if (o < 64) v = o < cd ? W.b_out[o] : 0.0f;
    else if (o < 128) v = (o - 64) < cd ? W.u_std[o - 64] : 0.0f;
    else if (o < 192) v = (o - 128) < cd ? W.u_mean[o - 128] : 0.0f;
  }
  const size_t panel = (size_t)blk * tf_panels_per_layer(FF);     // first panel of the layer / of the output head
  float* dst = reinterpret_cast<float*>(ws + (size_t)EMB_FRAGS * FRAG_E + panel * 8 * FRAG_E);
  dst[o] = v;
}

bool stream_shape_ok(const quattro_tf_weights& W) {
  const int L = W.n_state_tok + W.prompt_len + W.target_len;
  return W.d_model == D && W.n_head == 4 && W.d_ff >= 64 && W.d_ff % 64 == 0 && W.d_ff <= 1024 && W.c_dim > 0 &&
         W.c_dim <= 64 && W.n_x > 0 && W.n_x <= QUATTRO_MAX_NX && L <= 128 && W.n_layers > 0 &&
         W.n_layers <= QUATTRO_TF_MAX_LAYERS && W.n_state_tok > 0 && W.prompt_len > 0 && W.target_len > 0 &&
         (W.precision == QUATTRO_TF_PRECISION_BF16 || W.precision == QUATTRO_TF_PRECISION_F16);
}

}  // namespace

size_t quattro_tf_stream_elems_impl(const quattro_tf_weights& W) {
  if (!stream_shape_ok(W)) return 0;
  return (size_t)EMB_FRAGS * FRAG_E + (size_t)(W.n_layers * tf_panels_per_layer(W.d_ff) + tf_out_panels(W.c_dim)) * 8 * FRAG_E;
}
size_t quattro_tf_param_floats_impl(const quattro_tf_weights& W) {
  if (!stream_shape_ok(W)) return 0;
  return 256;   // the embedding normalisation block (everything else travels in the weight stream)
}

int quattro_launch_tf_pack(const quattro_tf_weights& W, uint16_t* ws, float* ps, hipStream_t stream) {
  if (!stream_shape_ok(W)) return QUATTRO_ERR_UNSUPPORTED;
  const long long tw = (long long)quattro_tf_stream_elems_impl(W);
  const int tp = (W.n_layers + 2) * 2048;
  hipLaunchKernelGGL(tf_pack_weights_kernel, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, stream, W, ws, tw);
  hipLaunchKernelGGL(tf_pack_params_kernel, dim3((tp + 255) / 256), dim3(256), 0, stream, W, ws, ps, tp);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

#ifdef QT_TF_PROFILE
extern "C" int quattro_tf_stream_profile(const quattro_tf_weights* Wp, const float* x_err, const float* prompt, int B,
                                         float* pred, unsigned long long* dbg, void* stream) {
  const quattro_tf_weights& W = *Wp;
  const TfGainsOut go{nullptr, nullptr, nullptr, 0, 0, 0};
  hipLaunchKernelGGL((tf_stream_kernel<4, 512, __bf16>), dim3(B), dim3(256), 0, (hipStream_t)stream, W, x_err, prompt, pred, go, dbg);
  return (int)hipGetLastError();
}
#else
int quattro_launch_tf_stream(const quattro_tf_weights& W, const float* x_err, const float* prompt, int B, float* pred,
                             float* Kout, float* kout, const int32_t* active, int N, int n, int m, hipStream_t stream) {
  if (!stream_shape_ok(W) || W.w_stream == nullptr || W.p_stream == nullptr || W.tok_bias_t == nullptr)
    return QUATTRO_ERR_UNSUPPORTED;
  const TfGainsOut go{Kout, kout, active, N, n, m};
  const int L = W.n_state_tok + W.prompt_len + W.target_len;
  const int nw = (L + 31) / 32;
#define QT_TF_LAUNCH(NW_, FF_)                                                                                       \
  do {                                                                                                               \
    if (W.precision == QUATTRO_TF_PRECISION_F16)                                                                     \
      hipLaunchKernelGGL((tf_stream_kernel<NW_, FF_, _Float16>), dim3(B), dim3(64 * NW_), 0, stream, W, x_err, prompt, pred, go); \
    else                                                                                                             \
      hipLaunchKernelGGL((tf_stream_kernel<NW_, FF_, __bf16>), dim3(B), dim3(64 * NW_), 0, stream, W, x_err, prompt, pred, go);   \
  } while (0)
  if (W.d_ff <= 512) {
    if (nw == 1) QT_TF_LAUNCH(1, 512);
    else if (nw == 2) QT_TF_LAUNCH(2, 512);
    else if (nw == 3) QT_TF_LAUNCH(3, 512);
    else QT_TF_LAUNCH(4, 512);
  } else {
    if (nw == 1) QT_TF_LAUNCH(1, 1024);
    else if (nw == 2) QT_TF_LAUNCH(2, 1024);
    else if (nw == 3) QT_TF_LAUNCH(3, 1024);
    else QT_TF_LAUNCH(4, 1024);
  }
#undef QT_TF_LAUNCH
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
#endif
