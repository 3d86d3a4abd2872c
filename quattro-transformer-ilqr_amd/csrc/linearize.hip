// Linearisation kernels: derivative records for every (trajectory, step), terminal V_x / V_xx, and the
// utility that packs separately stored blocks into records.
//
// Arithmetic replaced (reference quattro_ilqr_tf/quattro_ilqr_tf.py): _compute_dynamics_jacobians :182-204,
// _compute_cost_derivatives :217-275, _finite_diff_gradient_final :149-157, _finite_diff_hessian_final :163-174,
// with exact derivatives of the built-in device models (models_device.h) instead of finite differences.
//
// HBM-bound, write side: one record is 1,664 B (quadrotor) for 64 B of input.  Each thread computes the
// state-dependent entries of one (b,t) item; records are assembled in LDS and leave as full-width (16 B per lane,
// 1 KiB per wave-instruction) contiguous stores.  Structural zeros and constants are written to the LDS staging
// records once per block and never again.
#include "models_device.h"
#ifdef QT_USER_MODEL_HEADER
#include "user_linearize.h"
#endif

#define COMMA ,

namespace {

constexpr int LIN_THREADS = 64;
constexpr int LIN_STAGE = 16;  // records staged per round

template <int MODEL, class L>
__global__ __launch_bounds__(LIN_THREADS) void linearize_euler_kernel(const quattro_model_params p,
                                                                      const float* __restrict__ x,
                                                                      const float* __restrict__ u, int N, int t_start,
                                                                      int total, float* __restrict__ rec) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU;
  constexpr int STRIDE = L::STRIDE;
  constexpr int CH = STRIDE / 4;
  __shared__ __attribute__((aligned(16))) float s_stage[LIN_STAGE * STRIDE];
  const int lane = threadIdx.x;
  const int S = N - t_start;
  const int g0 = blockIdx.x * LIN_THREADS;

  for (int i = lane; i < LIN_STAGE * STRIDE; i += LIN_THREADS) s_stage[i] = 0.0f;
  __syncthreads();
  if (lane < LIN_STAGE) EulerRecord<MODEL, L>::fill_const(&s_stage[lane * STRIDE], p);

  const int g = g0 + lane;
  float xs[NX], us[NU];
  if (g < total) {
    const int b = g / S, t = t_start + g % S;
    const float* px = x + ((size_t)b * (N + 1) + t) * NX;
    const float* pu = u + ((size_t)b * N + t) * NU;
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = px[i];
#pragma unroll
    for (int a = 0; a < NU; ++a) us[a] = pu[a];
  }
  __syncthreads();
  for (int round = 0; round < LIN_THREADS / LIN_STAGE; ++round) {
    if ((lane / LIN_STAGE) == round && g < total)
      EulerRecord<MODEL, L>::fill_state(&s_stage[(lane % LIN_STAGE) * STRIDE], p, xs, us);
    __syncthreads();
    const int first = g0 + round * LIN_STAGE;
    int cnt = total - first;
    cnt = cnt > LIN_STAGE ? LIN_STAGE : cnt;
    if (cnt > 0) {
      float4* dst = reinterpret_cast<float4*>(rec + (size_t)first * STRIDE);
      const float4* src = reinterpret_cast<const float4*>(s_stage);
      for (int i = lane; i < cnt * CH; i += LIN_THREADS) dst[i] = src[i];
    }
    __syncthreads();
  }
}

// TILE16C (Euler quadrotor): only the state-dependent part of a record is produced per item, 76 floats instead of
// 416, plus ONE constant header record per call (block 0).  All 64 lanes of a block linearise their own item at the
// same time into a 64-record LDS stage (19 KB; fill_const / fill_state are the shared model code, writing through the
// Tile16CRec offset map, which sends every constant entry to a sink slot), then the stage leaves as contiguous
// 16-byte-per-lane stores.
__global__ __launch_bounds__(LIN_THREADS) void linearize_compact_kernel(const quattro_model_params p,
                                                                        const float* __restrict__ x,
                                                                        const float* __restrict__ u, int N, int t_start,
                                                                        int total, float* __restrict__ rec,
                                                                        float* __restrict__ VxN,
                                                                        float* __restrict__ VxxN) {
  constexpr int MODEL = QUATTRO_MODEL_QUADROTOR;
  using L = Tile16CRec;
  constexpr int NX = 12, NU = 4, STRIDE = L::STRIDE, CH = STRIDE / 4;
  __shared__ __attribute__((aligned(16))) float s_stage[LIN_THREADS * STRIDE];
  const int lane = threadIdx.x;
  const int S = N - t_start;
  const int g0 = blockIdx.x * LIN_THREADS;

  if (blockIdx.x == 0) {   // header: a plain TILE16 record holding every constant entry (and zeros elsewhere)
    static_assert(Tile16Rec::STRIDE <= LIN_THREADS * STRIDE, "header is staged in the same LDS buffer");
    for (int i = lane; i < Tile16Rec::STRIDE; i += LIN_THREADS) s_stage[i] = 0.0f;
    __syncthreads();
    if (lane == 0) EulerRecord<MODEL, Tile16Rec>::fill_const(s_stage, p);
    __syncthreads();
    for (int i = lane; i < Tile16Rec::STRIDE / 4; i += LIN_THREADS)
      reinterpret_cast<float4*>(rec)[i] = reinterpret_cast<const float4*>(s_stage)[i];
    __syncthreads();
  }

  float* mine = &s_stage[lane * STRIDE];
#pragma unroll
  for (int i = 0; i < CH; ++i) reinterpret_cast<float4*>(mine)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const int g = g0 + lane;
  if (g < total) {
    const int b = g / S, t = t_start + g % S;
    const float* px = x + ((size_t)b * (N + 1) + t) * NX;
    const float* pu = u + ((size_t)b * N + t) * NU;
    float xs[NX], us[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = px[i];
#pragma unroll
    for (int a = 0; a < NU; ++a) us[a] = pu[a];
    EulerRecord<MODEL, L>::fill_const(mine, p);
    EulerRecord<MODEL, L>::fill_state(mine, p, xs, us);
    if (t == N - 1 && VxN != nullptr) {
      // the item of a trajectory's last step also writes the terminal pair V_x(N) = 2 Qf (x_N - x_ref), V_xx(N) = 2 Qf
      // (the values terminal_kernel produces; one launch less per iteration)
      float* vx = VxN + (size_t)b * NX;
      float4* vxx = reinterpret_cast<float4*>(VxxN + (size_t)b * NX * NX);
#pragma unroll
      for (int i = 0; i < NX; ++i) vx[i] = 2.0f * p.qf[i] * (px[NX + i] - p.x_ref[i]);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int j4 = 0; j4 < NX / 4; ++j4) {
          const float d = 2.0f * p.qf[i];
          vxx[i * (NX / 4) + j4] = make_float4(i == 4 * j4 ? d : 0.0f, i == 4 * j4 + 1 ? d : 0.0f,
                                               i == 4 * j4 + 2 ? d : 0.0f, i == 4 * j4 + 3 ? d : 0.0f);
        }
      }
    }
  }
  __syncthreads();
  int cnt = total - g0;
  cnt = cnt > LIN_THREADS ? LIN_THREADS : cnt;
  float4* dst = reinterpret_cast<float4*>(rec + L::HEADER + (size_t)g0 * STRIDE);
  const float4* src = reinterpret_cast<const float4*>(s_stage);
  for (int i = lane; i < cnt * CH; i += LIN_THREADS) dst[i] = src[i];
}

// RK4 discretisation: column j of [A | B] = d x_next / d z_j is the forward-mode derivative of the four-stage step along
// the unit direction e_j (zero-order-hold u): LPI lanes per (b,t) item, lane j pushes direction j through the stages with
// the analytic JVP of the rate function (models_device.h).  The record is zero-filled by the caller (hipMemsetAsync);
// lane 0 of each item adds the integrator-independent cost entries.  Not on the headline path (both shipped drivers
// integrate with Euler, quadrotor_sim.py:100, cartpole_sim.py:63), so simple rather than staged through LDS.
template <int MODEL, class L, int LPI>
__global__ __launch_bounds__(64) void linearize_rk4_kernel(const quattro_model_params p, const float* __restrict__ x,
                                                          const float* __restrict__ u, int N, int t_start, int total,
                                                          float* __restrict__ rec) {
  constexpr int NX = ModelDims<MODEL>::NX, NU = ModelDims<MODEL>::NU, NZ = NX + NU;
  const int lane = threadIdx.x;
  const int g = blockIdx.x * (64 / LPI) + lane / LPI;
  const int j = lane % LPI;
  if (g >= total || j >= NZ) return;
  const int S = N - t_start;
  const int b = g / S, t = t_start + g % S;
  float xs[NX], us[NU];
  const float* px = x + ((size_t)b * (N + 1) + t) * NX;
  const float* pu = u + ((size_t)b * N + t) * NU;
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = px[i];
#pragma unroll
  for (int a = 0; a < NU; ++a) us[a] = pu[a];
  float dx0[NX], du[NU];
#pragma unroll
  for (int i = 0; i < NX; ++i) dx0[i] = (i == j) ? 1.0f : 0.0f;
#pragma unroll
  for (int a = 0; a < NU; ++a) du[a] = (NX + a == j) ? 1.0f : 0.0f;
  const float dt = p.dt;
  float k[NX], dk[NX], xst[NX], dxs[NX], acc[NX];
  // stage 1
  qt_rate<MODEL>(p, xs, us, k);
  qt_rate_jvp<MODEL>(p, xs, us, dx0, du, dk);
#pragma unroll
  for (int i = 0; i < NX; ++i) { acc[i] = dk[i]; xst[i] = fmaf(0.5f * dt, k[i], xs[i]); dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
  // stage 2
  qt_rate<MODEL>(p, xst, us, k);
  qt_rate_jvp<MODEL>(p, xst, us, dxs, du, dk);
#pragma unroll
  for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); xst[i] = fmaf(0.5f * dt, k[i], xs[i]); dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
  // stage 3
  qt_rate<MODEL>(p, xst, us, k);
  qt_rate_jvp<MODEL>(p, xst, us, dxs, du, dk);
#pragma unroll
  for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); xst[i] = fmaf(dt, k[i], xs[i]); dxs[i] = fmaf(dt, dk[i], dx0[i]); }
  // stage 4
  qt_rate_jvp<MODEL>(p, xst, us, dxs, du, dk);
  float* r = rec + (size_t)g * L::STRIDE;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const float v = fmaf(dt / 6.0f, acc[i] + dk[i], dx0[i]);
    if (j < NX) {
      r[L::a(i, j < NX ? j : 0)] = v;
    } else {
      r[L::b(i, j >= NX ? j - NX : 0)] = v;
    }
  }
  if (j == 0) fill_cost_entries<MODEL, L>(r, p, xs, us);
}

// RK4 quadrotor -> TILE16R records, ONE LANE PER ITEM.  The lane evaluates the four stage points once (QuadStage: trig,
// body rates, thrust), then pushes the unit directions of z = (x, u) that can reach anything — the three angles, the three
// body rates, the four controls; position and velocity directions give constant columns (quattro_device.h) — through the
// stages one after the other; with a unit seed most of a direction's first-stage arithmetic folds away at compile time.
// Against 16 lanes per item (linearize_rk4_kernel) every lane of a wave works on a different item: the trig of a stage is
// computed once per item instead of once per direction, there is no zero-fill pass over the record buffer (the kernel writes
// every float of a record), and a record is 624 B instead of 1,664 B (912 B while all 16 columns were stored).
// Stores: a lane finishing a column holds 48 bytes of ITS record; 64 lanes storing those directly are 64 separate
// 16-byte writes per instruction, and the write-through L2 forwards each as a request of its own (measured: 150 us
// for 184 MB).  So the columns go to an LDS stage, and the workgroup's 64 consecutive records leave in THREE passes of
// contiguous runs — four columns (192 B per record), four columns, then the last two with l_uu + l_z + padding behind them
// (240 B per record): 68 floats of stage per lane, 17 KB per wave.  What holds the kernel at one wave per SIMD is registers,
// not LDS: the coefficients of the stage Jacobians 2-4 (~90 values) stay live across all columns.  Block 0 also writes the
// header record (l_xx = 2Q and the six constant columns).
constexpr int RK4Q_PITCH = 68;    // floats per staged row = the widest pass (60) rounded so that 68 / 4 = 17 is odd: 16-byte rows on distinct banks

__global__ __launch_bounds__(64) void linearize_rk4_quad_kernel(const quattro_model_params p,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ u, int N, int t_start,
                                                                 int total, float* __restrict__ rec) {
  using L = Tile16RRec;
  constexpr int NX = 12, NU = 4, MODEL = QUATTRO_MODEL_QUADROTOR;
  static_assert(L::F == 0 && L::NCOL == 10 && L::LUU == 120 && L::LZ == 136 && L::STRIDE == 156, "three-pass flush assumes this record");
  __shared__ __attribute__((aligned(16))) float s_stage[64 * RK4Q_PITCH];
  const int lane = threadIdx.x;
  if (blockIdx.x == 0) {
    // header (a plain TILE16 record): zeros, the constant cost entries, and the six constant columns of [A | B] — a position
    // direction stays a unit vector, a velocity direction is e_v + dt e_p through all four stages (quattro_device.h)
    for (int i = lane; i < Tile16Rec::STRIDE; i += 64) rec[i] = 0.0f;
    __syncthreads();
    if (lane < NX) rec[Tile16Rec::lxx(lane, lane)] = 2.0f * p.q[lane];
    if (lane < 6) rec[Tile16Rec::a(lane, lane)] = 1.0f;
    if (lane < 3) rec[Tile16Rec::a(lane, 3 + lane)] = p.dt;
  }
  const int g0 = blockIdx.x * 64;
  const int cnt = total - g0 < 64 ? total - g0 : 64;         // items of this block
  const int g = g0 + (lane < cnt ? lane : 0);                // idle lanes of the last block recompute item g0 (never stored)
  const int S = N - t_start;
  const int b = g / S, t = t_start + g % S;
  float xs[NX], us[NU];
  {
    const float4* px = reinterpret_cast<const float4*>(x + ((size_t)b * (N + 1) + t) * NX);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 v = px[i];
      xs[4 * i + 0] = v.x; xs[4 * i + 1] = v.y; xs[4 * i + 2] = v.z; xs[4 * i + 3] = v.w;
    }
    const float4 v = *reinterpret_cast<const float4*>(u + ((size_t)b * N + t) * NU);
    us[0] = v.x; us[1] = v.y; us[2] = v.z; us[3] = v.w;
  }
  const float dt = p.dt;
  float k[NX], xst[NX];
  const QuadStage s1 = quad_stage(p, xs, us);
  quad_rate_at(s1, p, xs, us, k);
#pragma unroll
  for (int i = 0; i < NX; ++i) xst[i] = fmaf(0.5f * dt, k[i], xs[i]);
  const QuadStage s2 = quad_stage(p, xst, us);
  quad_rate_at(s2, p, xst, us, k);
#pragma unroll
  for (int i = 0; i < NX; ++i) xst[i] = fmaf(0.5f * dt, k[i], xs[i]);
  const QuadStage s3 = quad_stage(p, xst, us);
  quad_rate_at(s3, p, xst, us, k);
#pragma unroll
  for (int i = 0; i < NX; ++i) xst[i] = fmaf(dt, k[i], xs[i]);
  const QuadStage s4 = quad_stage(p, xst, us);

  float* mine = s_stage + lane * RK4Q_PITCH;
  float* out = rec + L::HEADER + (size_t)g0 * L::STRIDE;     // the block's 64 consecutive records
  // column j of [A | B] -> 12 floats at `dst`
  auto column = [&](int j, float* dst) __attribute__((always_inline)) {
    float dx0[NX], du[NU], dk[NX], dxs[NX], acc[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) dx0[i] = (i == j) ? 1.0f : 0.0f;
#pragma unroll
    for (int a = 0; a < NU; ++a) du[a] = (NX + a == j) ? 1.0f : 0.0f;
    quad_jvp_at(s1, p, dx0, du, dk);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] = dk[i]; dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
    quad_jvp_at(s2, p, dxs, du, dk);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); dxs[i] = fmaf(0.5f * dt, dk[i], dx0[i]); }
    quad_jvp_at(s3, p, dxs, du, dk);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] = fmaf(2.0f, dk[i], acc[i]); dxs[i] = fmaf(dt, dk[i], dx0[i]); }
    quad_jvp_at(s4, p, dxs, du, dk);
    float col[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) col[i] = fmaf(dt / 6.0f, acc[i] + dk[i], dx0[i]);
    float4* d4 = reinterpret_cast<float4*>(dst);
    d4[0] = make_float4(col[0], col[1], col[2], col[3]);
    d4[1] = make_float4(col[4], col[5], col[6], col[7]);
    d4[2] = make_float4(col[8], col[9], col[10], col[11]);
  };
  // `per_rec` float4 pieces of every staged row -> the records' floats [rec_off, rec_off + 4 per_rec)
  auto flush = [&](int per_rec, int rec_off) __attribute__((always_inline)) {
    __syncthreads();
    for (int q = lane; q < cnt * per_rec; q += 64) {
      const int r = q / per_rec, piece = q - r * per_rec;
      *reinterpret_cast<float4*>(out + (size_t)r * L::STRIDE + rec_off + 4 * piece) =
          *reinterpret_cast<const float4*>(s_stage + r * RK4Q_PITCH + 4 * piece);
    }
    __syncthreads();
  };
  // stored column d = direction of (angle d) for d < 3, (rate d - 3) for d < 6, (control d - 6) otherwise; the position and
  // velocity directions are never pushed through the stages (constant columns, header record)
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int d = 4 * pass + cc;
      if (d < L::NCOL) column(d < 6 ? 6 + d : NX + (d - 6), mine + 12 * cc);
    }
    if (pass < 2) {
      flush(12, 48 * pass);
    } else {
      float* tail = mine + 24 - L::LUU;                      // l_uu, l_z, padding: staged at their record offsets - 96
#pragma unroll
      for (int i = 0; i < 9; ++i) reinterpret_cast<float4*>(mine + 24)[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      fill_cost_entries<MODEL, L>(tail, p, xs, us);          // l_x, l_u, diag(l_uu); the constant diag(l_xx) goes to the sink
      flush(15, 96);
    }
  }
}

template <int MODEL>
__global__ void terminal_kernel(const quattro_model_params p, const float* __restrict__ x, int B, int N,
                                float* __restrict__ VxN, float* __restrict__ VxxN) {
  constexpr int NX = ModelDims<MODEL>::NX;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (b, i)
  if (g >= B * NX) return;
  const int b = g / NX, i = g % NX;
  const float xi = x[((size_t)b * (N + 1) + N) * NX + i];
  VxN[g] = 2.0f * p.qf[i] * (xi - p.x_ref[i]);
  float* row = VxxN + (size_t)g * NX;
#pragma unroll
  for (int j = 0; j < NX; ++j) row[j] = (i == j) ? 2.0f * p.qf[i] : 0.0f;
}

// one thread per record float, enumerated in ROWMAJOR order; writes to the requested layout
template <int NX, int NU, class L>
__global__ void pack_kernel(const float* __restrict__ A, const float* __restrict__ Bm, const float* __restrict__ lx,
                            const float* __restrict__ lu, const float* __restrict__ lxx, const float* __restrict__ luu,
                            const float* __restrict__ lux, long long items, float* __restrict__ rec) {
  using R = RowMajorRec<NX, NU>;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= items * R::SIZE) return;
  const long long it = gid / R::SIZE;
  const int e = (int)(gid % R::SIZE);
  float v;
  int dst;
  if (e < R::B) {
    const int i = e / NX, j = e % NX;
    v = A[it * NX * NX + e];
    dst = L::a(i, j);
  } else if (e < R::LXX) {
    const int q = e - R::B, i = q / NU, a = q % NU;
    v = Bm[it * NX * NU + q];
    dst = L::b(i, a);
  } else if (e < R::LUX) {
    const int q = e - R::LXX, i = q / NX, j = q % NX;
    v = lxx[it * NX * NX + q];
    dst = L::lxx(i, j);
  } else if (e < R::LUU) {
    const int q = e - R::LUX, a = q / NX, j = q % NX;
    v = lux[it * NU * NX + q];
    dst = L::lux(a, j);
  } else if (e < R::LX) {
    const int q = e - R::LUU, a = q / NU, c = q % NU;
    v = luu[it * NU * NU + q];
    dst = L::luu(a, c);
  } else if (e < R::LU) {
    const int q = e - R::LX;
    v = lx[it * NX + q];
    dst = L::lx(q);
  } else {
    const int q = e - R::LU;
    v = lu[it * NU + q];
    dst = L::lu(q);
  }
  rec[it * L::STRIDE + dst] = v;
}

#ifdef QT_USER_MODEL_HEADER
// (device bodies: user_linearize.h — shared with the device-resident loop of solve_user.hip)
template <class L, bool RK4, int LPI>
__global__ __launch_bounds__(64) void linearize_user_kernel(const quattro_model_params p, const float* __restrict__ x,
                                                           const float* __restrict__ u, int N, int t_start, int total,
                                                           float* __restrict__ rec) {
  const int lane = threadIdx.x;
  const int g = blockIdx.x * (64 / LPI) + lane / LPI;
  const int j = lane % LPI;
  if (g >= total || j >= QT_USER_NX + QT_USER_NU) return;
  const int S = N - t_start;
  const int b = g / S, t = t_start + g % S;
  user_linearize_item<L, RK4>(p, x + ((size_t)b * (N + 1) + t) * QT_USER_NX, u + ((size_t)b * N + t) * QT_USER_NU,
                              rec + (size_t)g * L::STRIDE, j);
}

__global__ void terminal_user_kernel(const quattro_model_params p, const float* __restrict__ x, int B, int N,
                                     float* __restrict__ VxN, float* __restrict__ VxxN) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (b, i)
  if (g >= B * QT_USER_NX) return;
  const int b = g / QT_USER_NX, i = g % QT_USER_NX;
  user_terminal_row(p, x + ((size_t)b * (N + 1) + N) * QT_USER_NX, i, VxN + (size_t)b * QT_USER_NX,
                    VxxN + (size_t)b * QT_USER_NX * QT_USER_NX);
}
#endif

template <int MODEL, class L>
int launch_linearize(const quattro_model_params& p, const float* x, const float* u, int B, int N, int t_start,
                     float* rec, hipStream_t stream) {
  const int total = B * (N - t_start);
  const int blocks = (total + LIN_THREADS - 1) / LIN_THREADS;
  hipLaunchKernelGGL((linearize_euler_kernel<MODEL, L>), dim3(blocks), dim3(LIN_THREADS), 0, stream, p, x, u, N,
                     t_start, total, rec);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

}  // namespace

template <int MODEL, class L, int LPI>
int launch_linearize_rk4(const quattro_model_params& p, const float* x, const float* u, int B, int N, int t_start,
                         float* rec, hipStream_t stream) {
  const int total = B * (N - t_start);
  if (hipMemsetAsync(rec, 0, (size_t)total * L::STRIDE * sizeof(float), stream) != hipSuccess) return QUATTRO_ERR_LAUNCH;
  const int per_block = 64 / LPI;
  hipLaunchKernelGGL((linearize_rk4_kernel<MODEL, L, LPI>), dim3((total + per_block - 1) / per_block), dim3(64), 0,
                     stream, p, x, u, N, t_start, total, rec);
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_linearize(const quattro_model_params& p, const float* x, const float* u, int B, int N, int t_start,
                             int layout, float* rec, float* VxN, float* VxxN, hipStream_t stream) {
  if (p.integrator != QUATTRO_INTEGRATOR_EULER && p.integrator != QUATTRO_INTEGRATOR_RK4) return QUATTRO_ERR_UNSUPPORTED;
  const bool rk4 = p.integrator == QUATTRO_INTEGRATOR_RK4;
  int st;
  if (rk4 && p.model_id == QUATTRO_MODEL_CARTPOLE && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    st = launch_linearize_rk4<QUATTRO_MODEL_CARTPOLE, RowMajorRec<4, 1>, 8>(p, x, u, B, N, t_start, rec, stream);
  } else if (rk4 && p.model_id == QUATTRO_MODEL_QUADROTOR && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    st = launch_linearize_rk4<QUATTRO_MODEL_QUADROTOR, RowMajorRec<12, 4>, 16>(p, x, u, B, N, t_start, rec, stream);
  } else if (rk4 && p.model_id == QUATTRO_MODEL_QUADROTOR && layout == QUATTRO_LAYOUT_TILE16R) {
    const int total = B * (N - t_start);
    hipLaunchKernelGGL(linearize_rk4_quad_kernel, dim3((total + 63) / 64), dim3(64), 0, stream, p, x, u, N, t_start, total,
                       rec);
    st = hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
  } else if (rk4 && p.model_id == QUATTRO_MODEL_QUADROTOR && layout == QUATTRO_LAYOUT_TILE16) {
    st = launch_linearize_rk4<QUATTRO_MODEL_QUADROTOR, Tile16Rec, 16>(p, x, u, B, N, t_start, rec, stream);
#ifdef QT_USER_MODEL_HEADER
  } else if (p.model_id == QUATTRO_MODEL_USER && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    using R = RowMajorRec<QT_USER_NX, QT_USER_NU>;
    constexpr int NZ = QT_USER_NX + QT_USER_NU, LPI = NZ <= 8 ? 8 : (NZ <= 16 ? 16 : 32);
    const int total = B * (N - t_start);
    const dim3 grid((total + 64 / LPI - 1) / (64 / LPI));
    if (rk4)
      hipLaunchKernelGGL((linearize_user_kernel<R, true, LPI>), grid, dim3(64), 0, stream, p, x, u, N, t_start, total, rec);
    else
      hipLaunchKernelGGL((linearize_user_kernel<R, false, LPI>), grid, dim3(64), 0, stream, p, x, u, N, t_start, total, rec);
    st = hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
#if QT_USER_NX == 12 && QT_USER_NU == 4
  } else if (p.model_id == QUATTRO_MODEL_USER && layout == QUATTRO_LAYOUT_TILE16) {
    // a user model of the quadrotor's shape gets the quadrotor's sweep: plain TILE16 records for the MFMA tile kernel
    const int total = B * (N - t_start);
    const dim3 grid((total + 3) / 4);
    if (rk4)
      hipLaunchKernelGGL((linearize_user_kernel<Tile16Rec, true, 16>), grid, dim3(64), 0, stream, p, x, u, N, t_start, total, rec);
    else
      hipLaunchKernelGGL((linearize_user_kernel<Tile16Rec, false, 16>), grid, dim3(64), 0, stream, p, x, u, N, t_start, total, rec);
    st = hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
#endif
#endif
  } else if (rk4) {
    return QUATTRO_ERR_UNSUPPORTED;
  } else if (p.model_id == QUATTRO_MODEL_CARTPOLE && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    st = launch_linearize<QUATTRO_MODEL_CARTPOLE, RowMajorRec<4, 1>>(p, x, u, B, N, t_start, rec, stream);
  } else if (p.model_id == QUATTRO_MODEL_QUADROTOR && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    st = launch_linearize<QUATTRO_MODEL_QUADROTOR, RowMajorRec<12, 4>>(p, x, u, B, N, t_start, rec, stream);
  } else if (p.model_id == QUATTRO_MODEL_QUADROTOR && layout == QUATTRO_LAYOUT_TILE16) {
    st = launch_linearize<QUATTRO_MODEL_QUADROTOR, Tile16Rec>(p, x, u, B, N, t_start, rec, stream);
  } else if (p.model_id == QUATTRO_MODEL_QUADROTOR && layout == QUATTRO_LAYOUT_TILE16C) {
    const int total = B * (N - t_start);
    hipLaunchKernelGGL(linearize_compact_kernel, dim3((total + LIN_THREADS - 1) / LIN_THREADS), dim3(LIN_THREADS), 0,
                       stream, p, x, u, N, t_start, total, rec, VxN, VxxN);
    return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;   // terminal pair included
  } else {
    return QUATTRO_ERR_UNSUPPORTED;
  }
  if (st != QUATTRO_OK) return st;
  if (VxN != nullptr && VxxN != nullptr) {
    const int threads = 256;
    if (p.model_id == QUATTRO_MODEL_CARTPOLE) {
      hipLaunchKernelGGL((terminal_kernel<QUATTRO_MODEL_CARTPOLE>), dim3((B * 4 + threads - 1) / threads),
                         dim3(threads), 0, stream, p, x, B, N, VxN, VxxN);
#ifdef QT_USER_MODEL_HEADER
    } else if (p.model_id == QUATTRO_MODEL_USER) {
      hipLaunchKernelGGL(terminal_user_kernel, dim3((B * QT_USER_NX + threads - 1) / threads), dim3(threads), 0, stream, p, x,
                         B, N, VxN, VxxN);
#endif
    } else {
      hipLaunchKernelGGL((terminal_kernel<QUATTRO_MODEL_QUADROTOR>), dim3((B * 12 + threads - 1) / threads),
                         dim3(threads), 0, stream, p, x, B, N, VxN, VxxN);
    }
    if (hipGetLastError() != hipSuccess) return QUATTRO_ERR_LAUNCH;
  }
  return QUATTRO_OK;
}

namespace {
// records -> separately stored row-major blocks (the inverse of pack_kernel), one thread per block float.  SrcOf maps
// (item, which block, indices) to the float's position in `rec` for the layout at hand.
template <int NX, int NU, class SrcOf>
__global__ void unpack_kernel(const float* __restrict__ rec, long long items, float* __restrict__ A,
                              float* __restrict__ Bm, float* __restrict__ lx, float* __restrict__ lu,
                              float* __restrict__ lxx, float* __restrict__ luu, float* __restrict__ lux) {
  using R = RowMajorRec<NX, NU>;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= items * R::SIZE) return;
  const long long it = gid / R::SIZE;
  const int e = (int)(gid % R::SIZE);
  if (e < R::B) {
    A[it * NX * NX + e] = rec[SrcOf::a(it, e / NX, e % NX)];
  } else if (e < R::LXX) {
    const int q = e - R::B;
    Bm[it * NX * NU + q] = rec[SrcOf::b(it, q / NU, q % NU)];
  } else if (e < R::LUX) {
    const int q = e - R::LXX;
    lxx[it * NX * NX + q] = rec[SrcOf::lxx(it, q / NX, q % NX)];
  } else if (e < R::LUU) {
    const int q = e - R::LUX;
    lux[it * NU * NX + q] = rec[SrcOf::lux(it, q / NX, q % NX)];
  } else if (e < R::LX) {
    const int q = e - R::LUU;
    luu[it * NU * NU + q] = rec[SrcOf::luu(it, q / NU, q % NU)];
  } else if (e < R::LU) {
    const int q = e - R::LX;
    lx[it * NX + q] = rec[SrcOf::lx(it, q)];
  } else {
    const int q = e - R::LU;
    lu[it * NU + q] = rec[SrcOf::lu(it, q)];
  }
}

template <class L>
struct PlainSrc {   // ROWMAJOR / TILE16: every entry sits in the item's own record
  static __device__ long long at(long long it, int off) { return it * L::STRIDE + off; }
  static __device__ long long a(long long it, int i, int j) { return at(it, L::a(i, j)); }
  static __device__ long long b(long long it, int i, int c) { return at(it, L::b(i, c)); }
  static __device__ long long lxx(long long it, int i, int j) { return at(it, L::lxx(i, j)); }
  static __device__ long long lux(long long it, int c, int j) { return at(it, L::lux(c, j)); }
  static __device__ long long luu(long long it, int c, int d) { return at(it, L::luu(c, d)); }
  static __device__ long long lx(long long it, int i) { return at(it, L::lx(i)); }
  static __device__ long long lu(long long it, int c) { return at(it, L::lu(c)); }
};
struct CompactSrc {  // TILE16C: state-dependent entries in the item's compact record, constants in the header record
  using C = Tile16CRec;
  using H = Tile16Rec;
  static __device__ long long pick(long long it, int coff, int hoff) {
    return coff == C::DUMP ? (long long)hoff : (long long)C::HEADER + it * C::STRIDE + coff;
  }
  static __device__ long long a(long long it, int i, int j) { return pick(it, C::a(i, j), H::a(i, j)); }
  static __device__ long long b(long long it, int i, int c) { return pick(it, C::b(i, c), H::b(i, c)); }
  static __device__ long long lxx(long long, int i, int j) { return H::lxx(i, j); }
  static __device__ long long lux(long long, int c, int j) { return H::lux(c, j); }
  static __device__ long long luu(long long it, int c, int d) { return (long long)C::HEADER + it * C::STRIDE + C::luu(c, d); }
  static __device__ long long lx(long long it, int i) { return (long long)C::HEADER + it * C::STRIDE + C::lx(i); }
  static __device__ long long lu(long long it, int c) { return (long long)C::HEADER + it * C::STRIDE + C::lu(c); }
};

struct DenseFSrc {   // TILE16R: the 10 changing columns of F, l_uu, l_z in the item's record; the rest in the header record
  using R = Tile16RRec;
  using H = Tile16Rec;
  static __device__ long long at(long long it, int off) { return (long long)R::HEADER + it * R::STRIDE + off; }
  static __device__ long long pick(long long it, int roff, int hoff) { return roff == R::DUMP ? (long long)hoff : at(it, roff); }
  static __device__ long long a(long long it, int i, int j) { return pick(it, R::a(i, j), H::a(i, j)); }
  static __device__ long long b(long long it, int i, int c) { return pick(it, R::b(i, c), H::b(i, c)); }
  static __device__ long long lxx(long long, int i, int j) { return H::lxx(i, j); }
  static __device__ long long lux(long long, int c, int j) { return H::lux(c, j); }
  static __device__ long long luu(long long it, int c, int d) { return at(it, R::luu(c, d)); }
  static __device__ long long lx(long long it, int i) { return at(it, R::lx(i)); }
  static __device__ long long lu(long long it, int c) { return at(it, R::lu(c)); }
};

}  // namespace

int quattro_launch_unpack(const float* rec, int B, int S, int n, int m, int layout, float* A, float* Bm, float* lx,
                          float* lu, float* lxx, float* luu, float* lux, hipStream_t stream) {
  const long long items = (long long)B * S;
  const int threads = 256;
#define QT_UNPACK(NX_, NU_, SRC)                                                                                   \
  {                                                                                                               \
    const long long tot = items * RowMajorRec<NX_, NU_>::SIZE;                                                    \
    hipLaunchKernelGGL((unpack_kernel<NX_, NU_, SRC>), dim3((unsigned)((tot + threads - 1) / threads)),           \
                       dim3(threads), 0, stream, rec, items, A, Bm, lx, lu, lxx, luu, lux);                       \
  }
  if (n == 4 && m == 1 && layout == QUATTRO_LAYOUT_ROWMAJOR) QT_UNPACK(4, 1, PlainSrc<RowMajorRec<4 COMMA 1>>)
  else if (n == 12 && m == 4 && layout == QUATTRO_LAYOUT_ROWMAJOR) QT_UNPACK(12, 4, PlainSrc<RowMajorRec<12 COMMA 4>>)
  else if (n == 12 && m == 4 && layout == QUATTRO_LAYOUT_TILE16) QT_UNPACK(12, 4, PlainSrc<Tile16Rec>)
  else if (n == 12 && m == 4 && layout == QUATTRO_LAYOUT_TILE16C) QT_UNPACK(12, 4, CompactSrc)
  else if (n == 12 && m == 4 && layout == QUATTRO_LAYOUT_TILE16R) QT_UNPACK(12, 4, DenseFSrc)
#ifdef QT_USER_MODEL_HEADER
  else if (n == QT_USER_NX && m == QT_USER_NU && layout == QUATTRO_LAYOUT_ROWMAJOR)
    QT_UNPACK(QT_USER_NX, QT_USER_NU, PlainSrc<RowMajorRec<QT_USER_NX COMMA QT_USER_NU>>)
#endif
  else return QUATTRO_ERR_UNSUPPORTED;
#undef QT_UNPACK
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}

int quattro_launch_pack(const float* A, const float* Bm, const float* lx, const float* lu, const float* lxx,
                        const float* luu, const float* lux, int B, int S, int n, int m, int layout, float* rec,
                        hipStream_t stream) {
  const long long items = (long long)B * S;
  const int threads = 256;
  if (n == 4 && m == 1 && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    using R = RowMajorRec<4, 1>;
    // padding floats of the stride are never read by the sweep's arithmetic; zero them for determinism
    if (hipMemsetAsync(rec, 0, (size_t)items * R::STRIDE * sizeof(float), stream) != hipSuccess) return QUATTRO_ERR_LAUNCH;
    const long long tot = items * R::SIZE;
    hipLaunchKernelGGL((pack_kernel<4, 1, R>), dim3((unsigned)((tot + threads - 1) / threads)), dim3(threads), 0,
                       stream, A, Bm, lx, lu, lxx, luu, lux, items, rec);
  } else if (n == 12 && m == 4 && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    using R = RowMajorRec<12, 4>;
    const long long tot = items * R::SIZE;
    hipLaunchKernelGGL((pack_kernel<12, 4, R>), dim3((unsigned)((tot + threads - 1) / threads)), dim3(threads), 0,
                       stream, A, Bm, lx, lu, lxx, luu, lux, items, rec);
  } else if (n == 12 && m == 4 && layout == QUATTRO_LAYOUT_TILE16) {
    const long long tot = items * RowMajorRec<12, 4>::SIZE;
    hipLaunchKernelGGL((pack_kernel<12, 4, Tile16Rec>), dim3((unsigned)((tot + threads - 1) / threads)),
                       dim3(threads), 0, stream, A, Bm, lx, lu, lxx, luu, lux, items, rec);
#ifdef QT_USER_MODEL_HEADER
  } else if (n == QT_USER_NX && m == QT_USER_NU && layout == QUATTRO_LAYOUT_ROWMAJOR) {
    using R = RowMajorRec<QT_USER_NX, QT_USER_NU>;
    if (hipMemsetAsync(rec, 0, (size_t)items * R::STRIDE * sizeof(float), stream) != hipSuccess) return QUATTRO_ERR_LAUNCH;
    const long long tot = items * R::SIZE;
    hipLaunchKernelGGL((pack_kernel<QT_USER_NX, QT_USER_NU, R>), dim3((unsigned)((tot + threads - 1) / threads)),
                       dim3(threads), 0, stream, A, Bm, lx, lu, lxx, luu, lux, items, rec);
#endif
  } else {
    return QUATTRO_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? QUATTRO_OK : QUATTRO_ERR_LAUNCH;
}
