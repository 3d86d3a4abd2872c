// Micro-benchmark: the sweep's record access pattern with no arithmetic (how fast can 4096 waves stream their
// 50 x 1664-byte records backwards?).  Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC stream_rec.hip -o libstream_rec.so
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PF>
__global__ __launch_bounds__(64) void stream_kernel(const float* __restrict__ rec, int S, float* __restrict__ out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* base = rec + (size_t)b * S * 416;
  const float* pf = base + 3 * lane;
  const float* pq = base + 192 + 4 * (lane % 52);   // LXB + LUU = floats 192..399 (52 float4)
  const float* pz = base + 400 + (lane & 15);
  float acc = 0.0f;
  float f0[PF], f1[PF], f2[PF], lz[PF];
  f32x4 lq[PF];
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    const int off = (S - 1 - d > 0 ? S - 1 - d : 0) * 416;
    f0[d] = pf[off]; f1[d] = pf[off + 1]; f2[d] = pf[off + 2]; lq[d] = *(const f32x4*)(pq + off); lz[d] = pz[off];
  }
  for (int s = S - 1; s >= 0; s -= PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      acc += f0[d] + f1[d] + f2[d] + lq[d][0] + lq[d][1] + lq[d][2] + lq[d][3] + lz[d];
      const int sn = s - d - PF;
      const int off = (sn > 0 ? sn : 0) * 416;
      f0[d] = pf[off]; f1[d] = pf[off + 1]; f2[d] = pf[off + 2]; lq[d] = *(const f32x4*)(pq + off); lz[d] = pz[off];
    }
  }
  out[(size_t)b * 64 + lane] = acc;
}
extern "C" int stream_rec(const float* rec, int B, int S, int pf, float* out, void* stream) {
  if (pf == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(B), dim3(64), 0, (hipStream_t)stream, rec, S, out);
  else if (pf == 6) hipLaunchKernelGGL(stream_kernel<6>, dim3(B), dim3(64), 0, (hipStream_t)stream, rec, S, out);
  else hipLaunchKernelGGL(stream_kernel<1>, dim3(B), dim3(64), 0, (hipStream_t)stream, rec, S, out);
  return (int)hipGetLastError();
}
