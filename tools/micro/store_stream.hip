// micro-benchmark: streaming read of an [N,20] fp32 matrix (80 B/row, 16-byte loads) plus an optional [N] fp32
// output stream written (1) one dword per lane per 64-row tile, (2) one 16-byte store per lane per 256 rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ M, float* __restrict__ out, long ntiles) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  __shared__ float so[4][256];
  float acc = 0.f;
  // groups of 4 consecutive tiles per wave
  for (long grp = gw; grp * 4 + 3 < ntiles; grp += nw) {
    f32x4 o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long tile = grp * 4 + j;
      const f32x4* src = reinterpret_cast<const f32x4*>(M + tile * 64 * 20);
      f32x4 v[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) v[q] = NT ? __builtin_nontemporal_load(src + lane + 64 * q) : src[lane + 64 * q];
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 5; ++q) s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
      acc += s;
      if (MODE == 1) { if (NT) __builtin_nontemporal_store(s, out + tile * 64 + lane); else out[tile * 64 + lane] = s; }
      if (MODE == 2) so[threadIdx.x >> 6][64 * j + lane] = s;
    }
    if (MODE == 2) {
      __builtin_amdgcn_wave_barrier();
      o4 = *reinterpret_cast<f32x4*>(&so[threadIdx.x >> 6][4 * lane]);
      f32x4* dst = reinterpret_cast<f32x4*>(out + grp * 256) + lane;
      if (NT) __builtin_nontemporal_store(o4, dst); else *dst = o4;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  const long N = 100000000, ntiles = N / 64;
  float *M, *out;
  hipMalloc(&M, N * 20 * 4); hipMalloc(&out, N * 4);
  hipMemset(M, 0, N * 20 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, double bytes) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, M, out, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, M, out, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-34s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  run("read only", k<0, false>, N * 80.0);
  run("read only nt", k<0, true>, N * 80.0);
  run("read + dword store/lane", k<1, false>, N * 84.0);
  run("read + dword store/lane nt", k<1, true>, N * 84.0);
  run("read + 16B store per 4 tiles", k<2, false>, N * 84.0);
  run("read + 16B store per 4 tiles nt", k<2, true>, N * 84.0);
  return 0;
}
