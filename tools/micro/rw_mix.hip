// micro-benchmark: does it matter WHICH wave issues the stores of a mixed read/write sweep?
// vmcnt retires loads and stores in issue order, so a wave that both loads and stores waits for its older stores'
// acknowledgements whenever it waits for younger loads.  Variants move the same bytes:
//   coupled    every wave reads its tiles ([N,20] fp32, 16-byte loads) and writes the output itself
//   decoupled  3 of 4 waves only read, the 4th only writes (constant data): no wave mixes loads and stores
// thin = 4 B/row output stream; thick = the whole 80 B/row tile written back.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OUT /*0 none, 1 thin, 2 thick*/, bool DECOUPLED>
__global__ __launch_bounds__(256) void k(const float* __restrict__ M, float* __restrict__ Mo, float* __restrict__ out,
                                         long ntiles) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.f;
  if (!DECOUPLED) {
    const long gw = (long)blockIdx.x * 4 + w, nw = (long)gridDim.x * 4;
    for (long tile = gw; tile < ntiles; tile += nw) {
      const f32x4* src = reinterpret_cast<const f32x4*>(M + tile * 64 * 20);
      f32x4 v[5];
#pragma unroll
      for (int q = 0; q < 5; ++q) v[q] = __builtin_nontemporal_load(src + lane + 64 * q);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 5; ++q) s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
      acc += s;
      if (OUT == 1) __builtin_nontemporal_store(s, out + tile * 64 + lane);
      if (OUT == 2) {
        f32x4* dst = reinterpret_cast<f32x4*>(Mo + tile * 64 * 20);
#pragma unroll
        for (int q = 0; q < 5; ++q) __builtin_nontemporal_store(v[q] + 1.0f, dst + lane + 64 * q);
      }
    }
  } else {
    // tiles of this block: b, b + G, ...; readers (waves 0..2) split them 3 ways, the writer takes all of them
    const long G = gridDim.x;
    if (w < 3) {
      for (long j = w; blockIdx.x + j * G < ntiles; j += 3) {
        const long tile = blockIdx.x + j * G;
        const f32x4* src = reinterpret_cast<const f32x4*>(M + tile * 64 * 20);
        f32x4 v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = __builtin_nontemporal_load(src + lane + 64 * q);
#pragma unroll
        for (int q = 0; q < 5; ++q) acc += v[q][0] + v[q][1] + v[q][2] + v[q][3];
      }
    } else {
      for (long tile = blockIdx.x; tile < ntiles; tile += G) {
        if (OUT == 1) __builtin_nontemporal_store(1.0f, out + tile * 64 + lane);
        if (OUT == 2) {
          f32x4* dst = reinterpret_cast<f32x4*>(Mo + tile * 64 * 20);
#pragma unroll
          for (int q = 0; q < 5; ++q) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, dst + lane + 64 * q);
        }
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}

int main() {
  const long N = 100000000, ntiles = N / 64;
  float *M, *Mo, *out;
  hipMalloc(&M, N * 20 * 4); hipMalloc(&Mo, N * 20 * 4); hipMalloc(&out, N * 4);
  hipMemset(M, 0, N * 20 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int grid, double bytes, bool inplace) {
    float* dst = inplace ? M : Mo;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, M, dst, out, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, M, dst, out, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  run("read only (coupled loop)", k<0, false>, 2048, N * 80.0, false);
  run("read only (3 of 4 waves read)", k<0, true>, 2048, N * 80.0, false);
  run("thin  coupled", k<1, false>, 2048, N * 84.0, false);
  run("thin  decoupled", k<1, true>, 2048, N * 84.0, false);
  run("thick coupled, out of place", k<2, false>, 2048, N * 160.0, false);
  run("thick decoupled, out of place", k<2, true>, 2048, N * 160.0, false);
  run("thick coupled, in place", k<2, false>, 2048, N * 160.0, true);
  run("thick decoupled, in place", k<2, true>, 2048, N * 160.0, true);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, (const f32x4*)M, (f32x4*)Mo, N * 5);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, (const f32x4*)M, (f32x4*)Mo, N * 5);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-44s %7.3f ms  %7.1f GB/s\n", "float4 grid-stride copy", ms, N * 160.0 / ms / 1e6);
  return 0;
}
