// micro-benchmark: is the read rate of a sweep limited by HOW MANY streams it walks at once?
// Reads NM [N,20] fp32 matrices (16-byte loads, tile = 64 rows) and NV [N] fp32 vectors with trivial compute and no
// LDS, same tile -> wave mapping as the product sweeps (uvd_kernels.h: sweep_rows).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, bool PREFETCH, bool STORE = false>
__global__ __launch_bounds__(256) void k(const float* __restrict__ M0, const float* __restrict__ M1,
                                         const float* __restrict__ V0, float* __restrict__ out, long N, long ntiles) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float acc = 0.f;
  f32x4 v[NM][5];
  float s[NV > 0 ? NV : 1];
  auto issue = [&](long tile) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const f32x4* src = reinterpret_cast<const f32x4*>((m == 0 ? M0 : M1) + tile * 64 * 20);
#pragma unroll
      for (int q = 0; q < 5; ++q) v[m][q] = __builtin_nontemporal_load(src + lane + 64 * q);
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) s[j] = __builtin_nontemporal_load(V0 + (long)j * N + tile * 64 + lane);
  };
  auto consume = [&]() {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int q = 0; q < 5; ++q) acc += v[m][q][0] + v[m][q][1] + v[m][q][2] + v[m][q][3];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc += s[j];
  };
  if (!PREFETCH) {
    for (long tile = gw; tile < ntiles; tile += nw) { issue(tile); consume(); }
  } else {
    // software pipeline through LDS like sweep_rows: wait, park in LDS, issue next, compute from LDS
    __shared__ f32x4 lds[4][NM * 5 * 64 + 64];
    f32x4* my = lds[threadIdx.x >> 6];
    long tile = gw;
    if (tile < ntiles) issue(tile);
    while (tile < ntiles) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int q = 0; q < 5; ++q) my[(m * 5 + q) * 64 + lane] = v[m][q];
      float sv = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) sv += s[j];
      const long next = tile + nw;
      issue(next < ntiles ? next : tile);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int q = 0; q < 5; ++q) { const f32x4 x = my[(m * 5 + q) * 64 + lane]; acc += x[0] + x[1] + x[2] + x[3]; }
      acc += sv;
      if (STORE) __builtin_nontemporal_store(acc, out + tile * 64 + lane);
      __builtin_amdgcn_wave_barrier();
      tile = next;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const long N = 100000000, ntiles = N / 64;
  float *M0, *M1, *V0, *out;
  hipMalloc(&M0, N * 20 * 4); hipMalloc(&M1, N * 20 * 4); hipMalloc(&V0, N * 4 * 4); hipMalloc(&out, N * 4);
  hipMemset(M0, 0, N * 20 * 4); hipMemset(M1, 0, N * 20 * 4); hipMemset(V0, 0, N * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int grid, double bytes) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, M0, M1, V0, out, N, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, M0, M1, V0, out, N, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-52s grid %5d %7.3f ms  %7.1f GB/s\n", name, grid, ms, bytes / ms / 1e6);
  };
  for (int grid : {2048}) {
    run("1 matrix", k<1, 0, false>, grid, N * 80.0);
    run("1 matrix + 2 vectors", k<1, 2, false>, grid, N * 88.0);
    run("2 matrices", k<2, 0, false>, grid, N * 160.0);
    run("2 matrices + 3 vectors (Gram's streams)", k<2, 3, false>, grid, N * 172.0);
    run("2 matrices + 3 vectors, LDS-parked prefetch", k<2, 3, true>, grid, N * 172.0);
    run("1 matrix + 2 vectors, LDS-parked prefetch", k<1, 2, true>, grid, N * 88.0);
    run("1 matrix + 2 vectors + thin store (apply_s2 streams)", k<1, 2, true, true>, grid, N * 92.0);
  }
  return 0;
}
