// micro-benchmark: ceiling for the last sparse-LU sweep's stream shape: one [N,R] row-major matrix read and rewritten
// in place (16-byte accesses) plus R "column" streams of length N (stride N apart) read and rewritten in place with
// 4-byte-per-lane accesses, plus 4 read-only vectors and 2 written vectors.  Trivial arithmetic, no LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int R = 10;            // 40-byte rows: tile = 128 rows = 1280 floats = 5 x (64 lanes x float4)

template <bool WRITE_COLS, bool WRITE_MAT, int COLVEC>
__global__ __launch_bounds__(256) void k(float* __restrict__ L, float* __restrict__ U, const float* __restrict__ vin,
                                         float* __restrict__ vout, long N, long ntiles) {
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  float acc = 0.f;
  for (long tile = gw; tile < ntiles; tile += nw) {
    f32x4* lp = reinterpret_cast<f32x4*>(L + tile * 128 * R);
    f32x4 m[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) m[q] = __builtin_nontemporal_load(lp + lane + 64 * q);
    float u[R][2], s[4][2];
    if (COLVEC == 1) {
#pragma unroll
      for (int k2 = 0; k2 < R; ++k2)
#pragma unroll
        for (int i = 0; i < 2; ++i) u[k2][i] = __builtin_nontemporal_load(U + (long)k2 * N + tile * 128 + lane + 64 * i);
    } else {   // 8-byte accesses: lane owns rows 2*lane, 2*lane+1
#pragma unroll
      for (int k2 = 0; k2 < R; ++k2) {
        const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(U + (long)k2 * N + tile * 128) + lane);
        u[k2][0] = v[0]; u[k2][1] = v[1];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) s[j][i] = __builtin_nontemporal_load(vin + (long)j * N + tile * 128 + lane + 64 * i);
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += s[j][0] + s[j][1];
    acc += t;
    if (WRITE_MAT) {
#pragma unroll
      for (int q = 0; q < 5; ++q) __builtin_nontemporal_store(m[q] + t, lp + lane + 64 * q);
    } else {
#pragma unroll
      for (int q = 0; q < 5; ++q) acc += m[q][0] + m[q][3];
    }
    if (WRITE_COLS) {
      if (COLVEC == 1) {
#pragma unroll
        for (int k2 = 0; k2 < R; ++k2)
#pragma unroll
          for (int i = 0; i < 2; ++i) __builtin_nontemporal_store(u[k2][i] + t, U + (long)k2 * N + tile * 128 + lane + 64 * i);
      } else {
#pragma unroll
        for (int k2 = 0; k2 < R; ++k2)
          __builtin_nontemporal_store(f32x2{u[k2][0] + t, u[k2][1] + t}, reinterpret_cast<f32x2*>(U + (long)k2 * N + tile * 128) + lane);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_nontemporal_store(t, vout + (long)j * N + tile * 128 + lane + 64 * i);
    } else {
#pragma unroll
      for (int k2 = 0; k2 < R; ++k2) acc += u[k2][0] + u[k2][1];
    }
  }
  if (acc == 12345.678f) vout[0] = acc;
}

int main() {
  const long N = 49999872, ntiles = N / 128;     // multiple of 128 and 2
  float *L, *U, *vin, *vout;
  hipMalloc(&L, N * R * 4); hipMalloc(&U, N * R * 4); hipMalloc(&vin, N * 4 * 4); hipMalloc(&vout, N * 2 * 4);
  hipMemset(L, 0, N * R * 4); hipMemset(U, 0, N * R * 4); hipMemset(vin, 0, N * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, double bytes) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, L, U, vin, vout, N, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, L, U, vin, vout, N, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-62s %7.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  run("read only: matrix + 10 columns + 4 vectors (96 B/row)", k<false, false, 1>, N * 96.0);
  run("+ rewrite the matrix in place (136 B/row)", k<false, true, 1>, N * 136.0);
  run("+ rewrite the 10 columns + 2 vectors, dword (144 B/row)", k<true, false, 1>, N * 144.0);
  run("rewrite both (sweep 4's shape, 184 B/row), dword columns", k<true, true, 1>, N * 184.0);
  run("rewrite both, 8-byte column accesses", k<true, true, 2>, N * 184.0);
  return 0;
}
