// Micro bench of the f16 x 2 plane product kernel (k_gemm_p3): time against the number of K steps at fixed output size -> per-launch
// fixed cost and per-step cost (round 4).
// build (repo root): hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ipsgd_tf_amd/csrc tools/micro/p3_bench.hip -Lpsgd_tf_amd/csrc -lpsgd_hip -Wl,-rpath,'$ORIGIN/../../psgd_tf_amd/csrc' -o tools/micro/p3_bench
#include "../../psgd_tf_amd/csrc/psgd_kron.hip"
#include <cstdio>
#include <vector>
#include <functional>
static float time_us(hipStream_t st, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3f;
}
int main() {
  const long cap = 4096L * 4096 * 2;               // two planes of a 4096 x 4096 operand
  std::vector<uint16_t> h(cap);
  unsigned s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3000 + ((s >> 16) & 0x7ff)); }   // f16 in [0.125, 0.5)
  __bf16 *A, *B; float* C; PlaneMeta* meta;
  hipMalloc(&A, cap * 2); hipMalloc(&B, cap * 2); hipMalloc(&C, 4096L * 4096 * 4); hipMalloc(&meta, 4 * sizeof(PlaneMeta));
  hipMemcpy(A, h.data(), cap * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), cap * 2, hipMemcpyHostToDevice);
  PlaneMeta pm[4] = {{1.f, 1.f, 0.5f, 0.5f}, {1.f, 1.f, 0.5f, 0.5f}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  hipMemcpy(meta, pm, sizeof(pm), hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  struct Case { int M, N, K, kmode; const char* what; };
  const Case cases[] = {{4096, 4096, 32, 0, "1024 tiles"}, {4096, 4096, 256, 0, ""}, {4096, 4096, 1024, 0, ""}, {4096, 4096, 2048, 0, ""},
                        {4096, 4096, 4096, 0, ""}, {4096, 2048, 32, 0, "512 tiles"}, {4096, 2048, 512, 0, ""}, {4096, 2048, 2048, 0, ""},
                        {4096, 2048, 2048, KHI_N, "512 tiles, K <= n0 + 127 (a solve's diagonal product)"},
                        {1024, 1024, 32, 0, "64 tiles"}, {1024, 1024, 512, 0, ""}, {2048, 2048, 32, 0, "256 tiles"}, {2048, 2048, 1024, 0, ""}};
  for (const Case& c : cases) {
    P3Buf a = {A, pad128(c.M), (long)((c.K + 31) / 32 * 32), meta}, b = {B, pad128(c.N), (long)((c.K + 31) / 32 * 32), meta + 1};
    P3Args g = p3_args(a, b, c.M, c.N, c.K, c.kmode);
    g.e.C = C; g.e.ldc = c.N; g.ometa = meta + 2;
    const float us = time_us(st, 20, [&] { launch_p3(g, st); });
    printf("%4d x %4d x %4d kmode %d: %7.1f us  (%d K steps) %s\n", c.M, c.N, c.K, c.kmode, us, c.K / 32, c.what);
  }
  // the gradient grid of the large update (two gradients, upper tiles, each tile two operand pairs of K = 4096: A A' - Bt Bt') against
  // one symmetric product with a single pair of K = 8192 (same MFMA work per tile)
  {
    const int n = 4096;
    __bf16 *A2, *B2;                                   // (8192-deep planes for the single-pair case)
    hipMalloc(&A2, cap * 4); hipMemcpy(A2, h.data(), cap * 2, hipMemcpyHostToDevice); hipMemcpy(A2 + cap, h.data(), cap * 2, hipMemcpyHostToDevice);
    float* scratch; unsigned* cnt;
    hipMalloc(&scratch, 512L * 64 * kThreads * 4 * 2); hipMalloc(&cnt, 4096); hipMemset(cnt, 0, 4096);
    P3Buf a = {A, pad128(n), (long)n, meta}, b = {B, pad128(n), (long)n, meta + 1};
    P3Args g1 = p3_args(a, a, n, n, n, 0);
    g1.A2 = p3_of(b); g1.B2 = p3_of(b); g1.e.A2 = C; g1.e.K2 = n;
    g1.e.epi = EPI_TRIU_MAX; g1.e.maxout = reinterpret_cast<float*>(meta + 3); g1.e.C = C; g1.e.ldc = n;
    P3Args g2 = g1;
    const float us = time_us(st, 10, [&] { launch_p3_grad(g1, g2, scratch, cnt, st); });
    printf("gradient grid, 2 x 528 tiles x (128 + 128) K steps: %.1f us\n", us);
    P3Buf a8 = {A2, pad128(n), 8192L, meta};
    {
      P3Args q = p3_args(a8, a8, n, n, 8192, 0);                   // the same grid, every tile ONE pair of 256 K steps
      q.e.epi = EPI_TRIU_MAX; q.e.maxout = reinterpret_cast<float*>(meta + 3); q.e.C = C; q.e.ldc = n;
      P3Args q2 = q;
      const float us3 = time_us(st, 10, [&] { launch_p3_grad(q, q2, scratch, cnt, st); });
      printf("gradient grid, 2 x 528 tiles x 256 K steps in ONE pair: %.1f us\n", us3);
    }
    P3Args s1 = p3_args(a8, a8, n, n, 8192, 0);
    s1.e.sym = 0; s1.e.epi = EPI_TRIU_MAX; s1.e.maxout = reinterpret_cast<float*>(meta + 3); s1.e.C = C; s1.e.ldc = n;
    const float us2 = time_us(st, 10, [&] { launch_p3(s1, st); });
    printf("one product 4096 x 4096 x 8192, triu epilogue (1024 tiles launched, the 496 below the diagonal skip their K loop): %.1f us  -> x 2 gradients = %.1f us\n", us2, 2 * us2);
  }
  // an empty kernel on the same stream, back to back: the launch floor
  const float e = time_us(st, 50, [&] { hipLaunchKernelGGL(k_absmax, dim3(1), dim3(kThreads), 0, st, C, 0L, 1L, 4L, C + 1024, (float*)nullptr, 0); });
  printf("one-block kernel, back to back: %.1f us\n", e);
  return 0;
}
