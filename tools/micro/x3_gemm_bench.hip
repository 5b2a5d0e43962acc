// Times the split (bf16 x 3) GEMM kernel of psgd_kron.hip on one M x N x K product per operand layout, with the
// what-if switches of X3_DBG (see r2s_x3 / x3_pass) to find what bounds the K loop.  Results are wrong with X3_DBG != 0.
//   for d in 0 1 2 4 7; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DX3_DBG=$d -Iinclude -Ipsgd_tf_amd/csrc \
//       tools/micro/x3_gemm_bench.hip -o /tmp/x3_$d && /tmp/x3_$d; done
#include "../../psgd_tf_amd/csrc/psgd_kron.hip"
#include <cstdio>
#include <vector>

static float run(const GemmArgs& g, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_gemm(g, 0);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch_gemm(g, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096;
  for (int i = 2; i + 1 < argc; i += 2) psgd_kron_set_tuning(atoi(argv[i]), atoi(argv[i + 1]));     // [key value] ...
  const size_t bytes = (size_t)n * n * sizeof(float);
  float *A, *B, *C;
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&C, bytes);
  std::vector<float> h((size_t)n * n);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  hipMemcpy(A, h.data(), bytes, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), bytes, hipMemcpyHostToDevice);
  const double flop = 2.0 * n * (double)n * n;
  const char* names[4] = {"A B   (KVEC, XROW)", "A B'  (KVEC, KVEC)", "A' B  (XROW, XROW)", "A' B' (XROW, KVEC)"};
  for (int v = 0; v < 4; ++v) {
    GemmArgs g = gemm_args(A, n, (v & 2) != 0, B, n, (v & 1) != 0, C, n, n, n, n);
    const float ms = run(g, 20);
    printf("X3_DBG=%d  %d^3  %s  %.3f ms  %.1f TFLOP/s fp32-equivalent  (%.0f issued bf16)\n", X3_DBG, n, names[v], ms,
           flop / ms * 1e-9, 6 * flop / ms * 1e-9);
  }
  {   // the same product on pre-split planes (k_gemm_p3)
    __bf16 *PA, *PB;
    hipMalloc(&PA, (size_t)n * n * 6); hipMalloc(&PB, (size_t)n * n * 6);
    const P3Buf a = {PA, n, n}, b = {PB, n, n};
    launch_split3(A, n, 1, n, n, a, 0);
    launch_split3(B, 1, n, n, n, b, 0);                      // (n, k) view of B
    __bf16* PC;
    hipMalloc(&PC, (size_t)n * n * 6);
    const P3Buf c = {PC, n, n};
    const char* outs[5] = {"fp32 C", "planes of C (row form)", "planes of C' (column form)", "fp32 C, KLO_N", "fp32 C, KHI_M"};
    for (int o = 0; o < 5; ++o) {
      P3Args g = p3_args(a, b, n, n, n, o == 3 ? KLO_N : (o == 4 ? KHI_M : 0));
      if (o == 0 || o >= 3) { g.e.C = C; g.e.ldc = n; }
      if (o == 1) p3_out_row(g, c);
      if (o == 2) p3_out_col(g, c);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) launch_p3(g, 0);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) launch_p3(g, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 20;
      printf("X3_DBG=%d  %d^3  planes (k_gemm_p3) -> %-27s %.3f ms  %.1f TFLOP/s fp32-equivalent  (%.0f issued bf16)\n", X3_DBG, n,
             outs[o], ms, flop / ms * 1e-9, 6 * flop / ms * 1e-9);
    }
  }
  {   // the same on f16 x 2 planes (k_gemm_p3<1>)
    __bf16 *PA, *PB;
    hipMalloc(&PA, (size_t)n * n * 6); hipMalloc(&PB, (size_t)n * n * 6);
    PlaneMeta* pm; float* part;
    hipMalloc(&pm, 64 * sizeof(PlaneMeta)); hipMalloc(&part, 4 * 2048 * 4);
    hipMemset(pm, 0, 64 * sizeof(PlaneMeta));
    P3Buf a = {PA, n, n, pm}, b = {PB, n, n, pm + 1};
    launch_absmax(A, (long)n * n, a, part, 0);
    launch_split3(A, n, 1, n, n, a, 0);
    launch_absmax(B, (long)n * n, b, part, 0);
    launch_split3(B, 1, n, n, n, b, 0);
    const char* outs[3] = {"fp32 C", "fp32 C, KLO_N", "fp32 C, KHI_M"};
    for (int o = 0; o < 3; ++o) {
      P3Args g = p3_args(a, b, n, n, n, o == 1 ? KLO_N : (o == 2 ? KHI_M : 0));
      g.e.C = C; g.e.ldc = n;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) launch_p3(g, 0);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) launch_p3(g, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 20;
      printf("X3_DBG=%d  %d^3  f16 x 2 planes (k_gemm_p3<1>) -> %-16s %.3f ms  %.1f TFLOP/s fp32-equivalent  (%.0f issued f16)\n", X3_DBG, n,
             outs[o], ms, flop / ms * 1e-9, 3 * flop / ms * 1e-9);
    }
  }
  {   // the gradient grid of the Kron update on f16 x 2 planes: triu(A A' - B B') and triu(A'A - B'B) in one launch (fp32 + max)
    __bf16* P[4];
    for (auto& q : P) hipMalloc(&q, (size_t)n * n * 6);
    PlaneMeta* pm; float *part, *G1, *G2, *scal, *scratch; unsigned* cnt;
    hipMalloc(&pm, 64 * sizeof(PlaneMeta)); hipMalloc(&part, 4 * 2048 * 4);
    hipMalloc(&G1, bytes); hipMalloc(&G2, bytes); hipMalloc(&scal, 256);
    hipMalloc(&scratch, (size_t)kGradSplitMax * kGradChunks * 64 * kThreads * 4); hipMalloc(&cnt, kGradSplitMax * 4);
    hipMemset(pm, 0, 64 * sizeof(PlaneMeta)); hipMemset(scal, 0, 256);
    P3Buf Ar = {P[0], n, n, pm}, Ac = {P[1], n, n, pm}, Br = {P[2], n, n, pm + 1}, Bc = {P[3], n, n, pm + 1};
    launch_absmax(A, (long)n * n, Ar, part, 0); Ac.part = Ar.part; Ac.npart = Ar.npart;
    launch_split3_both(A, n, 1, n, n, Ar, Ac, 0);
    launch_absmax(B, (long)n * n, Br, part, 0); Bc.part = Br.part; Bc.npart = Br.npart;
    launch_split3_both(B, n, 1, n, n, Br, Bc, 0);
    P3Args s2 = p3_args(Ar, Ar, n, n, n, 0);
    s2.A2 = p3_of(Br); s2.B2 = p3_of(Br); s2.e.A2 = B; s2.e.K2 = n;
    s2.e.epi = EPI_TRIU_MAX; s2.e.maxout = scal; s2.e.C = G1; s2.e.ldc = n;
    P3Args s3 = p3_args(Ac, Ac, n, n, n, 0);
    s3.A2 = p3_of(Bc); s3.B2 = p3_of(Bc); s3.e.A2 = B; s3.e.K2 = n;
    s3.e.epi = EPI_TRIU_MAX; s3.e.maxout = scal + 1; s3.e.C = G2; s3.e.ldc = n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_p3_grad(s2, s3, scratch, cnt, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch_p3_grad(s2, s3, scratch, cnt, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("X3_DBG=%d  %d^2  gradient grid on f16 x 2 planes (k_gemm_p3_grad<1>, both gradients)  %.3f ms  (%.0f issued f16 TFLOP/s)\n", X3_DBG, n, ms,
           3.0 * 2.0 * flop / ms * 1e-9);
    if (getenv("GRAD_WHATIF")) {        // where the grid's time goes: without the fp32 store, with one operand pair only
      auto timeit = [&](P3Args x, P3Args y, const char* what) {
        for (int i = 0; i < 3; ++i) launch_p3_grad(x, y, scratch, cnt, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch_p3_grad(x, y, scratch, cnt, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float t = 0.f; hipEventElapsedTime(&t, e0, e1);
        printf("    ... %s: %.3f ms\n", what, t / 20);
      };
      P3Args a2 = s2, a3 = s3;
      a2.e.C = nullptr; a3.e.C = nullptr;
      timeit(a2, a3, "no fp32 store (triu + max on the registers only)");
      P3Args b2 = s2, b3 = s3;
      b2.e.A2 = nullptr; b3.e.A2 = nullptr;
      timeit(b2, b3, "one operand pair (A A' only)");
      P3Args c2 = p3_args(Ar, Ar, n, n, n, 0), c3 = p3_args(Ac, Ac, n, n, n, 0);
      c2.e.C = G1; c2.e.ldc = n; c3.e.C = G2; c3.e.ldc = n;
      timeit(c2, c3, "one operand pair, plain store of the upper tiles");
    }
    if (getenv("GRAD_COLD")) {          // the same with the caches swept between the launches (1 GiB memset), the memsets timed alone too
      char* junk; hipMalloc(&junk, (size_t)1 << 30);
      float mset = 0.f, both = 0.f;
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) hipMemsetAsync(junk, i, (size_t)1 << 30, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&mset, e0, e1);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) { hipMemsetAsync(junk, i, (size_t)1 << 30, 0); launch_p3_grad(s2, s3, scratch, cnt, 0); }
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&both, e0, e1);
      printf("    ... behind a 1 GiB memset each time: %.3f ms per launch (memset alone %.3f ms)\n", (both - mset) / 10, mset / 10);
      hipFree(junk);
    }
  }
  {   // the two factor updates of the Kron update (K = [m0, n0 + 128): work = distance from the diagonal), as one grid
    __bf16 *PA, *PB;
    hipMalloc(&PA, (size_t)n * n * 6); hipMalloc(&PB, (size_t)n * n * 6);
    PlaneMeta* pm2; float* part2;
    hipMalloc(&pm2, 64 * sizeof(PlaneMeta)); hipMalloc(&part2, 4 * 2048 * 4);
    hipMemset(pm2, 0, 64 * sizeof(PlaneMeta));
    const bool f16 = getenv("PAIR_F16") != nullptr;          // PAIR_F16=1: the same leg on f16 x 2 planes
    P3Buf a = {PA, n, n, f16 ? pm2 : nullptr}, b = {PB, n, n, f16 ? pm2 + 1 : nullptr};
    if (f16) launch_absmax(A, (long)n * n, a, part2, 0);
    launch_split3(A, n, 1, n, n, a, 0);
    if (f16) launch_absmax(B, (long)n * n, b, part2, 0);
    launch_split3(B, 1, n, n, n, b, 0);
    float* scal; hipMalloc(&scal, 256); hipMemset(scal, 0, 256);
    for (int variant = 0; variant < 3; ++variant) {
      P3Args g = p3_args(a, b, n, n, n, KLO_M | KHI_N);
      g.e.C = C; g.e.ldc = n;
      if (variant >= 1) { g.e.epi = EPI_D_MINUS; g.e.D = A; g.e.ldd = n; }
      if (variant == 2) { g.e.scale_max = scal; g.e.step = 0.01f; g.e.tiny = 1e-30f; }
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) launch_p3_two(g, g, 0);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) launch_p3_two(g, g, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      printf("two products with K = [m0, n0 + 128) in one grid, %s: %.3f ms\n",
             variant == 0 ? "plain store" : (variant == 1 ? "C = D - A B" : "C = D - (step / max) A B"), ms / 20);
      if (variant == 0) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch_p3(g, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("one such product alone: %.3f ms\n", ms / 20);
        P3Args h = p3_args(a, b, n, n, n, 0);
        h.e.C = C; h.e.ldc = n; h.e.K = 128;                 // 4 K steps per tile: the fixed parts of 1024 blocks
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) launch_p3(h, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("a product with K = 128 (4 steps per tile, 1024 tiles): %.3f ms\n", ms / 20);
      }
    }
  }
  {   // one substitution strip of the blocked solves: 512 columns, n vectors (k_trsm_ut_reg_full), TRSM_DBG what-ifs
    float *Qs, *Xs, *Ys, *dinv;
    hipMalloc(&Qs, 512 * 512 * 4); hipMalloc(&Xs, (size_t)n * 512 * 4); hipMalloc(&Ys, (size_t)n * 512 * 4); hipMalloc(&dinv, 16 * 1024 * 4);
    std::vector<float> hq(512 * 512, 0.f);
    for (int i = 0; i < 512; ++i) { hq[i * 512 + i] = 1.0f; for (int j = i + 1; j < 512; ++j) hq[i * 512 + j] = 0.001f * ((i * 7 + j) % 13 - 6); }
    hipMemcpy(Qs, hq.data(), 512 * 512 * 4, hipMemcpyHostToDevice);
    hipMemcpy(Xs, A, (size_t)n * 512 * 4, hipMemcpyDeviceToDevice);
    hipLaunchKernelGGL(k_tri_inv32, dim3(16), dim3(64), 0, 0, Qs, 512, 512, dinv);
    TrsmArgs t = {Qs, 512, 512, Xs, Ys, n, 512L, 1L, 0L, 0L};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_strip(t, dinv, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) launch_strip(t, dinv, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("TRSM_DBG=%d  substitution strip, 512 columns x %d vectors: %.1f us\n", TRSM_DBG, n, ms / 50 * 1e3);
  }
  {   // the 128-block inversion of tri_inverse (k_tri_inv128: n / 128 workgroups, 132 KiB of LDS each) on its own
    float *Qt, *Inv, *dinv, *amax;
    hipMalloc(&Qt, bytes); hipMalloc(&Inv, bytes); hipMalloc(&dinv, (size_t)(n / 32 + 1) * 1024 * 4); hipMalloc(&amax, 256);
    hipMemset(amax, 0, 256);
    std::vector<float> hq((size_t)n * n, 0.f);
    for (int i = 0; i < n; ++i) { hq[(size_t)i * n + i] = 1.0f + 0.001f * (i % 7); for (int j = i + 1; j < n && j < i + 200; ++j) hq[(size_t)i * n + j] = 0.001f * ((i * 7 + j) % 13 - 6); }
    hipMemcpy(Qt, hq.data(), bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tri_inv32, dim3((n + 31) / 32), dim3(64), 0, 0, Qt, n, n, dinv);
    const size_t lds = (size_t)2 * 128 * 129 * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tri_inv128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_tri_inv128, dim3((n + 127) / 128), dim3(kThreads), lds, 0, Qt, n, dinv, Inv, amax);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_tri_inv128, dim3((n + 127) / 128), dim3(kThreads), lds, 0, Qt, n, dinv, Inv, amax);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    printf("k_tri_inv128, n = %d (%d workgroups): %.1f us\n", n, (n + 127) / 128, ms / 50 * 1e3);
  }
  {   // does the planes product depend on the data (matrix-core power) or on what ran before it?
    __bf16 *PA, *PB;
    hipMalloc(&PA, (size_t)n * n * 6); hipMalloc(&PB, (size_t)n * n * 6);
    const P3Buf a = {PA, n, n}, b = {PB, n, n};
    std::vector<float> hb((size_t)n * n);
    unsigned s2 = 777u;
    for (size_t i = 0; i < hb.size(); ++i) {                  // B = I + 0.02 * noise (a Gram of a near-identity factor)
      s2 = s2 * 1664525u + 1013904223u;
      hb[i] = 0.02f * (((s2 >> 8) & 0xFFFF) / 65536.0f - 0.5f) + ((i / n == i % n) ? 1.0f : 0.0f);
    }
    hipMemcpy(B, hb.data(), bytes, hipMemcpyHostToDevice);
    launch_split3(A, n, 1, n, n, a, 0);
    launch_split3(B, 1, n, n, n, b, 0);
    P3Args g = p3_args(a, b, n, n, n, 0);
    g.e.C = C; g.e.ldc = n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int i = 0; i < 3; ++i) launch_p3(g, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch_p3(g, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("planes, B = I + 0.02 noise:                     %.3f ms per product\n", ms / 20);
    float ms2 = 0.f;
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) launch_split3(A, n, 1, n, n, a, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms2, e0, e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) { launch_split3(A, n, 1, n, n, a, 0); launch_p3(g, 0); }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("planes, alternating with a split of A:          %.3f ms per product (+ %.3f ms per split)\n", (ms - ms2) / 20, ms2 / 20);
  }
  GemmArgs g = gemm_args(A, n, false, B, n, false, C, n, n, n, n, KLO_M);
  printf("X3_DBG=%d  triangular A (KLO_M)  %.3f ms\n", X3_DBG, run(g, 20));
  return 0;
}
