// Micro bench of the 256^2 bf16 kernels on the shapes of the bf16-operand update's gradient products (round 4):
//   dense 8-phase kernel (k_hgemm_nt_256) at K = 4096 / 8192, row stride = K or K + 64, B = A or another matrix; stream-K launches.
// build (repo root):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ipsgd_tf_amd/csrc tools/micro/sk_bench.hip -Lpsgd_tf_amd/csrc -lpsgd_hip -Wl,-rpath,'$ORIGIN/../../psgd_tf_amd/csrc' -o tools/micro/sk_bench
#include "../../psgd_tf_amd/csrc/psgd_kron_bf16.hip"
#include <cstdio>
#include <vector>
using namespace psgdh;
static float time_ms(hipStream_t st, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
int main() {
  const int M = 4096;
  const long cap = (long)M * 8320;
  std::vector<uint16_t> h(cap);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 16) & 0x3ff)); if (s & 0x80000000u) v |= 0x8000; }
  uint16_t *A, *B; float* C; float* partial; uint16_t* Cb;
  hipMalloc(&A, cap * 2); hipMalloc(&B, cap * 2); hipMalloc(&C, (long)M * M * 4); hipMalloc(&Cb, (long)M * M * 2);
  hipMalloc(&partial, (long)2 * 256 * 65536 * 4);
  hipMemcpy(A, h.data(), cap * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), cap * 2, hipMemcpyHostToDevice);
  float* maxout; hipMalloc(&maxout, 256); hipMemset(maxout, 0, 256);
  hipStream_t st; hipStreamCreate(&st);
  for (int K : {4096, 8192})
    for (int pad : {0, 64})
      for (int same : {0, 1}) {
        const long ld = K + pad;
        HGemmArgs g = {A, ld, same ? A : B, ld, C, M, 0, 0, M, M, K, 0, 0};
        const float ms = time_ms(st, 10, [&] { hipLaunchKernelGGL(k_hgemm_nt_256<0>, dim3((M / T2) * (M / T2)), dim3(kThreads2), 0, st, g); });
        printf("dense 256^2  4096 x 4096 x %d  ld %ld  B %s: %.1f us  %.0f TFLOP/s\n", K, ld, same ? "= A" : "other", ms * 1e3, 2.0 * M * M * K / ms / 1e9);
      }
  for (int pad : {0, 64})
    for (int nprob : {1, 2}) {
      const long ld = 8192 + pad;
      HGemmArgs g = {A, ld, A, ld, Cb, M, 1, 0, M, M, 8192, 0, 1};
      g.epi = HEPI_TRIU_MAX; g.kflip = 4096 / TK; g.maxout = maxout;
      HGemmArgs two[2] = {g, g};
      two[1].A = two[1].B = B;
      const float ms = time_ms(st, 10, [&] { launch_hgemm_sk(two, nprob, partial, st); });
      const double fl = nprob * 136.0 * 2 * 256 * 256 * 8192;
      printf("stream-K gradients x %d  ld %ld: %.1f us (main + fix)  %.0f TFLOP/s\n", nprob, ld, ms * 1e3, fl / ms / 1e9);
    }
  for (int cb : {0, 1})
    for (int ct : {0, 1}) {
      HGemmArgs g = {A, 8256, B, 8256, cb ? (void*)Cb : (void*)C, M, cb, ct, M, M, 8192, 0, 0};
      const float ms = time_ms(st, 10, [&] { hipLaunchKernelGGL(k_hgemm_nt_256<0>, dim3((M / T2) * (M / T2)), dim3(kThreads2), 0, st, g); });
      const float ms2 = time_ms(st, 10, [&] { launch_hgemm_sk(&g, 1, partial, st); });
      printf("4096 x 4096 x 8192  C %s %s: dense kernel %.1f us, stream-K launch (256 whole tiles) %.1f us\n", cb ? "bf16" : "fp32", ct ? "transposed" : "row-major", ms * 1e3, ms2 * 1e3);
    }
  // the main launch alone, and with the hook-free plain dense problem through stream-K (sym = 0, STORE)
  {
    HGemmArgs g = {A, 8256, B, 8256, Cb, M, 1, 0, M, M, 8192, 0, 0};
    const float ms = time_ms(st, 10, [&] { launch_hgemm_sk(&g, 1, partial, st); });
    printf("stream-K dense 4096 x 4096 x 8192 (256 tiles: every range a whole tile): %.1f us  %.0f TFLOP/s\n", ms * 1e3, 2.0 * M * M * 8192 / ms / 1e9);
  }
  return 0;
}
