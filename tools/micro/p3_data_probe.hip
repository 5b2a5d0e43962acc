// Does the speed of the planes GEMM depend on the operand VALUES?  Runs k_gemm_p3 at 4096^3 for ~1.5 s per data set and
// prints the time per product over the last second; tools/p3_data_probe.sh samples clocks and power alongside.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ipsgd_tf_amd/csrc tools/micro/p3_data_probe.hip -o tools/micro/p3_data_probe
#include "../../psgd_tf_amd/csrc/psgd_kron.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <unistd.h>

int main(int argc, char** argv) {
  const int n = 4096;
  const size_t bytes = (size_t)n * n * sizeof(float);
  float *A, *B, *C;
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&C, bytes);
  __bf16 *PA, *PB;
  hipMalloc(&PA, (size_t)n * n * 6); hipMalloc(&PB, (size_t)n * n * 6);
  const P3Buf a = {PA, n, n}, b = {PB, n, n};
  std::vector<float> ha((size_t)n * n), hb((size_t)n * n);
  const char* names[6] = {"A, B uniform(-0.5, 0.5)", "A uniform, B = I + 0.02 uniform", "A uniform, B = 0.02 uniform", "A uniform, B = I",
                          "A uniform, B = 0", "A uniform, B = triu(0.02 uniform) + I"};
  for (int ds = 0; ds < 6; ++ds) {
    unsigned s = 12345u + ds;
    for (size_t i = 0; i < ha.size(); ++i) {
      s = s * 1664525u + 1013904223u;
      const float u = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f;
      s = s * 1664525u + 1013904223u;
      const float v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f;
      const bool diag = (i / n == i % n), upper = (i % n >= i / n);
      ha[i] = u;
      hb[i] = ds == 0 ? v : ds == 1 ? 0.02f * v + (diag ? 1.f : 0.f) : ds == 2 ? 0.02f * v : ds == 3 ? (diag ? 1.f : 0.f) : ds == 4 ? 0.f
                                                                                           : (upper ? 0.02f * v : 0.f) + (diag ? 1.f : 0.f);
    }
    hipMemcpy(A, ha.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), bytes, hipMemcpyHostToDevice);
    launch_split3(A, n, 1, n, n, a, 0);
    launch_split3(B, 1, n, n, n, b, 0);
    P3Args g = p3_args(a, b, n, n, n, 0);
    g.e.C = C; g.e.ldc = n;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 1000; ++i) launch_p3(g, 0);         // ~0.5 s of warm-up at this data
    hipEventRecord(e0, 0);
    for (int i = 0; i < 2000; ++i) launch_p3(g, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms per product\n", names[ds], ms / 2000);
    fflush(stdout);
  }
  return 0;
}
