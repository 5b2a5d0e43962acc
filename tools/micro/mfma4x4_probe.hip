// Operand / result layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 blocks, k = 1), found empirically:
// lane l supplies A = a[l], B = b[l]; prints which (lane_a, lane_b) product lands in D[vgpr e][lane l].
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma4x4_probe.hip -o /tmp/mfma4x4_probe && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int e = 0; e < 4; ++e) d[e * 64 + l] = c[e];
}

int main() {
  float ha[64], hb[64], hd[256], *a, *b, *d;
  // a[l] = distinct primes-like powers so that a product identifies the pair: a = 1 + l, b = 100 + l  -> product unique?
  for (int l = 0; l < 64; ++l) { ha[l] = (float)(l + 1); hb[l] = (float)(1000 + 64 * l); }
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  for (int e = 0; e < 4; ++e)
    for (int l = 0; l < 64; ++l) {
      int fa = -1, fb = -1;
      for (int x = 0; x < 64 && fa < 0; ++x)
        for (int y = 0; y < 64; ++y)
          if (ha[x] * hb[y] == hd[e * 64 + l]) { fa = x; fb = y; break; }
      if (l < 12 || l > 59) printf("D[e=%d][lane %2d] = a[lane %2d] * b[lane %2d]\n", e, l, fa, fb);
    }
  return 0;
}
