// Cost of a device-wide barrier inside one launch (MI355X), for the persistent small-layer Kron apply.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier && tools/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VARIANT>
__global__ __launch_bounds__(256) void k_barriers(unsigned* ctl, int nbar, int nblocks, float* sink) {
  float acc = 0.f;
  for (int b = 0; b < nbar; ++b) {
    __syncthreads();
    if (threadIdx.x == 0) {
      if (VARIANT != 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(b + 1) * (unsigned)nblocks;
      while ((int)(__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (VARIANT == 0) __builtin_amdgcn_s_sleep(2);
      }
      if (VARIANT != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    acc += 1.0f;
  }
  if (acc < 0) sink[0] = acc;
}

template <int V>
static float run(int nblocks, int nbar, unsigned* ctl, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(ctl, 0, 64);
  hipLaunchKernelGGL(k_barriers<V>, dim3(nblocks), dim3(256), 0, 0, ctl, 4, nblocks, sink);
  hipMemset(ctl, 0, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_barriers<V>, dim3(nblocks), dim3(256), 0, 0, ctl, nbar, nblocks, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / nbar;
}

int main() {
  unsigned* ctl; float* sink;
  hipMalloc(&ctl, 256); hipMalloc(&sink, 256);
  for (int nb : {8, 57, 128, 256}) {
    printf("blocks %3d: sleep+fences %.2f us/barrier   fences, no sleep %.2f   no fences, no sleep %.2f\n", nb,
           run<0>(nb, 200, ctl, sink), run<1>(nb, 200, ctl, sink), run<2>(nb, 200, ctl, sink));
  }
  return 0;
}
