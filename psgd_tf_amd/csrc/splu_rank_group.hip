// splu_rank_group.hip -- instantiates the rank-templated sparse-LU kernels for the 8 ranks
// PSGD_RANK_LO .. PSGD_RANK_LO+7 (compiled four times, like uvd_rank_group.hip).
#include "splu_kernels.h"

#ifndef PSGD_RANK_LO
#error "compile with -DPSGD_RANK_LO=<first rank> -DPSGD_GROUP_FN=<symbol>"
#endif

namespace psgd {

// the sweep kernels use dynamic LDS (up to ~72 KB per block at r = 32): raise the per-kernel cap once
template <class F>
static inline void allow_lds(F* f, int bytes, bool& done) {
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done = true;
  }
}

#define SPLU_LAUNCH(KERNEL_T, KERNEL_F, LDS, ...)                                          \
  do {                                                                                     \
    static bool set_t = false, set_f = false;                                              \
    if (nt) {                                                                              \
      if (LDS) allow_lds(KERNEL_T, LDS, set_t);                                            \
      hipLaunchKernelGGL(KERNEL_T, dim3(grid), dim3(kThreads), LDS, st, __VA_ARGS__);     \
    } else {                                                                               \
      if (LDS) allow_lds(KERNEL_F, LDS, set_f);                                            \
      hipLaunchKernelGGL(KERNEL_F, dim3(grid), dim3(kThreads), LDS, st, __VA_ARGS__);     \
    }                                                                                      \
    return (int)hipGetLastError();                                                         \
  } while (0)

template <int R>
struct SpluLaunch {
  static constexpr int kLdsA2 = splu_lds_bytes<R, 3, R>();
  static constexpr int kLdsU2 = splu_lds_bytes<R, R + 4, 2 * R>();
  static constexpr int kLdsU3 = splu_lds_bytes<R, R + 4, R>();
  static constexpr int kLdsU4 = splu_lds_bytes<R, R + 4, 0>();

  static int u2dot(int nt, const float* U2, long ldu, const float* x, long n2, int head, float* part, int grid,
                   hipStream_t st) {
    SPLU_LAUNCH((k_splu_u2dot<R, true>), (k_splu_u2dot<R, false>), 0, U2, ldu, x, n2, head, part);
  }
  static int apply_s2(int nt, const float* L2s, const float* l3, const float* u3, const float* g2, float* qg2, long n2s,
                      int head, const float* coef, float* part, int grid, hipStream_t st) {
    SPLU_LAUNCH((k_splu_apply_s2<R, true>), (k_splu_apply_s2<R, false>), kLdsA2, L2s, l3, u3, g2, qg2, n2s, head, coef,
                part);
  }
  static int apply_s3(int nt, const float* U2, long ldu, const float* l3, const float* u3, float* out2, long n2, int head,
                      const float* coef, int grid, hipStream_t st) {
    SPLU_LAUNCH((k_splu_apply_s3<R, true>), (k_splu_apply_s3<R, false>), 0, U2, ldu, l3, u3, out2, n2, head, coef);
  }
  static int upd_s2(int nt, const float* L2s, const float* U2s, long ldu, const float* l3, const float* u3,
                    const float* x2, const float* g2, long n2s, int head, const float* coef, float* part, int grid,
                    hipStream_t st) {
    SPLU_LAUNCH((k_splu_upd_s2<R, true>), (k_splu_upd_s2<R, false>), kLdsU2, L2s, U2s, ldu, l3, u3, x2, g2, n2s, head,
                coef, part);
  }
  static int upd_s3(int nt, const float* L2s, const float* U2s, long ldu, const float* l3, const float* u3,
                    const float* g2, const float* x2, long n2s, int head, const float* coef, float* part, float* pmax,
                    int grid, hipStream_t st) {
    SPLU_LAUNCH((k_splu_upd_s3<R, true>), (k_splu_upd_s3<R, false>), kLdsU3, L2s, U2s, ldu, l3, u3, g2, x2, n2s, head,
                coef, part, pmax);
  }
  static int upd_s4(int nt, const float* L2s, const float* U2s, long ldu, const float* l3, const float* u3,
                    const float* g2, const float* x2, float* L2o, float* U2o, float* l3o, float* u3o, long n2s,
                    int head, const float* coef, int grid, hipStream_t st) {
    SPLU_LAUNCH((k_splu_upd_s4<R, true>), (k_splu_upd_s4<R, false>), kLdsU4, L2s, U2s, ldu, l3, u3, g2, x2, L2o, U2o,
                l3o, u3o, n2s, head, coef);
  }
  static int occupancy(int which) {
    const void* f = nullptr;
    int lds = 0;
    switch (which) {
      case 0: f = reinterpret_cast<const void*>(&k_splu_u2dot<R, true>); break;
      case 1: f = reinterpret_cast<const void*>(&k_splu_apply_s2<R, true>); lds = kLdsA2; break;
      case 2: f = reinterpret_cast<const void*>(&k_splu_apply_s3<R, true>); break;
      case 3: f = reinterpret_cast<const void*>(&k_splu_upd_s2<R, true>); lds = kLdsU2; break;
      case 4: f = reinterpret_cast<const void*>(&k_splu_upd_s3<R, true>); lds = kLdsU3; break;
      case 5: f = reinterpret_cast<const void*>(&k_splu_upd_s4<R, true>); lds = kLdsU4; break;
      default: return 0;
    }
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, kThreads, lds) != hipSuccess) return 0;
    return n;
  }
  static const SpluOps* ops() {
    static const SpluOps o = {Cfg<R>::kTileRows, kLdsA2, kLdsU2, kLdsU3, kLdsU4, &u2dot, &apply_s2, &apply_s3,
                              &upd_s2,           &upd_s3, &upd_s4, &occupancy};
    return &o;
  }
};

const SpluOps* PSGD_GROUP_FN(int r) {
  switch (r - PSGD_RANK_LO) {
    case 0: return SpluLaunch<PSGD_RANK_LO + 0>::ops();
    case 1: return SpluLaunch<PSGD_RANK_LO + 1>::ops();
    case 2: return SpluLaunch<PSGD_RANK_LO + 2>::ops();
    case 3: return SpluLaunch<PSGD_RANK_LO + 3>::ops();
    case 4: return SpluLaunch<PSGD_RANK_LO + 4>::ops();
    case 5: return SpluLaunch<PSGD_RANK_LO + 5>::ops();
    case 6: return SpluLaunch<PSGD_RANK_LO + 6>::ops();
    case 7: return SpluLaunch<PSGD_RANK_LO + 7>::ops();
    default: return nullptr;
  }
}

}  // namespace psgd
