// uvd_rank_group.hip -- instantiates the rank-templated UVd kernels for the 8
// ranks PSGD_RANK_LO .. PSGD_RANK_LO+7 and exposes them through a table of host
// launchers.  Compiled four times (ranks 1-8, 9-16, 17-24, 25-32) so that the
// groups build in parallel.
#include "uvd_kernels.h"

#ifndef PSGD_RANK_LO
#error "compile with -DPSGD_RANK_LO=<first rank> -DPSGD_GROUP_FN=<symbol>"
#endif

namespace psgd {

#define PSGD_LAUNCH(KERNEL_T, KERNEL_F, ...)                                              \
  do {                                                                                    \
    if (nt)                                                                               \
      hipLaunchKernelGGL(KERNEL_T, dim3(grid), dim3(kThreads), 0, st, __VA_ARGS__);      \
    else                                                                                  \
      hipLaunchKernelGGL(KERNEL_F, dim3(grid), dim3(kThreads), 0, st, __VA_ARGS__);      \
    return (int)hipGetLastError();                                                        \
  } while (0)

template <int R>
struct Launch {
  static int colreduce(int nt, int nvec, const float* M, const float* a, const float* b, long N, float* part,
                       int grid, hipStream_t st) {
    if (nvec == 2) PSGD_LAUNCH((k_colreduce<R, 2, true>), (k_colreduce<R, 2, false>), M, a, b, N, part);
    PSGD_LAUNCH((k_colreduce<R, 1, true>), (k_colreduce<R, 1, false>), M, a, b, N, part);
  }
  static int apply_s2(int nt, const float* U, const float* d, const float* g, float* g1out, long N,
                      const float* coef, float* part, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_apply_s2<R, true>), (k_apply_s2<R, false>), U, d, g, g1out, N, coef, part);
  }
  static int apply_s3(int nt, const float* V, const float* d, float* out, long N, const float* coef, int grid,
                      hipStream_t st) {
    PSGD_LAUNCH((k_apply_s3<R, true>), (k_apply_s3<R, false>), V, d, out, N, coef);
  }
  static int rowdot_axpy(int nt, const float* M, const float* x, float* out, long N, const float* coef, int grid,
                         hipStream_t st) {
    PSGD_LAUNCH((k_rowdot_axpy<R, true>), (k_rowdot_axpy<R, false>), M, x, out, N, coef);
  }
  static int update_gram(int nt, const float* U, const float* V, const float* d, const float* v, const float* h,
                         long N, double* part, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_update_gram<R, true>), (k_update_gram<R, false>), U, V, d, v, h, N, part);
  }
  static int update_s2(int nt, int update_U, float* U, float* V, const float* d, const float* v, const float* h,
                       const float* g, long N, const float* coef, float* nabla, float* part_max, double* part_pq,
                       int grid, hipStream_t st) {
    if (g) {
      if (update_U)
        PSGD_LAUNCH((k_update_s2<R, true, true, true>), (k_update_s2<R, true, false, true>), U, V, d, v, h, g, N, coef,
                    nabla, part_max, part_pq);
      PSGD_LAUNCH((k_update_s2<R, false, true, true>), (k_update_s2<R, false, false, true>), U, V, d, v, h, g, N, coef,
                  nabla, part_max, part_pq);
    }
    if (update_U)
      PSGD_LAUNCH((k_update_s2<R, true, true, false>), (k_update_s2<R, true, false, false>), U, V, d, v, h, g, N, coef,
                  nabla, part_max, part_pq);
    PSGD_LAUNCH((k_update_s2<R, false, true, false>), (k_update_s2<R, false, false, false>), U, V, d, v, h, g, N, coef,
                nabla, part_max, part_pq);
  }
  static int colreduce4(int nt, const float* M, const float* const* x, long N, double* part, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_colreduce4<R, true>), (k_colreduce4<R, false>), M, x[0], x[1], x[2], x[3], N, part);
  }
  static int rowdot_axpy4(int nt, const float* M, const float* const* x, float* const* o, int ncols, long N,
                          const float* coef, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_rowdot_axpy4<R, true>), (k_rowdot_axpy4<R, false>), M, x[0], x[1], x[2], x[3], o[0], o[1], o[2], o[3],
                ncols, N, coef);
  }
  static int rank2_update(int nt, float* M, const float* a, const float* b, long N, const float* coef, int grid,
                          hipStream_t st) {
    PSGD_LAUNCH((k_rank2_update<R, true>), (k_rank2_update<R, false>), M, a, b, N, coef);
  }
  static int final_sweep(int nt, const float* U, const float* V, float* d, const float* nabla, const float* g, float* out,
                         long N, const float* coef, const float* maxbuf, float step, float tiny, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_uvd_final<R, true>), (k_uvd_final<R, false>), U, V, d, nabla, g, out, N, coef, maxbuf, step, tiny);
  }
  static int apply4_s1(int nt, const float* V, const float* d, const float* const* x, long N, double* part, int grid,
                       hipStream_t st) {
    PSGD_LAUNCH((k_apply4_s1<R, true>), (k_apply4_s1<R, false>), V, d, x[0], x[1], x[2], x[3], N, part);
  }
  static int apply4_s2(int nt, const float* U, const float* d, const float* const* x, float* const* o, int ncols, long N,
                       const float* coef, double* part, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_apply4_s2<R, true>), (k_apply4_s2<R, false>), U, d, x[0], x[1], x[2], x[3], o[0], o[1], o[2], o[3], ncols,
                N, coef, part);
  }
  static int apply4_s3(int nt, const float* V, const float* d, float* const* o, int ncols, long N, const float* coef,
                       int grid, hipStream_t st) {
    PSGD_LAUNCH((k_apply4_s3<R, true>), (k_apply4_s3<R, false>), V, d, o[0], o[1], o[2], o[3], ncols, N, coef);
  }
  static int update_gram_ld(const float* U, long ldU, const float* V, long ldV, const float* d, const float* v, const float* h,
                            long N, double* part, int grid, hipStream_t st) {
    hipLaunchKernelGGL((k_update_gram<R, false, true>), dim3(grid), dim3(kThreads), 0, st, U, V, d, v, h, N, part, RowStrides{{ldU, ldV}});
    return (int)hipGetLastError();
  }
  static int colreduce4_ld(const float* M, long ld, const float* const* x, long N, double* part, int grid, hipStream_t st) {
    hipLaunchKernelGGL((k_colreduce4<R, false, true>), dim3(grid), dim3(kThreads), 0, st, M, x[0], x[1], x[2], x[3], N, part, ld);
    return (int)hipGetLastError();
  }
  static int rowdot_axpy4_ld(const float* M, long ld, const float* const* x, float* const* o, int ncols, long N,
                             const float* coef, int grid, hipStream_t st) {
    hipLaunchKernelGGL((k_rowdot_axpy4<R, false, true>), dim3(grid), dim3(kThreads), 0, st, M, x[0], x[1], x[2], x[3], o[0], o[1],
                       o[2], o[3], ncols, N, coef, ld);
    return (int)hipGetLastError();
  }
  static int rank2_update_ld(float* M, long ld, const float* a, const float* b, long N, const float* coef, int grid,
                             hipStream_t st) {
    hipLaunchKernelGGL((k_rank2_update<R, false, true>), dim3(grid), dim3(kThreads), 0, st, M, a, b, N, coef, ld);
    return (int)hipGetLastError();
  }
  static int occupancy(int which) {
    const void* f = nullptr;
    switch (which) {
      case kOccColreduce: f = reinterpret_cast<const void*>(&k_colreduce<R, 2, true>); break;
      case kOccApplyS2: f = reinterpret_cast<const void*>(&k_apply_s2<R, true>); break;
      case kOccApplyS3: f = reinterpret_cast<const void*>(&k_apply_s3<R, true>); break;
      case kOccRowdot: f = reinterpret_cast<const void*>(&k_rowdot_axpy<R, true>); break;
      case kOccGram: f = reinterpret_cast<const void*>(&k_update_gram<R, true>); break;
      case kOccUpdS2U: f = reinterpret_cast<const void*>(&k_update_s2<R, true, true, false>); break;
      case kOccUpdS2V: f = reinterpret_cast<const void*>(&k_update_s2<R, false, true, false>); break;
      case kOccUpdS2F: f = reinterpret_cast<const void*>(&k_update_s2<R, false, true, true>); break;
      case kOccFinal: f = reinterpret_cast<const void*>(&k_uvd_final<R, true>); break;
      default: return 0;
    }
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, kThreads, 0) != hipSuccess) return 0;
    return n;
  }
  static const UvdOps* ops() {
    static const UvdOps o = {Cfg<R>::kTileRows, GramCfg<R>::kLen, &colreduce, &apply_s2, &apply_s3,
                             &rowdot_axpy,      &update_gram,     &update_s2, &colreduce4, &rowdot_axpy4,
                             &rank2_update,     &final_sweep,     &apply4_s1,  &apply4_s2,  &apply4_s3,  Cfg<R>::kVec,     &update_gram_ld, &colreduce4_ld,
                             &rowdot_axpy4_ld,  &rank2_update_ld, &occupancy};
    return &o;
  }
};

const UvdOps* PSGD_GROUP_FN(int r) {
  switch (r - PSGD_RANK_LO) {
    case 0: return Launch<PSGD_RANK_LO + 0>::ops();
    case 1: return Launch<PSGD_RANK_LO + 1>::ops();
    case 2: return Launch<PSGD_RANK_LO + 2>::ops();
    case 3: return Launch<PSGD_RANK_LO + 3>::ops();
    case 4: return Launch<PSGD_RANK_LO + 4>::ops();
    case 5: return Launch<PSGD_RANK_LO + 5>::ops();
    case 6: return Launch<PSGD_RANK_LO + 6>::ops();
    case 7: return Launch<PSGD_RANK_LO + 7>::ops();
    default: return nullptr;
  }
}

}  // namespace psgd
