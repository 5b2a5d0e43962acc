// uvd_wide_group.hip -- the UVd kernels of the wide-rank path (uvd_wide.py) for the 8 ranks PSGD_RANK_LO .. PSGD_RANK_LO + 7 of
// 33 .. 64: the four-column building blocks (colsums, axpy, rank-2 row update) and the three sweeps of the apply, on the whole
// [N, r] matrix.  Rounds 3-4 ran ranks above 32 on column chunks of width <= 32 (views with a row stride): a chunk sweep of a 40-wide
// matrix touches every 128-byte line of its rows, so r = 40 moved its bytes twice.  Compiled four times (33-40, 41-48, 49-56, 57-64).
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
struct WideLaunch {
  static int colreduce4(int nt, const float* M, const float* const* x, long N, double* part, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_colreduce4<R, true>), (k_colreduce4<R, false>), M, x[0], x[1], x[2], x[3], N, part);
  }
  static int rowdot_axpy4(int nt, const float* M, const float* const* x, float* const* o, int ncols, long N, const float* coef,
                          int grid, hipStream_t st) {
    PSGD_LAUNCH((k_rowdot_axpy4<R, true>), (k_rowdot_axpy4<R, false>), M, x[0], x[1], x[2], x[3], o[0], o[1], o[2], o[3], ncols, N, coef);
  }
  static int rank2_update(int nt, float* M, const float* a, const float* b, long N, const float* coef, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_rank2_update<R, true>), (k_rank2_update<R, false>), M, a, b, N, coef);
  }
  static int apply4_s1(int nt, const float* V, const float* d, const float* const* x, long N, double* part, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_apply4_s1<R, true>), (k_apply4_s1<R, false>), V, d, x[0], x[1], x[2], x[3], N, part);
  }
  static int apply4_s2(int nt, const float* U, const float* d, const float* const* x, float* const* o, int ncols, long N,
                       const float* coef, double* part, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_apply4_s2<R, true>), (k_apply4_s2<R, false>), U, d, x[0], x[1], x[2], x[3], o[0], o[1], o[2], o[3], ncols, N, coef, part);
  }
  static int apply4_s3(int nt, const float* V, const float* d, float* const* o, int ncols, long N, const float* coef, int grid,
                       hipStream_t st) {
    PSGD_LAUNCH((k_apply4_s3<R, true>), (k_apply4_s3<R, false>), V, d, o[0], o[1], o[2], o[3], ncols, N, coef);
  }
  // sweep 2 of the update (psgd.py:569-584, :600-601 / :614-615), the kernel of the specialised ranks on 64-row tiles of U and V;
  // g / part_pq non-null: the fused form that also reduces [Unew | Vnew]' [d.*g, d.*g.*nablaD] (one ColSum per operand)
  static int update_s2(int nt, int update_U, float* U, float* V, const float* d, const float* v, const float* h, const float* g, long N,
                       const float* coef, float* nabla, float* part_max, double* part_pq, int grid, hipStream_t st) {
    if (g) {
      if (update_U)
        PSGD_LAUNCH((k_update_s2<R, true, true, true>), (k_update_s2<R, true, false, true>), U, V, d, v, h, g, N, coef, nabla, part_max, part_pq);
      PSGD_LAUNCH((k_update_s2<R, false, true, true>), (k_update_s2<R, false, false, true>), U, V, d, v, h, g, N, coef, nabla, part_max, part_pq);
    }
    if (update_U)
      PSGD_LAUNCH((k_update_s2<R, true, true, false>), (k_update_s2<R, true, false, false>), U, V, d, v, h, g, N, coef, nabla, part_max, part_pq);
    PSGD_LAUNCH((k_update_s2<R, false, true, false>), (k_update_s2<R, false, false, false>), U, V, d, v, h, g, N, coef, nabla, part_max, part_pq);
  }
  // last sweep of the fused update -> apply (k_uvd_final)
  static int final_sweep(int nt, const float* U, const float* V, float* d, const float* nabla, const float* g, float* out, long N,
                         const float* coef, const float* maxbuf, float step, float tiny, int grid, hipStream_t st) {
    PSGD_LAUNCH((k_uvd_final<R, true>), (k_uvd_final<R, false>), U, V, d, nabla, g, out, N, coef, maxbuf, step, tiny);
  }
  static const UvdWideOps* ops() {
    static const UvdWideOps o = {Cfg<R>::kTileRows, &colreduce4, &rowdot_axpy4, &rank2_update, &apply4_s1, &apply4_s2, &apply4_s3,
                                 &update_s2, &final_sweep};
    return &o;
  }
};

const UvdWideOps* PSGD_GROUP_FN(int r) {
  switch (r - PSGD_RANK_LO) {
    case 0: return WideLaunch<PSGD_RANK_LO + 0>::ops();
    case 1: return WideLaunch<PSGD_RANK_LO + 1>::ops();
    case 2: return WideLaunch<PSGD_RANK_LO + 2>::ops();
    case 3: return WideLaunch<PSGD_RANK_LO + 3>::ops();
    case 4: return WideLaunch<PSGD_RANK_LO + 4>::ops();
    case 5: return WideLaunch<PSGD_RANK_LO + 5>::ops();
    case 6: return WideLaunch<PSGD_RANK_LO + 6>::ops();
    case 7: return WideLaunch<PSGD_RANK_LO + 7>::ops();
    default: return nullptr;
  }
}

}  // namespace psgd
