// psgd_kron_bf16.hip -- bf16-operand / fp32-accumulate GEMM chain for the Kronecker
// dense (x) dense apply at Transformer scale (BASELINE config 5: 4096 x 4096 bf16 weight).
//
// The reference pins its Kron API to fp32 (psgd.py:113-115); this entry point is the
// extension SURVEY section 7 (hard part 5) describes: Ql, Qr stay fp32 master copies, they
// are rounded to bf16 once per call, the gradient arrives in bf16, every product of
// psgd.py:189-192 runs on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) with fp32
// accumulation, intermediates are kept in bf16, and the association order of the reference
// (branch on M < N) is preserved.
//
// One canonical GEMM: C[M,N] = A[M,K] * Bt[N,K]'  with both operands K-contiguous, so that
// every MFMA fragment (8 consecutive k of one row) is a single 16-byte LDS read.  All the
// transposes of the chain are absorbed by *which* copy of a factor is passed (Q or Q') and by
// an epilogue that can store C or C' in fp32 or bf16.  Upper-triangular factors restrict the
// K range of a tile (kmode), which removes about half of the MFMA work of the chain.
//
//   tile 128 x 128 x 64, 256 threads = 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4 MFMA tiles
//   LDS: 2 buffers x (A 16 KiB + B 16 KiB); 16-byte slots XOR-swizzled by (row & 7) so the
//        16 rows of a fragment read spread over 8 slots (2-way instead of 16-way conflict)
//   register-staged prefetch of tile k+1 while tile k is multiplied; one barrier per K tile
//   blockIdx -> tile map is XCD-aware (consecutive tiles of one XCD share A/B panels in its L2)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "psgd_hip.h"

namespace psgdh {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, TK = 64;
constexpr int kThreads = 256;
enum { KLO_M = 1, KHI_M = 2, KLO_N = 4, KHI_N = 8 };

struct HGemmArgs {
  const uint16_t* A; long lda;      // [M][K] bf16
  const uint16_t* B; long ldb;      // [N][K] bf16 (= B transposed)
  void* C; long ldc;
  int c_bf16, c_trans;              // element type and orientation of the stored result
  int M, N, K, kmode;
};

__device__ __forceinline__ uint16_t f2bf(float x) {
  // round-to-nearest-even; NaN stays NaN through the hardware conversion of a plain cast
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(x));
}

// blockIdx -> output tile.  Blocks b, b+8, ... run on one XCD (own 4 MiB L2).  When the tile grid is a
// multiple of 8 x 8, every XCD works through 8 x 8 tile PATCHES: the 64 tiles that run together share
// 8 A panels and 8 B panels, so a K step of the whole patch pulls 16 panel slices (256 KiB) through L2
// instead of 34 for a 2 x 32 strip -- with 64 flop/B per 128 x 128 tile the GEMM is otherwise bound by
// L2-miss (Infinity Cache) bandwidth, not by the MFMA rate.  Patches are dealt to XCDs so that the long
// and the short K ranges of the triangular modes are balanced; inside a patch the longest tiles go first.
__device__ __forceinline__ void hgemm_tile_coords(const HGemmArgs& g, int& m0, int& n0, int& klo, int& nk) {
  const int tiles_n = (g.N + TN - 1) / TN, tiles_m = (g.M + TM - 1) / TM;
  const int nt = tiles_m * tiles_n;
  int trow, tcol;
  if (tiles_m % 8 == 0 && tiles_n % 8 == 0 && nt % 512 == 0) {
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;      // j-th block of this XCD
    const int pm = tiles_m / 8, pn = tiles_n / 8;            // patch grid
    const int npatch = pm * pn, ppx = npatch / 8;            // patches per XCD
    const int pl = j / 64, e = j % 64;                       // local patch, element in patch
    // patches are dealt to the XCDs in serpentine order (0..7, 7..0, ...): bijective, and an XCD that got
    // a long-K patch row in one round gets a short one in the next (triangular modes)
    (void)ppx;
    const int pid = (pl & 1) ? pl * 8 + 7 - xcd : pl * 8 + xcd;
    const int prow = pid / pn, pcol = pid % pn;
    trow = prow * 8 + e / 8;
    tcol = pcol * 8 + e % 8;
    if (g.kmode & (KHI_M | KHI_N)) { trow = tiles_m - 1 - trow; tcol = tiles_n - 1 - tcol; }
  } else {
    int id = blockIdx.x;
    const int q = nt / 8, r = nt % 8, xcd = id % 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
    trow = id / tiles_n; tcol = id % tiles_n;
    if (g.kmode) {
      if (tiles_m % 8 == 0) { const int rpx = tiles_m / 8; trow = (trow % rpx) * 8 + trow / rpx; }
      if (g.kmode & (KHI_M | KHI_N)) { trow = tiles_m - 1 - trow; tcol = tiles_n - 1 - tcol; }
    }
  }
  m0 = trow * TM; n0 = tcol * TN;
  klo = 0;
  int khi = g.K;
  if (g.kmode & KLO_M) klo = max(klo, m0);
  if (g.kmode & KLO_N) klo = max(klo, n0);
  if (g.kmode & KHI_M) khi = min(khi, m0 + TM);
  if (g.kmode & KHI_N) khi = min(khi, n0 + TN);
  klo = (klo / TK) * TK;
  nk = (khi > klo) ? (khi - klo + TK - 1) / TK : 0;
}

__global__ __launch_bounds__(kThreads) void k_hgemm_nt(HGemmArgs g) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][2][TM * (TK / 8)];   // [buf][A|B][row*8 + slot]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;

  int m0, n0, klo, nk;
  hgemm_tile_coords(g, m0, n0, klo, nk);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  auto load_tile = [&](int kt) {
    const int k0 = klo + kt * TK;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = tid + kThreads * t, row = c >> 3, slot = c & 7;
      const int k = k0 + slot * 8;
      const int gm = m0 + row, gn = n0 + row;
      ra[t] = (gm < g.M && k < g.K) ? *reinterpret_cast<const u32x4*>(g.A + (long)gm * g.lda + k) : u32x4{0, 0, 0, 0};
      rb[t] = (gn < g.N && k < g.K) ? *reinterpret_cast<const u32x4*>(g.B + (long)gn * g.ldb + k) : u32x4{0, 0, 0, 0};
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = tid + kThreads * t, row = c >> 3, slot = c & 7;
      lds[buf][0][row * 8 + (slot ^ (row & 7))] = ra[t];
      lds[buf][1][row * 8 + (slot ^ (row & 7))] = rb[t];
    }
  };

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      bf16x8 a[4], b[4];
      const int slot = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + (lane & 15);
        a[i] = __builtin_bit_cast(bf16x8, lds[buf][0][row * 8 + (slot ^ (row & 7))]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + (lane & 15);
        b[j] = __builtin_bit_cast(bf16x8, lds[buf][1][row * 8 + (slot ^ (row & 7))]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: acc[i][j][e] is C[row = ..+(lane>>4)*4+e][col = ..+(lane&15)]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
      const int col = n0 + wn * 64 + j * 16 + (lane & 15);
      if (col >= g.N) continue;
      if (g.c_trans) {
        if (row0 + 3 < g.M) {
          if (g.c_bf16) {
            uint16_t* p = static_cast<uint16_t*>(g.C) + (long)col * g.ldc + row0;
            ushort4 v = make_ushort4(f2bf(acc[i][j][0]), f2bf(acc[i][j][1]), f2bf(acc[i][j][2]), f2bf(acc[i][j][3]));
            *reinterpret_cast<ushort4*>(p) = v;
          } else {
            float* p = static_cast<float*>(g.C) + (long)col * g.ldc + row0;
            *reinterpret_cast<f32x4*>(p) = acc[i][j];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (row0 + e < g.M) {
              if (g.c_bf16) static_cast<uint16_t*>(g.C)[(long)col * g.ldc + row0 + e] = f2bf(acc[i][j][e]);
              else static_cast<float*>(g.C)[(long)col * g.ldc + row0 + e] = acc[i][j][e];
            }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (row0 + e < g.M) {
            if (g.c_bf16) static_cast<uint16_t*>(g.C)[(long)(row0 + e) * g.ldc + col] = f2bf(acc[i][j][e]);
            else static_cast<float*>(g.C)[(long)(row0 + e) * g.ldc + col] = acc[i][j][e];
          }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (opt-in, psgd_kron_bf16_set_tuning(0, 2)) for full interior problems (M, N multiples of
// 128, K ranges multiples of 64).  Measured SLOWER than the register-staged kernel at 4096^2 (653 vs 549 us
// for the chain, profiles/r01_kron_bf16_variants.txt): with 64 flop/B per tile the GEMM is bound by
// L2-miss traffic, not by load latency, and the 128 KiB ring allows only one block per CU.  Kept as the
// starting point for a 256 x 256 tile.
// a ring of 4 stages (4 x 32 KiB, one __shared__ array) is filled by global_load_lds_dwordx4
// (16 B per lane, no VGPR round trip, no ds_write), three K tiles in flight across raw s_barriers
// with counted s_waitcnt vmcnt(N) -- a one-tile register prefetch only covers ~0.25 us of the
// ~1 us load latency per K tile.  The LDS image is the same swizzled layout as above; since the DMA
// writes linearly, the swizzle is applied to the per-lane SOURCE address (rule: linear destination,
// inverse-swizzled source, swizzled read).
constexpr int NS = 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ __launch_bounds__(kThreads) void k_hgemm_nt_dma(HGemmArgs g) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[NS][2][TM * (TK / 8)];   // the ONLY __shared__ object
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  int m0, n0, klo, nk;
  hgemm_tile_coords(g, m0, n0, klo, nk);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane source offsets of the 4 chunks this lane moves per operand per stage
  long offA[4], offB[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int p = (w * 4 + t) * 64 + lane;          // linear LDS position (16-byte units) within the tile
    const int row = p >> 3, slot = (p & 7) ^ (row & 7);
    offA[t] = (long)(m0 + row) * g.lda + slot * 8;
    offB[t] = (long)(n0 + row) * g.ldb + slot * 8;
  }
  auto issue = [&](int kt) {
    const int st = kt % NS;
    const long k0 = klo + (long)kt * TK;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g.A + offA[t] + k0), (lds_ptr_t)&lds[st][0][(w * 4 + t) * 64], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(g.B + offB[t] + k0), (lds_ptr_t)&lds[st][1][(w * 4 + t) * 64], 16, 0, 0);
    }
  };

  for (int kt = 0; kt < NS - 1 && kt < nk; ++kt) issue(kt);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed: allow the (up to two) younger stages to stay in flight
    const int younger = min(nk - 1 - kt, NS - 2);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave's DMA of stage kt is visible; stage kt-1 is free
    if (kt + NS - 1 < nk) issue(kt + NS - 1);
    const int st = kt % NS;
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      bf16x8 a[4], b[4];
      const int slot = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + (lane & 15);
        a[i] = __builtin_bit_cast(bf16x8, lds[st][0][row * 8 + (slot ^ (row & 7))]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + (lane & 15);
        b[j] = __builtin_bit_cast(bf16x8, lds[st][1][row * 8 + (slot ^ (row & 7))]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
      const int col = n0 + wn * 64 + j * 16 + (lane & 15);
      if (g.c_trans) {
        if (g.c_bf16) {
          uint16_t* p = static_cast<uint16_t*>(g.C) + (long)col * g.ldc + row0;
          *reinterpret_cast<ushort4*>(p) = make_ushort4(f2bf(acc[i][j][0]), f2bf(acc[i][j][1]), f2bf(acc[i][j][2]), f2bf(acc[i][j][3]));
        } else {
          *reinterpret_cast<f32x4*>(static_cast<float*>(g.C) + (long)col * g.ldc + row0) = acc[i][j];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (g.c_bf16) static_cast<uint16_t*>(g.C)[(long)(row0 + e) * g.ldc + col] = f2bf(acc[i][j][e]);
          else static_cast<float*>(g.C)[(long)(row0 + e) * g.ldc + col] = acc[i][j][e];
        }
      }
    }
}

// dst (bf16) = src or src', 64 x 64 tiles through LDS.  SRC_BF16 selects the source element type.
template <bool SRC_BF16>
__global__ __launch_bounds__(kThreads) void k_to_bf16(const void* src, long lds_, uint16_t* dst, long ldd, int rows,
                                                      int cols, int transpose) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += kThreads) {
    const int r = e >> 6, c = e & 63;
    uint16_t v = 0;
    if (r0 + r < rows && c0 + c < cols) {
      if (SRC_BF16) v = static_cast<const uint16_t*>(src)[(long)(r0 + r) * lds_ + c0 + c];
      else v = f2bf(static_cast<const float*>(src)[(long)(r0 + r) * lds_ + c0 + c]);
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += kThreads) {
    const int a = e >> 6, b = e & 63;
    if (transpose) {
      if (c0 + a < cols && r0 + b < rows) dst[(long)(c0 + a) * ldd + r0 + b] = tile[b][a];
    } else {
      if (r0 + a < rows && c0 + b < cols) dst[(long)(r0 + a) * ldd + c0 + b] = tile[a][b];
    }
  }
}

// dst = bf16(src) and dstT = bf16(src') in one pass over a square fp32 matrix (64 x 64 tiles through LDS)
__global__ __launch_bounds__(kThreads) void k_to_bf16_both(const float* src, uint16_t* dst, uint16_t* dstT, int n) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += kThreads) {
    const int r = e >> 6, c = e & 63;
    uint16_t v = 0;
    if (r0 + r < n && c0 + c < n) {
      v = f2bf(src[(long)(r0 + r) * n + c0 + c]);
      dst[(long)(r0 + r) * n + c0 + c] = v;
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += kThreads) {
    const int a = e >> 6, b = e & 63;
    if (c0 + a < n && r0 + b < n) dstT[(long)(c0 + a) * n + r0 + b] = tile[b][a];
  }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

struct HWs {
  uint16_t *Ql, *QlT, *Qr, *QrT, *GT, *T1, *T2, *T3;
  int64_t total;
};

static HWs hws_layout(char* base, int M, int N) {
  HWs k;
  const int64_t mm = (int64_t)M * M * 2, nn = (int64_t)N * N * 2, mn = (int64_t)M * N * 2;
  const int64_t sq = mm > nn ? mm : nn;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { uint16_t* p = reinterpret_cast<uint16_t*>(base + off); off = align256(off + bytes); return p; };
  k.Ql = take(mm); k.QlT = take(mm); k.Qr = take(nn); k.QrT = take(nn);
  k.GT = take(mn); k.T1 = take(sq); k.T2 = take(mn); k.T3 = take(mn);
  k.total = off;
  return k;
}

static int g_hgemm_variant = 0;   // 0 / 1: register-staged double buffer (2 blocks/CU; measured faster), 2: LDS-DMA ring

static int launch_hgemm(const uint16_t* A, long lda, const uint16_t* B, long ldb, void* C, long ldc, int c_bf16,
                        int c_trans, int M, int N, int K, int kmode, hipStream_t st) {
  HGemmArgs g = {A, lda, B, ldb, C, ldc, c_bf16, c_trans, M, N, K, kmode};
  const int nt = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  const bool interior = (M % TM == 0) && (N % TN == 0) && (K % TM == 0) && (lda % 8 == 0) && (ldb % 8 == 0) &&
                        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  if (g_hgemm_variant == 2 && interior) hipLaunchKernelGGL(k_hgemm_nt_dma, dim3(nt), dim3(kThreads), 0, st, g);
  else hipLaunchKernelGGL(k_hgemm_nt, dim3(nt), dim3(kThreads), 0, st, g);
  return (int)hipGetLastError();
}

static int launch_cvt(const void* src, int src_bf16, long lds_, uint16_t* dst, long ldd, int rows, int cols,
                      int transpose, hipStream_t st) {
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (src_bf16) hipLaunchKernelGGL((k_to_bf16<true>), grid, dim3(kThreads), 0, st, src, lds_, dst, ldd, rows, cols, transpose);
  else hipLaunchKernelGGL((k_to_bf16<false>), grid, dim3(kThreads), 0, st, src, lds_, dst, ldd, rows, cols, transpose);
  return (int)hipGetLastError();
}

}  // namespace psgdh

using namespace psgdh;

#define HK(expr)                              \
  do {                                        \
    if ((expr) != 0) return PSGD_ERR_LAUNCH;  \
  } while (0)

extern "C" {

int psgd_kron_bf16_set_tuning(int key, int value) {
  if (key == 0) { g_hgemm_variant = value; return PSGD_OK; }
  return PSGD_ERR_BAD_ARG;
}

int64_t psgd_kron_dd_workspace_bytes_bf16(int M, int N) {
  if (M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  return hws_layout(nullptr, M, N).total;
}

int psgd_kron_dd_apply_bf16(const float* Ql, const float* Qr, const void* G, void* out, int M, int N, void* ws,
                            int64_t ws_bytes, void* stream) {
  if (!Ql || !Qr || !G || !out) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0 || (M % 8) || (N % 8)) return PSGD_ERR_SHAPE;   // 16-byte bf16 chunks along K
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < hws_layout(nullptr, M, N).total)
    return PSGD_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(G) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  HWs k = hws_layout(static_cast<char*>(ws), M, N);
  const uint16_t* Gb = static_cast<const uint16_t*>(G);
  if (M < N) {                                                                     // psgd.py:189-190
    HK(launch_cvt(Ql, 0, M, k.QlT, M, M, M, 1, st));
    hipLaunchKernelGGL(k_to_bf16_both, dim3((N + 63) / 64, (N + 63) / 64), dim3(kThreads), 0, st, Qr, k.Qr, k.QrT, N);
    HK((int)hipGetLastError());
    HK(launch_cvt(G, 1, N, k.GT, M, M, N, 1, st));
    // T1 = Ql'Ql              A = Ql' [M][K=M], Bt = Ql' ; k <= min(m, n)
    HK(launch_hgemm(k.QlT, M, k.QlT, M, k.T1, M, 1, 0, M, M, M, KHI_M | KHI_N, st));
    // T2 = T1 G               A = T1 [M][K=M], Bt = G' [N][M]
    HK(launch_hgemm(k.T1, M, k.GT, M, k.T2, N, 1, 0, M, N, M, 0, st));
    // T3 = T2 Qr'             A = T2 [M][K=N], Bt[n][k] = Qr'[k][n] = Qr[n][k] ; k >= n
    HK(launch_hgemm(k.T2, N, k.Qr, N, k.T3, N, 1, 0, M, N, N, KLO_N, st));
    // out = T3 Qr             Bt[n][k] = Qr[k][n] = Qr'[n][k] ; k <= n
    HK(launch_hgemm(k.T3, N, k.QrT, N, out, N, 1, 0, M, N, N, KHI_N, st));
  } else {                                                                         // psgd.py:191-192
    HK(launch_cvt(Qr, 0, N, k.QrT, N, N, N, 1, st));
    hipLaunchKernelGGL(k_to_bf16_both, dim3((M + 63) / 64, (M + 63) / 64), dim3(kThreads), 0, st, Ql, k.Ql, k.QlT, M);
    HK((int)hipGetLastError());
    // T1 = Qr'Qr  (symmetric, so it is its own Bt layout)
    HK(launch_hgemm(k.QrT, N, k.QrT, N, k.T1, N, 1, 0, N, N, N, KHI_M | KHI_N, st));
    // T2 = G T1               A = G [M][K=N], Bt = T1' = T1 ; stored transposed: T2' [N][M]
    HK(launch_hgemm(Gb, N, k.T1, N, k.T2, M, 1, 1, M, N, N, 0, st));
    // T3 = Ql T2              A = Ql [M][K=M] (k >= m), Bt = T2' ; stored transposed: T3' [N][M]
    HK(launch_hgemm(k.Ql, M, k.T2, M, k.T3, M, 1, 1, M, N, M, KLO_M, st));
    // out = Ql' T3            A = Ql' [M][K=M] (k <= m), Bt = T3'
    HK(launch_hgemm(k.QlT, M, k.T3, M, out, N, 1, 0, M, N, M, KHI_M, st));
  }
  return PSGD_OK;
}

}  // extern "C"
