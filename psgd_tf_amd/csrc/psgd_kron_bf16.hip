// psgd_kron_bf16.hip -- bf16-operand / fp32-accumulate GEMM chain for the Kronecker
// dense (x) dense apply at Transformer scale (BASELINE config 5: 4096 x 4096 bf16 weight).
//
// The reference pins its Kron API to fp32 (psgd.py:113-115); this entry point is the
// extension SURVEY section 7 (hard part 5) describes: Ql, Qr stay fp32 master copies, they
// are rounded to bf16 once per call, the gradient arrives in bf16, every product of
// psgd.py:189-192 runs on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) with fp32
// accumulation and intermediates are kept in bf16.  The staged chain keeps the reference's
// association order (branch on M < N); where both factors qualify the default re-associates the
// Gram into the chain, out = Ql'(Ql((G Qr')Qr)), as two fused triangular pairs (see below).
//
// One canonical GEMM: C[M,N] = A[M,K] * Bt[N,K]'  with both operands K-contiguous, so that
// every MFMA fragment (8 consecutive k of one row) is a single 16-byte LDS read.  All the
// transposes of the chain are absorbed by *which* copy of a factor is passed (Q or Q') and by
// an epilogue that can store C or C' in fp32 or bf16.  Upper-triangular factors restrict the
// K range of a tile (kmode), which removes about half of the MFMA work of the chain.
//
// Kernels in this file:
//   k_hgemm_nt            128 x 128 x 64 tile, 4 waves, register-staged double buffer, XOR-swizzled LDS, XCD-aware
//                         8 x 8 tile patches; general shapes, triangular K ranges, symmetric mode (Q'Q: upper tiles
//                         computed, stored twice, longest-K-first with long/short pairing per CU)
//   k_hgemm_nt_dma        the same tile on a 4-stage LDS-DMA ring (opt-in; measured slower)
//   k_hgemm_nt_256        256 x 256 x 64 tile, 8 waves, 8-phase LDS-DMA schedule (counted vmcnt, raw barriers,
//                         staggered wave rows) for large dense products: 1.2 PFLOP/s
//   k_hgemm_tri_pair_256  two triangular products (Q X, then Q'(Q X)) fused into one persistent, wavefront-scheduled
//                         launch with in-launch tile hand-offs; the default apply is two of these and no Gram
//   k_factors_to_bf16     fp32 -> bf16 copies (plain and transposed) of both factors, upper 256-blocks only, one launch
//   k_to_bf16             generic convert / transpose (gradient operand of the staged chain)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "psgd_hip.h"
#include "kron_shared.h"
#include "nanmax.h"
using psgd::amaxf;
using psgd::nmaxf;

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
  int sym;                          // 1: C is symmetric (A == B): tiles below the diagonal are skipped, the others stored twice
                                    // 2: C is upper triangular (k_hgemm_nt, HEPI_D_MINUS): upper tiles only, C = D below
  // k_hgemm_nt only (the bf16-operand update):
  int epi;                          // HEPI_STORE | HEPI_TRIU_MAX (C = triu(acc), max|C| -> maxout, no mirror store) |
                                    // HEPI_D_MINUS (C = D - step / (*scale_max + tiny) * acc, fp32)
  int kflip;                        // > 0: the accumulators change sign before K tile `kflip` (C = -A1 B1' + A2 B2' over a
                                    // concatenated K axis)
  float* maxout;
  const float* D; long ldd;
  const float* scale_max; float step, tiny;
  int patch_order;                  // sym == 2: the tiles in 4 x 4 patches (tuning key 5)
  int lower_zero;                   // sym == 2, HEPI_D_MINUS: zeros below the diagonal instead of copies of D (D is not read there:
                                    // kron_balance_planes leaves it unwritten)
};
enum { HEPI_STORE = 0, HEPI_TRIU_MAX = 1, HEPI_D_MINUS = 2 };
__device__ __forceinline__ bool g_pair_patch_dev(const HGemmArgs& g) { return g.patch_order != 0; }

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
__device__ __forceinline__ void hgemm_tile_coords(const HGemmArgs& g, int bid, int& m0, int& n0, int& klo, int& nk) {
  const int tiles_n = (g.N + TN - 1) / TN, tiles_m = (g.M + TM - 1) / TM;
  const int nt = tiles_m * tiles_n;
  int trow, tcol;
  if (g.sym == 2) {
    // upper-triangular OUTPUT with a K range that grows with the distance from the diagonal (triu * triu: KLO_M | KHI_N,
    // tile (r, c) has c - r + 1 K chunks): only the T (T + 1) / 2 tiles on or above the diagonal are launched (each also
    // copies D into its mirror tile below the diagonal).  T % 16 == 0: XCD x (blocks x, x + 8, ...) works through the
    // tile rows 16 g + x and 16 g + 15 - x, every row from its longest tile (c = T - 1) down -- equal tile counts and
    // within 8 % equal K chunks per XCD, the A panel of a row shared in that XCD's L2.  (8 x 8 patches in launch order
    // put whole far-from-diagonal patches on one XCD: 2.1x imbalance, 150 us for 23 GFLOP at 4096.)
    const int T = tiles_m;
    if (g_pair_patch_dev(g) && T % 4 == 0 && T >= 32) {
      // 4 x 4 tile patches (psgd_kron.hip gemm_tile_from_id case 3'): the patches by decreasing distance from the diagonal, the tiles
      // of a patch by decreasing K length, dealt to the XCDs in runs of 16, serpentine per round of 128
      const int P = T >> 2, ntu = T * (T + 1) / 2, full = ntu & ~127, x = bid & 7, j = bid >> 3;
      int pos = bid;
      if (j < (full >> 3)) {
        const int m = j >> 4, xx = (m & 1) ? 7 - x : x;
        pos = m * 128 + 16 * xx + (j & 15);
      }
      const int noff = 16 * (P * (P - 1) / 2);
      const unsigned long long ord = 0xcd8e94fa50b61723ULL;      // nibble t = (r' << 2) | c' of a patch's t-th tile
      int pr, pc, t;
      if (pos < noff) {
        const int sidx = pos >> 4;
        t = pos & 15;
        int kk = (int)((sqrtf(8.0f * sidx + 1.0f) - 1.0f) * 0.5f);
        while (kk * (kk + 1) / 2 > sidx) --kk;
        while ((kk + 1) * (kk + 2) / 2 <= sidx) ++kk;
        pr = sidx - kk * (kk + 1) / 2; pc = pr + (P - 1 - kk);
      } else {
        const int u = pos - noff;
        pr = pc = u / 10; t = u % 10;
      }
      const int nib = (int)((ord >> (4 * t)) & 15ULL);
      trow = 4 * pr + (nib >> 2); tcol = 4 * pc + (nib & 3);
    } else if (T % 16 == 0) {
      const int x = bid & 7;
      int j = bid >> 3;
      trow = 0; tcol = 0;
      for (int grp = 0; grp < T / 16; ++grp) {
        const int ra = 16 * grp + x, rb = 16 * grp + 15 - x;
        if (j < T - ra) { trow = ra; tcol = T - 1 - j; break; }
        j -= T - ra;
        if (j < T - rb) { trow = rb; tcol = T - 1 - j; break; }
        j -= T - rb;
      }
    } else {                      // rows from the top, each from its longest tile down
      const int ntu = T * (T + 1) / 2, idx = ntu - 1 - bid;
      int sdiag = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
      while (sdiag * (sdiag + 1) / 2 > idx) --sdiag;
      while ((sdiag + 1) * (sdiag + 2) / 2 <= idx) ++sdiag;
      trow = T - 1 - sdiag;
      tcol = trow + idx - sdiag * (sdiag + 1) / 2;
    }
  } else if (g.sym) {
    // symmetric product (M == N, kmode = KHI_M | KHI_N): only the T (T + 1) / 2 tiles on or above the diagonal are
    // launched.  L = those tiles sorted by K length (row index) descending.  The first 256 blocks (one per CU) take
    // L[0..255]; the second 256 take the NEXT 256 entries in ascending order, so that the CU holding the longest
    // tile gets the shortest companion (two blocks are resident per CU); the rest follow in order.
    const int T = tiles_m, ntu = T * (T + 1) / 2, W = 256;
    int idx = bid;
    if (idx >= W && idx < 2 * W) { const int hi = min(ntu, 2 * W); idx = hi - 1 - (idx - W); }
    int sdiag = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
    while (sdiag * (sdiag + 1) / 2 > idx) --sdiag;
    while ((sdiag + 1) * (sdiag + 2) / 2 <= idx) ++sdiag;
    trow = T - 1 - sdiag;
    tcol = trow + idx - sdiag * (sdiag + 1) / 2;
  } else if (tiles_m % 8 == 0 && tiles_n % 8 == 0 && nt % 512 == 0) {
    const int xcd = bid % 8, j = bid / 8;      // j-th block of this XCD
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
    int id = bid;
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

__device__ __forceinline__ void hgemm_nt_body(const HGemmArgs& g, int bid, u32x4 (*lds)[2][TM * (TK / 8)]) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;

  int m0, n0, klo, nk;
  hgemm_tile_coords(g, bid, m0, n0, klo, nk);

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
    if (kt == g.kflip && kt > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = -acc[i][j];
    }
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
  float vmax = 0.0f;
  const float dscale = (g.epi == HEPI_D_MINUS) ? g.step / (*g.scale_max + g.tiny) : 0.0f;
  if (g.epi == HEPI_D_MINUS) {
    // all D elements of the tile are requested before the first one is used (clamped addresses): loaded one by one
    // inside the store loop they cost a full memory latency each
    f32x4 dv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = min(m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + e, g.M - 1);
          const int col = min(n0 + wn * 64 + j * 16 + (lane & 15), g.N - 1);
          dv[i][j][e] = g.D[(long)row * g.ldd + col];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = dv[i][j] - dscale * acc[i][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
      const int col = n0 + wn * 64 + j * 16 + (lane & 15);
      if (col >= g.N) continue;
      if (g.epi == HEPI_TRIU_MAX) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (col >= row0 + e && row0 + e < g.M) ? acc[i][j][e] : 0.0f;
          vmax = amaxf(vmax, fabsf(v));
          acc[i][j][e] = v;
        }
      }
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
        if (g.sym && n0 > m0 && g.epi == HEPI_STORE) {   // mirror: C[col][row0 .. row0+3] (M == N, multiples of 8 by contract)
          if (g.c_bf16) {
            uint16_t* p = static_cast<uint16_t*>(g.C) + (long)col * g.ldc + row0;
            if (row0 + 3 < g.M) {
              *reinterpret_cast<ushort4*>(p) = make_ushort4(f2bf(acc[i][j][0]), f2bf(acc[i][j][1]), f2bf(acc[i][j][2]), f2bf(acc[i][j][3]));
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (row0 + e < g.M) p[e] = f2bf(acc[i][j][e]);
            }
          } else {
            float* p = static_cast<float*>(g.C) + (long)col * g.ldc + row0;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (row0 + e < g.M) p[e] = acc[i][j][e];
          }
        }
      }
    }
  if (g.epi == HEPI_TRIU_MAX) {     // max |triu(C)| of the whole product: one atomic per wave (values are >= 0: int order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
    if (lane == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(g.maxout)) atomicMax(reinterpret_cast<int*>(g.maxout), __float_as_int(vmax));      // (looks first: atomics on one address serialise in L2)
  }
  if (g.sym == 2 && n0 > m0 && g.epi == HEPI_D_MINUS) {
    // the tile below the diagonal is not launched: its product is zero, C = D there (M == N, multiples of 8; fp32)
    const float* D = g.D;
    float* C = static_cast<float*>(g.C);
#pragma unroll 4
    for (int k = 0; k < TM * TN / 4 / kThreads; ++k) {
      const int idx = tid + kThreads * k, row = n0 + (idx >> 5), col = m0 + 4 * (idx & 31);
      if (row < g.M && col < g.N)
        *reinterpret_cast<f32x4*>(C + (long)row * g.ldc + col) =
            g.lower_zero ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(D + (long)row * g.ldd + col);
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_hgemm_nt(HGemmArgs g) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][2][TM * (TK / 8)];   // [buf][A|B][row*8 + slot]
  hgemm_nt_body(g, blockIdx.x, lds);
}

// Two independent products in one launch (the two gradients / the two factor updates of the bf16-operand Kron update):
// their blocks are interleaved in groups of 8 (one per XCD), so both start with their longest tiles and the leftover
// tiles of one fill the CUs the other has left (2 x 528 equal tiles on 512 slots: 2.06 rounds instead of 2 x 1.03 -> 2 x 2).
__global__ __launch_bounds__(kThreads) void k_hgemm_nt_two(HGemmArgs g0, HGemmArgs g1, int n0, int n1) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][2][TM * (TK / 8)];
  const int m = (n0 < n1 ? n0 : n1) & ~7, b = blockIdx.x;
  if (b < 2 * m) {
    const int local = (b >> 4) * 8 + (b & 7);
    if ((b >> 3) & 1) hgemm_nt_body(g1, local, lds); else hgemm_nt_body(g0, local, lds);
  } else {
    const int rest = b - 2 * m;
    if (rest < n0 - m) hgemm_nt_body(g0, m + rest, lds); else hgemm_nt_body(g1, m + rest - (n0 - m), lds);
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
  hgemm_tile_coords(g, blockIdx.x, m0, n0, klo, nk);

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

// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4, wave tile 128 x 64), 128 KiB of LDS, for dense interior problems
// (M, N multiples of 256, K ranges multiples of 64).  A 128 x 128 tile needs 34 TB/s of L2->CU traffic to keep
// the matrix cores busy and tops out at ~0.75 PFLOP/s; this tile halves the traffic per flop and hides the rest
// behind a finer-grained schedule (after the "8-phase" structure of cdna_hip_programming.md section 5):
//
//   * a K tile is staged as four HALF-TILES of 128 rows x 64 k (16 KiB): B-h0, A-h0, B-h1, A-h1, where A-h{q}
//     holds the q-th 64-row half of BOTH wave rows and B-h{q} the q-th 32-column half of all four wave
//     columns.  Every half-tile is therefore consumed (copied into fragment registers) in exactly one phase;
//   * a K tile is computed in four phases, one 64 x 32 quadrant of the wave tile each (16 MFMAs):
//       ph0 (m0,n0): reads B-h0, A-h0   ph1 (m0,n1): reads B-h1   ph2 (m1,n1): reads A-h1   ph3 (m1,n0): no reads
//   * the LDS holds 8 half-tile slots (two K tiles).  Phase q issues the LDS-DMA of half-tile q + 7, i.e. the
//     slot a half-tile occupied is refilled one (B-h0) or two phases after it was read, and the loads run
//     a full K tile plus three half-tiles ahead of the math.  One counted wait per K tile (vmcnt(6) in ph3)
//     retires the whole next K tile while the three youngest half-tiles stay in flight;
//   * the two wave rows run staggered by one barrier: while one group issues its fragment reads and DMA, the
//     other runs its 16 MFMAs, so the LDS pipe and the matrix cores of a SIMD (one wave of each group) overlap.
//
// Hazards (# = global barrier count; a wave of group g passes #(2q+1+g) and #(2q+2+g) around its MFMAs of phase q):
//   RAW  DMA -> ds_read: the wait sits in ph3 before the first barrier of that phase, the reads of the next K
//        tile start in the following phase, i.e. after every wave has waited and passed a barrier;
//   WAR  ds_read -> DMA: B-h0 is refilled one phase after it is read, by a wave that may be one barrier ahead:
//        the reader retires its four B reads (issued first) with lgkmcnt(8) BEFORE its first barrier of ph0.
//        All other slots are refilled two phases after their read, when the reader has passed its MFMAs.
constexpr int T2 = 256;
constexpr int kThreads2 = 512;

__device__ __forceinline__ void hgemm256_tile_coords(const HGemmArgs& g, int& m0, int& n0, int& klo, int& nk) {
  const int tiles_m = g.M / T2, tiles_n = g.N / T2, nt = tiles_m * tiles_n;
  int trow, tcol;
  if (tiles_m % 4 == 0 && tiles_n % 8 == 0 && nt % 256 == 0) {
    // 4 x 8 tile patches per XCD (32 CUs): one K step of a patch pulls 4 A panels and 8 B panels through its L2
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
    const int pn = tiles_n / 8;
    const int pid = (j / 32) * 8 + xcd, e = j % 32;
    trow = (pid / pn) * 4 + e / 8;
    tcol = (pid % pn) * 8 + e % 8;
  } else {
    int id = blockIdx.x;
    const int q = nt / 8, r = nt % 8, xcd = id % 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
    trow = id / tiles_n; tcol = id % tiles_n;
  }
  m0 = trow * T2; n0 = tcol * T2;
  klo = 0;
  int khi = g.K;
  if (g.kmode & KLO_M) klo = max(klo, m0);
  if (g.kmode & KLO_N) klo = max(klo, n0);
  if (g.kmode & KHI_M) khi = min(khi, m0 + T2);
  if (g.kmode & KHI_N) khi = min(khi, n0 + T2);
  klo = (klo / TK) * TK;
  nk = (khi > klo) ? (khi - klo + TK - 1) / TK : 0;
}

#define HG_FENCE() asm volatile("" ::: "memory")

template <int N> struct IntC { static constexpr int value = N; };

// The 8-phase main loop over `nk` K tiles.  src(n, t) = global address of the 16 bytes this lane moves with DMA
// instruction t (0, 1) of half-tile n (n = 4 * sequence index of the K tile + {0 B-h0, 1 A-h0, 2 B-h1, 3 A-h1});
// pre(n) runs (uniformly, all waves) right before half-tile n is issued inside the loop.  Self-contained: the
// stagger barrier of the second wave row at entry is matched at exit.
// DMAPOS: where a phase issues its two LDS-DMA pieces.  0 (the template as published): in the load half of the phase,
// after the fragment reads and before the phase's first barrier -- there a piece costs the issuing wave 100-185 cycles
// (MI355X_MICROARCH.md cycle constants) and the load half, not the 16 MFMAs of the other wave row, sets the length of
// a barrier interval.  1: both pieces inside the MFMA cluster (after MFMA 4 and MFMA 10), where a piece costs ~60 cycles
// of the wave's issue; the load half then holds only the fragment reads.  2: first piece as 0, second inside the cluster.
// The counted wait of phase 3 follows the placement (pieces of THIS phase are not yet issued at the wait for 1).
#ifndef HG_DBG
#define HG_DBG 0      // what-if builds (wrong results, timing only): 1 = no DMA after the prologue, 2 = every fragment read from one LDS
#endif                // address, 4 = no barriers inside the phases
template <int DMAPOS, class Src, class Pre, class Top>
__device__ __forceinline__ void hg256_mainloop(f32x4 (&acc)[8][4], u32x4* lds, int nk, int w, int lane, Src&& src,
                                               Pre&& pre, Top&& top) {      // top(kt): every wave, before the math of K tile kt
  const int wr = w >> 2, wc = w & 3;
  auto issue1 = [&](int n, int t) {
    const int slot = ((n >> 2) & 1) * 4 + (n & 3);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src(n, t), (lds_ptr_t)&lds[slot * 1024 + (t * 8 + w) * 64], 16, 0, 0);
  };
  auto issue = [&](int n) {
    issue1(n, 0);
    issue1(n, 1);
  };
  // fragment read positions (16-byte units inside a slot): row*8 + (chunk ^ (row & 7)), chunk = ks*4 + (lane>>4)
  const int c0 = (lane >> 4) ^ (lane & 7);
  const int rowA = (wr * 64 + (lane & 15)) * 8, rowB = (wc * 32 + (lane & 15)) * 8;
  const int posA[2] = {(HG_DBG & 2) ? 0 : rowA + c0, (HG_DBG & 2) ? 0 : rowA + (c0 ^ 4)};
  const int posB[2] = {(HG_DBG & 2) ? 0 : rowB + c0, (HG_DBG & 2) ? 0 : rowB + (c0 ^ 4)};

  bf16x8 a[4][2], b0[2][2], b1[2][2];
  const int total = 4 * nk;

  // prologue: half-tiles 0..6 in flight, K tile 0 landed
#pragma unroll
  for (int n = 0; n < 7; ++n)
    if (n < total) issue(n);
  if (nk >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HG_FENCE();
  __builtin_amdgcn_s_barrier();
  HG_FENCE();
  if (wr == 1) __builtin_amdgcn_s_barrier();            // stagger the second wave row by one barrier
  HG_FENCE();

  auto phase = [&](auto PHc, int kt) {
    constexpr int PH = decltype(PHc)::value;
    const u32x4* base = lds + (kt & 1) * 4096;
    // ---- fragment reads of this phase
    if constexpr (PH == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b0[j][ks] = __builtin_bit_cast(bf16x8, base[0 * 1024 + posB[ks] + j * 128]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[i][ks] = __builtin_bit_cast(bf16x8, base[1 * 1024 + posA[ks] + i * 128]);
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (PH == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) b1[j][ks] = __builtin_bit_cast(bf16x8, base[2 * 1024 + posB[ks] + j * 128]);
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (PH == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[i][ks] = __builtin_bit_cast(bf16x8, base[3 * 1024 + posA[ks] + i * 128]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- DMA of half-tile q + 7
    const int n = 4 * kt + PH + 7;
    if constexpr (DMAPOS == 0) {
      if (n < total) {
        pre(n);
        if (!(HG_DBG & 1)) issue(n);
      }
    } else if constexpr (DMAPOS == 2) {
      if (n < total) {
        pre(n);
        issue1(n, 0);
      }
    }
    if constexpr (PH == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the four B-h0 reads are done
    if constexpr (PH == 3) {
      // retire K tile kt + 1: everything but the pieces issued after its last half-tile -- 3 half-tiles (6 pieces) when this
      // phase's own pieces are already out, 2 half-tiles + none (4) / + one piece (5) of this phase otherwise
      if (kt + 2 < nk) {
        if constexpr (DMAPOS == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (DMAPOS == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      } else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    HG_FENCE();
    if (!(HG_DBG & 4)) __builtin_amdgcn_s_barrier();
    HG_FENCE();
    // ---- 16 MFMAs: one 64 x 32 quadrant x K = 64
    constexpr int MQ = (PH >= 2) ? 1 : 0;
    constexpr int NQ = (PH == 1 || PH == 2) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
    int mcount = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 bv = NQ ? b1[j][ks] : b0[j][ks];
          acc[4 * MQ + i][2 * NQ + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bv, acc[4 * MQ + i][2 * NQ + j], 0, 0, 0);
          ++mcount;                                          // (compile-time after unrolling)
          if constexpr (DMAPOS == 1) {
            if (mcount == 4 && n < total) {
              __builtin_amdgcn_sched_barrier(0);
              pre(n);
              issue1(n, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if constexpr (DMAPOS != 0) {
            if (mcount == 10 && n < total) {
              __builtin_amdgcn_sched_barrier(0);
              issue1(n, 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
    __builtin_amdgcn_s_setprio(0);
    HG_FENCE();
    if (!(HG_DBG & 4)) __builtin_amdgcn_s_barrier();
    HG_FENCE();
  };

  for (int kt = 0; kt < nk; ++kt) {
    top(kt);
    phase(IntC<0>{}, kt);
    phase(IntC<1>{}, kt);
    phase(IntC<2>{}, kt);
    phase(IntC<3>{}, kt);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();            // matches the stagger barrier of the other wave row
  HG_FENCE();
}

template <int DMAPOS, class Src, class Pre>
__device__ __forceinline__ void hg256_mainloop(f32x4 (&acc)[8][4], u32x4* lds, int nk, int w, int lane, Src&& src, Pre&& pre) {
  hg256_mainloop<DMAPOS>(acc, lds, nk, w, lane, src, pre, [](int) {});
}

// per-lane element offsets of the two DMA instructions of each half type inside an operand tile whose row 0 is
// `row0` (A: tile rows, B: tile columns).  The DMA writes linearly (wave base + 16 * lane), so the XOR swizzle of the
// LDS image is applied to the per-lane SOURCE address.
__device__ __forceinline__ void hg256_offsets(int w, int lane, long lda, long ldb, long (&offA)[2][2], long (&offB)[2][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int p = (t * 8 + w) * 64 + lane;            // 16-byte position within the half-tile
    const int lr = p >> 3, chunk = (p & 7) ^ (lr & 7);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      offA[q][t] = (long)((lr >> 6) * 128 + q * 64 + (lr & 63)) * lda + chunk * 8;
      offB[q][t] = (long)((lr >> 5) * 64 + q * 32 + (lr & 31)) * ldb + chunk * 8;
    }
  }
}

// acc[i][j][e] is C[m0 + wr*128 + i*16 + (lane>>4)*4 + e][n0 + wc*64 + j*16 + (lane&15)]
template <bool PUBLISH>
__device__ __forceinline__ void hg256_store(const f32x4 (&acc)[8][4], void* C, long ldc, int c_bf16, int c_trans, int m0,
                                            int n0, int w, int lane) {
  const int wr = w >> 2, wc = w & 3;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = m0 + wr * 128 + i * 16 + (lane >> 4) * 4;
      const int col = n0 + wc * 64 + j * 16 + (lane & 15);
      if (c_trans) {
        if (c_bf16) {
          uint16_t* p = static_cast<uint16_t*>(C) + (long)col * ldc + row0;
          const unsigned long long pk = (unsigned long long)f2bf(acc[i][j][0]) | ((unsigned long long)f2bf(acc[i][j][1]) << 16) |
                                        ((unsigned long long)f2bf(acc[i][j][2]) << 32) | ((unsigned long long)f2bf(acc[i][j][3]) << 48);
          if constexpr (PUBLISH)   // write-through (sc1) 8-byte store: visible to other XCDs once this wave's vmcnt drains
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            *reinterpret_cast<unsigned long long*>(p) = pk;
        } else {
          *reinterpret_cast<f32x4*>(static_cast<float*>(C) + (long)col * ldc + row0) = acc[i][j];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (c_bf16) static_cast<uint16_t*>(C)[(long)(row0 + e) * ldc + col] = f2bf(acc[i][j][e]);
          else static_cast<float*>(C)[(long)(row0 + e) * ldc + col] = acc[i][j][e];
        }
      }
    }
}

template <int DMAPOS>
__global__ __launch_bounds__(kThreads2) void k_hgemm_nt_256(HGemmArgs g) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[8 * 1024];   // 8 half-tile slots, the ONLY __shared__ object
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int m0, n0, klo, nk;
  hgemm256_tile_coords(g, m0, n0, klo, nk);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  long offA[2][2], offB[2][2];
  hg256_offsets(w, lane, g.lda, g.ldb, offA, offB);
  const uint16_t* Abase = g.A + (long)m0 * g.lda + klo;
  const uint16_t* Bbase = g.B + (long)n0 * g.ldb + klo;
  auto src = [&](int n, int t) -> const uint16_t* {
    const int j = n & 3;
    const long k0 = (long)(n >> 2) * TK;
    const long o = (j == 0) ? offB[0][t] : (j == 1) ? offA[0][t] : (j == 2) ? offB[1][t] : offA[1][t];
    return ((j & 1) ? Abase : Bbase) + o + k0;
  };
  hg256_mainloop<DMAPOS>(acc, lds, nk, w, lane, src, [](int) {});
  hg256_store<false>(acc, g.C, g.ldc, g.c_bf16, g.c_trans, m0, n0, w, lane);
}

// ---------------------------------------------------------------------------------------------
// Fused pair of triangular products  T3 = Ql T2 (k >= m)  ->  out = Ql' T3 (k <= m)   (psgd.py:191-192, M >= N branch)
// as ONE launch of one persistent block per 256 x 256 tile (r, c).  Alone, each product leaves the chip half idle at
// this tile size: tile row r of the first has K length 16 - r (in 256-wide chunks), of the second r + 1, and with one
// tile per CU the full-K tile is the critical path (2 x 16 chunks).  Fused, block (r, c) computes its T3 tile
// (16 - r chunks), publishes it, and then accumulates its out tile over the T3 tiles (j, c), j = r, r-1, ..., 0 IN THAT
// ORDER: tile j is finished by its producer at time 16 - j, exactly when this block gets to it -- 17 chunks for
// every block instead of 32 on the critical path.
// Hand-off (cdna_hip_programming.md, Guideline 16, R1): the T3 tile is stored write-through (8-byte agent-scope
// atomic stores = sc1), every storing wave drains vmcnt, the block's barrier, one lane stores the flag with an
// agent-scope atomic; the consumer polls that one word (one lane, relaxed, bounded), does ONE agent-scope acquire,
// drains, and the whole block passes a barrier before any wave issues a load of the handed-off bytes.  Flags are
// zeroed before every launch (by the factor-conversion kernel that opens the call).  The schedule wants every block
// resident (grid <= number of CUs, one block per CU: checked by the launcher), but CORRECTNESS does not depend on it: that
// check cannot see other streams or processes, so a block of this launch may have to wait for a CU.  Its consumers
// spin for a bounded time (~0.5 s).  A consumer whose bound is hit does not go on with bytes that were never published
// and does not give up either: it PRODUCES the missing T3 tile itself (phase A of the missing tile row -- the same
// instructions on the same inputs as the owner would run, so both write identical bytes), publishes it, and redoes
// its phase B from the start (same chunk order, hence the same bits as an undisturbed run).  Every spin therefore ends,
// every block finishes whatever the residency, and the result is always the undisturbed one.  The word in front of the
// flags counts such recoveries (diagnostics: psgd_kron_bf16_handoff_timeouts; never cleared by the library).
struct HPairArgs {
  const uint16_t* A1; long lda1;   // Ql  [M][M]   (k >= m)
  const uint16_t* B1; long ldb1;   // T2' [N][M]
  uint16_t* T3; long ldt;          // T3' [N][M]   written by phase A, read as the B operand of phase B
  const uint16_t* A2; long lda2;   // Ql' [M][M]   (k <= m)
  void* out; long ldo; int out_bf16, out_trans;
  int M, N;                        // M = the triangular factor's dimension (tile rows), N = the other one
  int c_begin, c_count;            // tile columns of this launch (hand-offs stay inside a column)
  unsigned* flags;                 // [M/256][N/256], zeroed before the launch
  unsigned* timeout;               // recoveries so far: +1 whenever a spin gave up and the consumer produced the tile itself
  unsigned spin_limit;             // polls before giving up (2^22 x ~64 cycles ~ 0.5 s; tests shrink it)
  int patch;                       // > 0: tile rows of an XCD's patch (tri_pair_body); 0: whole tile columns per XCD
};

template <int DMAPOS>
__device__ __forceinline__ void tri_pair_body(const HPairArgs& p, const int bid, u32x4* lds) {
  int& missing = *reinterpret_cast<int*>(&lds[8 * 1024]);   // tile row whose T3 tile a poll gave up on (-1: none)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) missing = -1;                              // ordered before its first reader by the barriers of phase A
  const int tiles_m = p.M / T2, tiles_n = p.N / T2;
  // blocks b, b + 8, ... share an XCD: give each XCD whole tile columns (the hand-offs of a column stay on one L2
  // when the column count allows); rows ascend with the block index inside a column
  // Round 6 (patch > 0): an XCD gets a PATCH of `patch` tile rows x (tiles per XCD / patch) tile columns instead.  A K step of the
  // XCD then pulls patch + columns operand panels through its L2 (4 + 8 = 12 at 4096^2, what the dense kernel's 4 x 8 patch pulls)
  // where whole columns pull 16 + 2 = 18; in phase B every row of a patch reads the SAME T3 chunk at the same time (chunk j is
  // consumed at time 16 - j by every row >= j).  The hand-offs of a column then cross XCDs: they always were agent-scope (write-through
  // stores, agent-scope flag, agent-scope acquire) because the non-multiple-of-8 mapping below spreads a column over the XCDs too.
  // Which CU computes a tile changes; what it computes, and in which order, does not: results are bitwise the same.
  int r, c;
  const int total = tiles_m * p.c_count;
  if (p.patch > 0 && total % 8 == 0 && tiles_m % p.patch == 0 && (total / 8) % p.patch == 0 &&
      p.c_count % ((total / 8) / p.patch) == 0) {
    const int xcd = bid % 8, j = bid / 8, pc = (total / 8) / p.patch;      // patch: p.patch rows x pc columns
    const int pcols = p.c_count / pc;                                       // patches side by side
    r = (xcd / pcols) * p.patch + j % p.patch;
    c = p.c_begin + (xcd % pcols) * pc + j / p.patch;
  } else if (p.c_count % 8 == 0) {
    const int xcd = bid % 8, j = bid / 8, cpx = p.c_count / 8;
    c = p.c_begin + xcd * cpx + j / tiles_m;
    r = j % tiles_m;
  } else {
    c = p.c_begin + bid / tiles_m;
    r = bid % tiles_m;
  }
  const int m0 = r * T2, n0 = c * T2;

  f32x4 acc[8][4];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // One pass = produce the T3 tile of row `prod` (this block's own row first), then consume.  A pass whose consumer
  // gave up on a tile comes round again with prod = that tile's row; an undisturbed run is exactly one pass.
  int prod = r;
  int gave = -1;                                           // wave 0, lane 0: its own copy of `missing` (no LDS read in the K loop)
  for (;;) {
    const int pm0 = prod * T2;
    // ---- phase A: T3 tile (prod, c) = sum over k >= pm0 of Ql[m][k] T2'[n][k]
    zero();
    {
      // (the lane id is made opaque per pass: otherwise the per-lane offsets of BOTH phases are hoisted out of the pass
      // loop as invariants and their 32 registers push the kernel into scratch)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      long offA[2][2], offB[2][2];
      hg256_offsets(w, ln, p.lda1, p.ldb1, offA, offB);
      const uint16_t* Abase = p.A1 + (long)pm0 * p.lda1 + pm0;
      const uint16_t* Bbase = p.B1 + (long)n0 * p.ldb1 + pm0;
      auto src = [&](int n, int t) -> const uint16_t* {
        const int j = n & 3;
        const long k0 = (long)(n >> 2) * TK;
        const long o = (j == 0) ? offB[0][t] : (j == 1) ? offA[0][t] : (j == 2) ? offB[1][t] : offA[1][t];
        return ((j & 1) ? Abase : Bbase) + o + k0;
      };
      hg256_mainloop<DMAPOS>(acc, lds, (p.M - pm0) / TK, w, lane, src, [](int) {});
    }
    // publish: T3'[n0 + col][pm0 + row] (transposed, bf16), write-through; drain; barrier; flag
    hg256_store<true>(acc, p.T3, p.ldt, 1, 1, pm0, n0, w, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(p.flags + prod * tiles_n + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- phase B: out tile (r, c) = sum over chunks j = r .. 0 of Ql'[m][k in chunk j] T3'[n][k in chunk j]
    zero();
    {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      long offA[2][2], offB[2][2];
      hg256_offsets(w, ln, p.lda2, p.ldt, offA, offB);
      const uint16_t* Abase = p.A2 + (long)m0 * p.lda2;
      const uint16_t* Bbase = p.T3 + (long)n0 * p.ldt;
      auto src = [&](int n, int t) -> const uint16_t* {
        const int j = n & 3, i = n >> 2;                     // i-th K tile of the sequence
        const long k0 = (long)((r - (i >> 2)) * 4 + (i & 3)) * TK;   // chunk r - i/4, K tile i%4 inside it
        const long o = (j == 0) ? offB[0][t] : (j == 1) ? offA[0][t] : (j == 2) ? offB[1][t] : offA[1][t];
        return ((j & 1) ? Abase : Bbase) + o + k0;
      };
      auto pre = [&](int n) {
        if ((n & 15) != 0) return;                           // first half-tile of a new chunk (n > 0 here)
        const int jchunk = r - (n >> 4);
        if (w == 0) {
          if (lane == 0 && gave < 0) {                       // (after a give-up the rest of this pass is discarded: no more polls)
            const unsigned* f = p.flags + jchunk * tiles_n + c;
            unsigned spins = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(4);
              if (++spins > p.spin_limit) {
                __hip_atomic_fetch_add(p.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                missing = gave = jchunk;                     // `missing`: read by every wave after the pass (barriers in between)
                break;
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        HG_FENCE();
        __builtin_amdgcn_s_barrier();
        HG_FENCE();
      };
      // the first chunk is this block's own tile: its stores were drained before the barrier above
      hg256_mainloop<DMAPOS>(acc, lds, (r + 1) * 4, w, lane, src, pre);
    }
    __syncthreads();
    const int ms = missing;
    if (ms < 0) break;                                       // the undisturbed case: one pass
    __syncthreads();                                         // every wave has read `missing` before it is reset
    if (tid == 0) missing = -1;
    gave = -1;
    prod = ms;                                               // produce the tile nobody published in time, then consume again
  }
  hg256_store<false>(acc, p.out, p.ldo, p.out_bf16, p.out_trans, m0, n0, w, lane);
}

template <int DMAPOS>
__global__ __launch_bounds__(kThreads2) void k_hgemm_tri_pair_256(HPairArgs p) {
  // ONE __shared__ object: a second one beside the LDS-DMA staging array makes hipcc drain every outstanding DMA
  // (s_waitcnt vmcnt(0)) in front of the first fragment read of EVERY K tile (cdna_hip_programming.md section 5, "Projection
  // GEMM" item 4a; found in this kernel's ISA in round 3: the flag word of the hand-off had been such an object since
  // round 2).  The flag lives in a 9th-slot word of the same array.
  __shared__ __attribute__((aligned(16))) u32x4 lds[8 * 1024 + 1];
  tri_pair_body<DMAPOS>(p, (int)blockIdx.x, lds);
}

// WHAT-IF (round 5; timing only, WRONG results; psgd_kron_bf16_set_tuning(6, 1)): both fused pairs of the apply as
// ONE grid -- blocks [0, n0) are the first pair's, the rest the second pair's, which start on the CUs the first pair's blocks leave and
// do NOT wait for the tiles of Y they read.  What this launch gains over two launches is the most a real one-grid merge (with a second
// hand-off protocol between the pairs) could gain.  profiles/r05_bf16_pair_merge_whatif.txt
template <int DMAPOS>
__global__ __launch_bounds__(kThreads2) void k_hgemm_tri_pair2_256(HPairArgs p0, HPairArgs p1, int n0) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[8 * 1024 + 1];
  if ((int)blockIdx.x < n0) tri_pair_body<DMAPOS>(p0, (int)blockIdx.x, lds);
  else tri_pair_body<DMAPOS>(p1, (int)blockIdx.x - n0, lds);
}

// ---------------------------------------------------------------------------------------------
// Stream-K over 256 x 256 tiles (round 4): products of the bf16-operand UPDATE on the 8-phase main loop.
// The two gradients (psgd.py:175-176) are 2 x 136 equal upper tiles -- 272 on 256 CUs: two rounds for 1.06 rounds of work -- and
// the factor updates (:179) have K ranges that grow with the distance from the diagonal (the full-K tile is the critical path).
// They ran on the 128 x 128 register-staged kernel (22 % of the matrix peak against this loop's 55 %).  Here the launch is ONE
// workgroup per CU; the K tiles of all output tiles, in tile order, form one line of `total` units, and workgroup b (in
// XCD-contiguous order) takes the units [total b / G, total (b + 1) / G) -- at most one unit more than any other.  A range covers
// pieces of one or more tiles: a piece that is a WHOLE tile goes through the epilogue at once; any other piece is stored as an
// fp32 partial tile (at most two per workgroup: the tail of the tile its range starts in, the head of the tile it ends in), and
// a second, small launch (k_hgemm_sk_fix: 32 workgroups per cut tile) adds the pieces of every cut tile in the order of
// the K axis and runs the epilogue.  No workgroup ever waits for another one: nothing here depends on residency, on other
// streams or on launch order, and the sums are reproducible.  (An earlier form finished cut tiles inside the main launch --
// flags, bounded polls, a recovery path: two concurrent launches of it could starve each other, and its second inlined copy of
// the main loop pushed the kernel into scratch, which costs ~350 us per LAUNCH on this stack.)
// Contract: M, N multiples of 256, K ranges multiples of 64 (launcher: sk_legal).
struct SkArgs {
  HGemmArgs g[2];
  int nprob;
  int total, units0;       // K tiles (TK = 64) of all tiles of the launch / of the first product's
  int dp_rounds, dp_nk;    // whole-tile rounds ahead of the stream-K part (tiles of dp_nk K tiles each; see k_hgemm_sk_256)
  float* partial;          // [grid][2][65536]: slot 0 = a piece that does not hold its tile's first K tile, slot 1 = one that does
};

// Tile order of one product: patches of 4 x 8 tiles (ragged at the edges), patch rows from the top, the patches of a row from the
// left (an upper-triangular output: from the patch that holds the diagonal), the tiles of a patch row by row; `upper` skips c < r.
// Consecutive ranges run on one XCD: the 32 workgroups of an XCD then work on ~one patch, whose K step pulls 4 + 8 operand panels
// through that L2 (tile-row order pulled 1 + 30: the gradients ran at 340 us instead of 230, bound by Infinity-Cache traffic).
constexpr int kSkPR = 4, kSkPC = 8;
__device__ __host__ __forceinline__ int sk_nk(int K, int kmode, int r, int c, int& klo) {
  const int m0 = r * T2, n0 = c * T2;
  int lo = 0, hi = K;
  if (kmode & KLO_M) lo = lo > m0 ? lo : m0;
  if (kmode & KLO_N) lo = lo > n0 ? lo : n0;
  if (kmode & KHI_M) hi = hi < m0 + T2 ? hi : m0 + T2;
  if (kmode & KHI_N) hi = hi < n0 + T2 ? hi : n0 + T2;
  klo = (lo / TK) * TK;
  return hi > klo ? (hi - klo + TK - 1) / TK : 0;
}
__device__ __host__ __forceinline__ bool sk_next(int Tm, int Tn, int upper, int& r, int& c) {      // false: past the last tile
  do {
    const int c0 = c & ~(kSkPC - 1), r0 = r & ~(kSkPR - 1);
    if (((c + 1) & (kSkPC - 1)) != 0 && c + 1 < Tn) {
      ++c;
    } else if (((r + 1) & (kSkPR - 1)) != 0 && r + 1 < Tm) {
      ++r; c = c0;
    } else if (c0 + kSkPC < Tn) {
      c = c0 + kSkPC; r = r0;
    } else {
      r = r0 + kSkPR;
      if (r >= Tm) return false;
      c = upper ? (r & ~(kSkPC - 1)) : 0;
    }
  } while (upper && c < r);
  return true;
}
__device__ __forceinline__ bool sk_prev(int Tm, int Tn, int upper, int& r, int& c) {               // false: before the first tile
  do {
    const int c0 = c & ~(kSkPC - 1), r0 = r & ~(kSkPR - 1);
    if (c > c0) {
      --c;
    } else if (r > r0) {
      --r; c = (c0 + kSkPC < Tn ? c0 + kSkPC : Tn) - 1;
    } else if (c0 > (upper ? (r0 & ~(kSkPC - 1)) : 0)) {
      c = c0 - 1; r = (r0 + kSkPR < Tm ? r0 + kSkPR : Tm) - 1;
    } else {
      if (r0 == 0) return false;
      r = r0 - 1; c = Tn - 1;
    }
  } while (upper && c < r);
  return true;
}

__device__ __forceinline__ void sk_negate(f32x4 (&acc)[8][4]) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = -acc[i][j];
}

// acc = the K tiles [a, b) (counted from the tile's klo) of tile (m0, n0), from zero
__device__ __forceinline__ void sk_piece(f32x4 (&acc)[8][4], u32x4* lds, const HGemmArgs& g, int m0, int n0, int klo, int a, int b) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int td = threadIdx.x;
  asm volatile("" : "+v"(td));                       // (keeps the per-lane offsets of the pieces from being hoisted together)
  const int lane = td & 63, w = td >> 6;
  long offA[2][2], offB[2][2];
  hg256_offsets(w, lane, g.lda, g.ldb, offA, offB);
  const uint16_t* Abase = g.A + (long)m0 * g.lda + klo + (long)a * TK;
  const uint16_t* Bbase = g.B + (long)n0 * g.ldb + klo + (long)a * TK;
  auto src = [&](int n, int t) -> const uint16_t* {
    const int j = n & 3;
    const long k0 = (long)(n >> 2) * TK;
    const long o = (j == 0) ? offB[0][t] : (j == 1) ? offA[0][t] : (j == 2) ? offB[1][t] : offA[1][t];
    return ((j & 1) ? Abase : Bbase) + o + k0;
  };
  // kflip (C = -A1 B1' + A2 B2' over a concatenated K axis): the accumulators change sign before K tile kflip
  const int flip = g.kflip > 0 ? g.kflip - klo / TK - a : -1;
#ifdef SK_NOHOOK
  hg256_mainloop<0>(acc, lds, b - a, w, lane, src, [](int) {});
#else
  hg256_mainloop<0>(acc, lds, b - a, w, lane, src, [](int) {}, [&](int kt) {
    if (kt == flip && kt > 0) sk_negate(acc);
  });
#endif
  if (g.kflip > 0 && flip >= b - a) sk_negate(acc);  // the whole piece lies in the negated half
}

// partial tiles: [q = i * 4 + j][thread] f32x4.  ONE per-lane pointer walks the 32 rows and is made opaque at every step: left to
// itself the compiler materialises all 32 addresses ahead of the accesses and spills them and the accumulators.
__device__ __forceinline__ void sk_put(const f32x4 (&acc)[8][4], float* dst, int tid) {
  f32x4* d = reinterpret_cast<f32x4*>(dst) + tid;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(d));
      *d = acc[i][j];
      d += kThreads2;
    }
}
// epilogues on the 256^2 accumulator layout: acc[i][j][e] = C[m0 + wr*128 + (ibase + i)*16 + (lane>>4)*4 + e][n0 + wc*64 + j*16 + (lane&15)];
// NI = 8, NJ = 4, ibase = jbase = 0: a whole tile (main launch); NI = NJ = 1: one 16 x 16 block per wave (fix-up launch)
// red: eight floats of LDS for one max per WORKGROUP (fix-up launch: 32 x 8 waves per cut tile all finish together, and thousands of
// atomics on one address serialise in L2 -- 40 of that launch's 49 us); nullptr: one atomic per wave (main launch: spread over its run)
template <int NI, int NJ>
__device__ __forceinline__ void sk_epilogue(const HGemmArgs& g, f32x4 (&acc)[NI][NJ], int ibase, int jbase, int m0, int n0, int w, int lane,
                                            float* red = nullptr) {
  const int wr = w >> 2, wc = w & 3;
  if (g.epi == HEPI_TRIU_MAX) {
    float vmax = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row0 = m0 + wr * 128 + (ibase + i) * 16 + (lane >> 4) * 4, col = n0 + wc * 64 + (jbase + j) * 16 + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (col >= row0 + e) ? acc[i][j][e] : 0.0f;
          vmax = amaxf(vmax, fabsf(v));
          acc[i][j][e] = v;
        }
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
    if (red) {
      if (lane == 0) red[w] = vmax;
      __syncthreads();
      if (w == 0) {
        vmax = lane < kThreads2 / 64 ? red[lane] : 0.0f;
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
      }
      if (w != 0) vmax = 0.0f;
    }
    if (lane == 0 && __float_as_uint(vmax) > *reinterpret_cast<volatile unsigned*>(g.maxout)) atomicMax(reinterpret_cast<int*>(g.maxout), __float_as_int(vmax));      // (looks first: atomics on one address serialise in L2)
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row0 = m0 + wr * 128 + (ibase + i) * 16 + (lane >> 4) * 4, col = n0 + wc * 64 + (jbase + j) * 16 + (lane & 15);
      if (g.c_trans) {
        if (g.c_bf16) {
          const unsigned long long pk = (unsigned long long)f2bf(acc[i][j][0]) | ((unsigned long long)f2bf(acc[i][j][1]) << 16) |
                                        ((unsigned long long)f2bf(acc[i][j][2]) << 32) | ((unsigned long long)f2bf(acc[i][j][3]) << 48);
          *reinterpret_cast<unsigned long long*>(static_cast<uint16_t*>(g.C) + (long)col * g.ldc + row0) = pk;
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

// the tile that holds unit u (0 <= u < total): its problem, coordinates, first unit, K range.
// NOT a walk over the tiles: a tile-by-tile scan is ~100 dependent scalar instructions per tile -- 50 us for the last workgroups of a
// 272-tile launch, in the main launch and again in the fix-up.  The K-tile count of a tile is affine in its column (sk_legal's
// modes), so a run of tiles of one row sums in closed form: patch rows, then the patches of the row, then the rows of the patch, then
// the tiles of the row -- T/4 + T/8 + 4 + 8 short steps, in the order sk_next walks.
struct SkAt { int prob, r, c, ubase, klo, nk, Tm, Tn, upper, K, kmode; };
__device__ __forceinline__ void sk_dims(const SkArgs& p, int prob, SkAt& t) {
  const HGemmArgs& g = prob ? p.g[1] : p.g[0];
  t.Tm = g.M / T2; t.Tn = g.N / T2; t.upper = g.sym != 0; t.K = g.K; t.kmode = g.kmode;
}
__device__ __forceinline__ int sk_seg(const SkAt& t, int r, int ca, int cb) {       // units of the tiles (r, ca .. cb)
  if (t.upper && ca < r) ca = r;
  if (cb >= t.Tn) cb = t.Tn - 1;
  if (cb < ca) return 0;
  int klo;
  return (sk_nk(t.K, t.kmode, r, ca, klo) + sk_nk(t.K, t.kmode, r, cb, klo)) * (cb - ca + 1) / 2;
}
__device__ __forceinline__ void sk_locate(const SkArgs& p, int u, SkAt& t) {
  t.prob = (p.nprob > 1 && u >= p.units0) ? 1 : 0;
  t.ubase = t.prob ? p.units0 : 0;
  sk_dims(p, t.prob, t);
  int left = u - t.ubase, r0 = 0, c0, s;
  for (;; r0 += kSkPR) {                                             // patch rows
    s = 0;
    for (int r = r0; r < r0 + kSkPR && r < t.Tm; ++r) s += sk_seg(t, r, 0, t.Tn - 1);
    if (left < s || r0 + kSkPR >= t.Tm) break;
    left -= s; t.ubase += s;
  }
  for (c0 = t.upper ? (r0 & ~(kSkPC - 1)) : 0;; c0 += kSkPC) {        // the patches of that row
    s = 0;
    for (int r = r0; r < r0 + kSkPR && r < t.Tm; ++r) s += sk_seg(t, r, c0, c0 + kSkPC - 1);
    if (left < s || c0 + kSkPC >= t.Tn) break;
    left -= s; t.ubase += s;
  }
  for (t.r = r0;; ++t.r) {                                            // the rows of that patch
    s = sk_seg(t, t.r, c0, c0 + kSkPC - 1);
    if (left < s || t.r + 1 >= t.Tm || ((t.r + 1) & (kSkPR - 1)) == 0) break;
    left -= s; t.ubase += s;
  }
  for (t.c = (t.upper && c0 < t.r) ? t.r : c0;; ++t.c) {              // the tiles of that row
    t.nk = sk_nk(t.K, t.kmode, t.r, t.c, t.klo);
    if (left < t.nk || t.c + 1 >= t.Tn) break;
    left -= t.nk; t.ubase += t.nk;
  }
}
__device__ __forceinline__ int sk_lb(int G) {       // blocks b, b + 8, ... share an XCD: consecutive ranges (neighbouring tiles) on one L2
  return (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
}

// Whole-tile rounds first.  Workgroups that share operand panels must walk K TOGETHER: with every range starting at its own K offset
// (pure stream-K) a panel slice is fetched by one workgroup at a time, nothing is reused in the 4-MiB L2 and the gradients ran at 62 %
// of the dense kernel's rate on Infinity-Cache traffic.  So while all tiles cost the same (dp_nk K tiles: the gradients when M = N),
// tiles [round G, (round + 1) G) are taken one per workgroup from their first K tile, like the dense kernel, and only the tiles of
// the last, incomplete round are dealt out as ranges of units.
__global__ __launch_bounds__(kThreads2) void k_hgemm_sk_256(SkArgs p) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[8 * 1024];          // ONE __shared__ object (see k_hgemm_tri_pair_256)
  const int G = gridDim.x, lb = sk_lb(G);
  SkAt t;
  f32x4 acc[8][4];
  const int dp_units = p.dp_rounds * G * p.dp_nk;
  for (int round = 0; round < p.dp_rounds; ++round) {
    const int u = (round * G + lb) * p.dp_nk;
    if (u >= p.total) break;                           // (a last round that is not full)
    sk_locate(p, u, t);
    const HGemmArgs g = t.prob ? p.g[1] : p.g[0];
    sk_piece(acc, lds, g, t.r * T2, t.c * T2, t.klo, 0, t.nk);
    int td = threadIdx.x;
    asm volatile("" : "+v"(td));
    sk_epilogue<8, 4>(g, acc, 0, 0, t.r * T2, t.c * T2, td >> 6, td & 63);
  }
  if (dp_units >= p.total) return;
  auto start = [&](int b) { return dp_units + (int)((unsigned)(p.total - dp_units) * (unsigned)b / (unsigned)G); };      // (total * G < 2^31: launcher)
  const int u0 = start(lb), u1 = start(lb + 1);
  if (u1 <= u0) return;
  sk_locate(p, u0, t);
  int lo = u0;
  for (;;) {
    const HGemmArgs g = t.prob ? p.g[1] : p.g[0];
    const int end = t.ubase + t.nk, hi = u1 < end ? u1 : end;
    const int a = lo - t.ubase, b = hi - t.ubase;      // K tiles [a, b) of tile t
    const int m0 = t.r * T2, n0 = t.c * T2;
    sk_piece(acc, lds, g, m0, n0, t.klo, a, b);
    // (the thread id is made opaque per piece: otherwise the per-lane offsets of the partial-tile accesses and of every epilogue are
    //  hoisted out of the piece loop as invariants, stay live across the K loop and push it into scratch -- a reload inside the K
    //  loop is followed by s_waitcnt vmcnt(0), which drains the LDS-DMA pipeline on every K tile)
    int td = threadIdx.x;
    asm volatile("" : "+v"(td));
    if (a == 0 && b == t.nk) sk_epilogue<8, 4>(g, acc, 0, 0, m0, n0, td >> 6, td & 63);
    else sk_put(acc, p.partial + ((long)lb * 2 + (a == 0 ? 1 : 0)) * 65536, td);
    lo = hi;
    if (lo >= u1) break;
    t.ubase = end;
    if (!sk_next(t.Tm, t.Tn, t.upper, t.r, t.c)) {
      ++t.prob; t.r = t.c = 0;
      sk_dims(p, t.prob, t);
    }
    t.nk = sk_nk(t.K, t.kmode, t.r, t.c, t.klo);
  }
}

// 32 workgroups (one per 16 x 16 block of the wave tiles: one f32x4 per lane) per CUT TILE, named by the first range boundary b
// inside it (the launcher lists them: a launch over all G - 1 boundaries spent 145 us on workgroups that only found out they had
// nothing to do).  They finish the tile: the head
// piece of workgroup b - 1, then the pieces of b, b + 1, ... while their ranges reach into the tile, added in that order (the
// loads of eight pieces are in flight together: one memory latency per eight pieces, not per piece); epilogue.
constexpr int kSkMaxGrid = 256;   // workgroups of a stream-K launch (one per CU); sizes the partial-tile scratch
// the cut tiles of a launch (host: launch_hgemm_sk): first boundary inside, number of boundaries inside (= pieces after the head), tile
struct SkCuts { int n; unsigned char b[kSkMaxGrid], cnt[kSkMaxGrid], prob[kSkMaxGrid], r[kSkMaxGrid], c[kSkMaxGrid]; };
__global__ __launch_bounds__(kThreads2) void k_hgemm_sk_fix(SkArgs p, SkCuts cuts) {
  const int ci = blockIdx.x / 32, blk = blockIdx.x % 32;
  const int b = cuts.b[ci], cnt = cuts.cnt[ci];
  const HGemmArgs g = cuts.prob[ci] ? p.g[1] : p.g[0];
  const int tid = threadIdx.x;
  const f32x4* base = reinterpret_cast<const f32x4*>(p.partial) + (long)blk * kThreads2 + tid;      // this lane's f32x4 of slot 0 of workgroup 0
  f32x4 acc[1][1];
  const f32x4 head = base[((long)(b - 1) * 2 + 1) * 16384];
  for (int k0 = 0; k0 < cnt; k0 += 16) {               // (sixteen pieces in flight: one memory latency for the usual cut tile)
    f32x4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (k0 + k < cnt) ? base[(long)(b + k0 + k) * 2 * 16384] : f32x4{0.f, 0.f, 0.f, 0.f};
    if (k0 == 0) acc[0][0] = head;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k0 + k < cnt) acc[0][0] += v[k];
  }
  __shared__ float red[kThreads2 / 64];
  sk_epilogue<1, 1>(g, acc, blk >> 2, blk & 3, cuts.r[ci] * T2, cuts.c[ci] * T2, tid >> 6, tid & 63, red);
}

// dst (bf16) = src or src', 64 x 64 tiles through LDS.  SRC_BF16 selects the source element type.
template <bool SRC_BF16>
__global__ __launch_bounds__(kThreads) void k_to_bf16(const void* src, long lds_, uint16_t* dst, long ldd, int rows,
                                                      int cols, int transpose, uint16_t* dst2 = nullptr, long ldd2 = 0) {
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
    if (dst2 && c0 + a < cols && r0 + b < rows) dst2[(long)(c0 + a) * ldd2 + r0 + b] = tile[b][a];      // (+ the transposed copy from the same read)
  }
}

// bf16 copies of an upper-triangular fp32 factor Q [n][n]: dst = bf16(Q) and/or dstT = bf16(Q').  64 x 64 tiles,
// 16-byte global accesses on both sides.  Only the 256 x 256 blocks on or above the block diagonal are touched:
// the GEMMs restrict their K ranges at tile granularity (<= 256), so they never read the rest of the copies.
// Both factors of a call in ONE launch (blockIdx.z selects the job); block (0,0,0) also zeroes the hand-off words of
// the fused triangular pair, which runs later on the same stream.
struct FactorJob { const float* src; uint16_t* dst; uint16_t* dstT; int n; };

__global__ __launch_bounds__(kThreads) void k_factors_to_bf16(FactorJob j0, FactorJob j1, unsigned* zero_words, int nzero) {
  __shared__ uint16_t tile[64][72];
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < nzero; i += kThreads) zero_words[i] = 0u;
  const FactorJob j = blockIdx.z ? j1 : j0;
  const float* __restrict__ src = j.src;
  const int n = j.n;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (r0 >= n || c0 >= n || (r0 >> 8) > (c0 >> 8)) return;
  const int t = threadIdx.x, row = t >> 2, cq = (t & 3) * 16;
  uint16_t v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r0 + row < n && c0 + cq + 4 * q < n) x = *reinterpret_cast<const f32x4*>(src + (long)(r0 + row) * n + c0 + cq + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = f2bf(x[e]);
  }
  if (j.dst) {
    if (r0 + row < n) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (c0 + cq + 8 * h < n) {
          u32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = (unsigned)v[8 * h + 2 * e] | ((unsigned)v[8 * h + 2 * e + 1] << 16);
          *reinterpret_cast<u32x4*>(j.dst + (long)(r0 + row) * n + c0 + cq + 8 * h) = pk;
        }
    }
  }
  if (j.dstT) {                                   // uniform per block
#pragma unroll
    for (int e = 0; e < 16; ++e) tile[row][cq + e] = v[e];
    __syncthreads();
    const int a = t >> 2, bq = (t & 3) * 16;      // row a of the transposed tile, 16 consecutive columns
    if (c0 + a < n) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (r0 + bq + 8 * h < n) {
          u32x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            pk[e] = (unsigned)tile[bq + 8 * h + 2 * e][a] | ((unsigned)tile[bq + 8 * h + 2 * e + 1][a] << 16);
          *reinterpret_cast<u32x4*>(j.dstT + (long)(c0 + a) * n + r0 + bq + 8 * h) = pk;
        }
    }
  }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

struct HWs {
  uint16_t *Ql, *QlT, *Qr, *QrT, *GT, *T1, *T2, *T3;
  unsigned* flags;        // hand-off words of the fused triangular pairs: [timeout, pad x 3, flag set 0, flag set 1]
  int64_t flag_bytes;
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
  k.flag_bytes = ((int64_t)((M + T2 - 1) / T2) * ((N + T2 - 1) / T2) * 8 + 16 + 15) / 16 * 16;   // two flag sets
  k.flags = reinterpret_cast<unsigned*>(take(k.flag_bytes));
  k.total = off;
  return k;
}

static int g_two_pairs = 1;       // tuning key 1: 0 = keep the Gram-first chain even where two fused pairs are possible
static unsigned g_spin_limit = 1u << 22;   // psgd_kron_bf16_set_tuning key 2 (log2): hand-off polls before a consumer gives up
static int g_trsm_lite = 1;        // psgd_kron_bf16_set_tuning key 3: 1 = the trailing PRODUCTS of the blocked solves keep the three
                                   // leading terms of the bf16 x 3 split (2^-16 per product, below the 2^-9 rounding dX arrives
                                   // with); 0 = all six terms (fp32-level).  Strip substitutions and diagonal-block inverses are
                                   // fp32 either way.  tests/test_kron_gpu.py::test_bf16_update_solves_with_ill_conditioned_factors
// (where a phase of the 256^2 kernels issues its LDS-DMA pieces -- hg256_mainloop's template parameter -- was tuning key 4 until round 4:
// the published template's placement, 0, wins; 1 / 2 measured 5-12 % slower, profiles/r03_bf16_dmapos_ab.txt)
static int g_hgemm_variant = 0;   // 0: auto (256^2 8-phase kernel for large dense products, fused triangular pair when every
                                  //    tile gets its own CU, else 128^2 register-staged); 4: auto without the fused pair;
                                  // 1: always 128^2 register-staged; 2: 128^2 LDS-DMA ring; 3: 256^2 wherever its shape contract holds

static int launch_hgemm_args(const HGemmArgs& g, hipStream_t st) {
  const uint16_t *A = g.A, *B = g.B;
  const long lda = g.lda, ldb = g.ldb;
  const int M = g.M, N = g.N, K = g.K, kmode = g.kmode, sym = g.sym;
  if (g.epi != HEPI_STORE || g.kflip) {       // the update epilogues live in the 128^2 register-staged kernel only
    const int tq = (M + TM - 1) / TM;
    hipLaunchKernelGGL(k_hgemm_nt, dim3(sym ? tq * (tq + 1) / 2 : tq * ((N + TN - 1) / TN)), dim3(kThreads), 0, st, g);
    return (int)hipGetLastError();
  }
  const int tm_ = (M + TM - 1) / TM;
  const int nt = sym ? tm_ * (tm_ + 1) / 2 : tm_ * ((N + TN - 1) / TN);
  const bool interior = (M % TM == 0) && (N % TN == 0) && (K % TM == 0) && (lda % 8 == 0) && (ldb % 8 == 0) &&
                        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  const bool big = interior && !sym && (M % T2 == 0) && (N % T2 == 0) && (K % T2 == 0);
  // auto: dense products only (a triangular K range makes the single full-K tile per CU the critical path: measured
  // equal to the 128^2 kernel), and only when the 256^2 tiles fill >= 80 % of the CU slots of their last wave
  const int nt2 = (M / T2) * (N / T2);
  const bool fills = nt2 * 5 >= ((nt2 + 255) / 256) * 256 * 4;
  const bool use256 = big && (((g_hgemm_variant == 0 || g_hgemm_variant == 4) && kmode == 0 && fills) || g_hgemm_variant == 3);
  if (use256) {
    const dim3 grid((M / T2) * (N / T2));
    hipLaunchKernelGGL(k_hgemm_nt_256<0>, grid, dim3(kThreads2), 0, st, g);
  }
  else if (g_hgemm_variant == 2 && interior && !sym) hipLaunchKernelGGL(k_hgemm_nt_dma, dim3(nt), dim3(kThreads), 0, st, g);
  else hipLaunchKernelGGL(k_hgemm_nt, dim3(nt), dim3(kThreads), 0, st, g);
  return (int)hipGetLastError();
}

static int hgemm_blocks(const HGemmArgs& g) {
  const int tq = (g.M + TM - 1) / TM;
  return g.sym ? tq * (tq + 1) / 2 : tq * ((g.N + TN - 1) / TN);
}

static int launch_hgemm_two(const HGemmArgs& g0, const HGemmArgs& g1, hipStream_t st) {
  const int n0 = hgemm_blocks(g0), n1 = hgemm_blocks(g1);
  hipLaunchKernelGGL(k_hgemm_nt_two, dim3(n0 + n1), dim3(kThreads), 0, st, g0, g1, n0, n1);
  return (int)hipGetLastError();
}

static int device_cu_count();
// ---- stream-K launches (k_hgemm_sk_256) ---------------------------------------------------------------------------------
static int g_pair_xcd_patch = -1; // psgd_kron_bf16_set_tuning key 7: tile rows of an XCD's patch in the fused pair; 0 = whole tile columns per XCD
                                  // (rounds 1-5); -1 (default) = by shape: 8 from 32 tile rows on (8192 x 2048: 0.411 -> 0.386 ms), 0 below --
                                  // at 4096^2 the patch (12 operand panels per K step and L2 instead of 18) changes NOTHING: 0.2575 ms either
                                  // way, bitwise equal (profiles/r06_bf16_patch_ab.txt): the pair's K loop is not bound by its L2 panels
static int g_pair_merge_whatif = 0;   // psgd_kron_bf16_set_tuning key 6: 1 = WHAT-IF, wrong results: the apply's two fused pairs as one grid (see k_hgemm_tri_pair2_256)
static int g_pair_patch = 1;      // psgd_kron_bf16_set_tuning key 5: 1 (default) = the factor updates' tiles in 4 x 4 patches (hgemm_tile_coords,
                                  // sym == 2; from 32 x 32 tiles of 128 on: 4096^2 update 1.716 -> 1.698 ms, equal elsewhere), 0 = tile rows
static int g_streamk = 1;         // psgd_kron_bf16_set_tuning key 4: 0 = the update's products stay on the 128^2 register-staged kernel,
                                  // 2 = stream-K for every shape the kernel can take (tests), 3 = 2 without whole-tile rounds
static bool sk_legal(const HGemmArgs& g) {
  if (g.M <= 0 || g.N <= 0 || (g.M % T2) || (g.N % T2) || (g.K % TK) || (g.lda % 8) || (g.ldb % 8)) return false;
  if (g.M / T2 > 255 || g.N / T2 > 255) return false;          // (SkCuts names tiles in bytes)
  if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15) return false;
  if (g.sym && g.M != g.N) return false;
  if (g.kflip && g.kmode) return false;
  if (g.kmode != 0 && g.kmode != KLO_M && g.kmode != KLO_N && g.kmode != (KLO_M | KHI_N)) return false;      // (sk_locate: K tiles affine in the column)
  if (g.epi == HEPI_STORE) return g.sym == 0;
  return g.epi == HEPI_TRIU_MAX && g.sym == 1;       // (the factor updates' HEPI_D_MINUS stays on the 128^2 kernel: see the call)
}
static int sk_units(const HGemmArgs& g) {            // K tiles of all tiles of one product
  const int Tm = g.M / T2, Tn = g.N / T2, o = g.sym != 0;
  int r = 0, c = 0, klo, tot = 0;
  do tot += sk_nk(g.K, g.kmode, r, c, klo); while (sk_next(Tm, Tn, o, r, c));
  return tot;
}
// one or two products in one launch (+ the fix-up launch); returns 2 when the shapes are not the kernel's (the caller falls back)
static int launch_hgemm_sk(const HGemmArgs* g, int n, float* partial, hipStream_t st) {
  if (!g_streamk || !partial || n < 1 || n > 2) return 2;
  SkArgs a = {};
  a.nprob = n;
  for (int i = 0; i < n; ++i) {
    if (!sk_legal(g[i])) return 2;
    a.g[i] = g[i];
    if (i == 0) a.units0 = sk_units(g[0]);
    a.total += sk_units(g[i]);
  }
  int grid = device_cu_count();
  if (grid > kSkMaxGrid) grid = kSkMaxGrid;
  if ((long)a.total * grid >= (1L << 31)) return 2;
  if (g_streamk >= 2) {                              // (tests: every legal shape, however small -- ranges of a few K tiles, tiles cut many times)
    if (grid > a.total / 2) grid = a.total / 2 > 0 ? a.total / 2 : 1;
  } else if (a.total < 8 * grid) {
    return 2;                                        // too little work for one workgroup per CU
  }
  if (grid >= 8) grid = grid / 8 * 8;
  a.partial = partial;
  // whole-tile rounds while every tile costs the same; a last round that would be more than ~60 % full is run as a round too
  // (ranges of units run at ~0.6 of the lockstep rate: no operand reuse in L2)
  bool uniform = g[0].kmode == 0;
  for (int i = 1; i < n; ++i) uniform = uniform && g[i].kmode == 0 && g[i].K == g[0].K;
  if (uniform && g_streamk != 3) {
    a.dp_nk = g[0].K / TK;
    const int tiles = a.total / a.dp_nk;
    a.dp_rounds = tiles / grid;
    if ((tiles % grid) * 10 > grid * 6 || (tiles % grid) * a.dp_nk < 2 * grid) ++a.dp_rounds;      // (... or too short to deal out)
    if (tiles % grid == 0 && a.dp_rounds > tiles / grid) --a.dp_rounds;
  }
  // (ranges only -- no whole-tile round: M != N, or fewer tiles than workgroups -- measured no better than the 128^2 kernels:
  //  2048 x 4096 1.17 -> 1.23 ms, 2048^2 0.65 -> 0.68: profiles/r04_streamk_ab.txt)
  if (g_streamk == 1 && a.dp_rounds == 0) return 2;
  hipLaunchKernelGGL(k_hgemm_sk_256, dim3(grid), dim3(kThreads2), 0, st, a);
  if (hipGetLastError() != hipSuccess) return 1;
  if (grid > 1 && (long)a.dp_rounds * grid * a.dp_nk < a.total) {
    // the cut tiles: walk the tiles and the range boundaries of the stream-K part together (both ascend along the line of units)
    SkCuts cuts = {};
    const int dp_units = a.dp_rounds * grid * a.dp_nk;
    auto start = [&](int q) { return dp_units + (int)((unsigned)(a.total - dp_units) * (unsigned)q / (unsigned)grid); };
    int prob = 0, r = 0, c = 0, ubase = 0, klo, b = 1;
    for (;;) {
      const HGemmArgs& gp = g[prob];
      const int nk = sk_nk(gp.K, gp.kmode, r, c, klo), end = ubase + nk;
      while (b < grid && start(b) <= ubase) ++b;                       // boundaries at or before this tile's first unit cut nothing here
      if (b < grid && start(b) < end) {                                // the first boundary strictly inside this tile
        const int i = cuts.n++;
        cuts.b[i] = (unsigned char)b; cuts.prob[i] = (unsigned char)prob; cuts.r[i] = (unsigned char)r; cuts.c[i] = (unsigned char)c;
        int cnt = 0;
        while (b < grid && start(b) < end) { ++b; ++cnt; }
        cuts.cnt[i] = (unsigned char)cnt;
      }
      ubase = end;
      if (!sk_next(gp.M / T2, gp.N / T2, gp.sym != 0, r, c)) {
        if (++prob >= n) break;
        r = c = 0;
      }
    }
    if (cuts.n) hipLaunchKernelGGL(k_hgemm_sk_fix, dim3(cuts.n * 32), dim3(kThreads2), 0, st, a, cuts);
  }
  return (int)hipGetLastError();
}

static int launch_hgemm(const uint16_t* A, long lda, const uint16_t* B, long ldb, void* C, long ldc, int c_bf16,
                        int c_trans, int M, int N, int K, int kmode, hipStream_t st, int sym = 0) {
  HGemmArgs g = {A, lda, B, ldb, C, ldc, c_bf16, c_trans, M, N, K, kmode, sym};
  return launch_hgemm_args(g, st);
}

static int device_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 1;
    n = v;
  }
  return n;
}

// T3' = (Ql T2)' and out = Ql' T3 in one launch (k_hgemm_tri_pair_256); false if the shape contract does not hold
static bool pair_legal(int M, int N) {
  if (g_hgemm_variant != 0) return false;
  if ((M % T2) || (N % T2)) return false;
  const int tiles = (M / T2) * (N / T2);
  // a launch takes whole tile columns (hand-offs stay inside a column) and must be fully resident, one block per CU
  // Round 2 (tools/bf16_pair_min_ab.py): the fused pairs win from 16 tiles on (2560^2 0.21 -> 0.16 ms, 1024 x 2048 0.136 ->
  // 0.111, 256 x 4096 0.171 -> 0.133; equal at 1024^2, slower at 768^2 and below); round 1 had put the threshold at 128.
  static const int min_tiles = getenv("PSGD_BF16_PAIR_MIN_TILES") ? atoi(getenv("PSGD_BF16_PAIR_MIN_TILES")) : 16;   // (env: A/B runs)
  return tiles >= min_tiles && (M / T2) <= device_cu_count();
}

// Q, Qt: the triangular factor and its transpose (dimension Mk); B1: the other operand [Nk][Mk]; T3: [Nk][Mk] hand-off
// buffer; the result tile (rows = factor dimension) is stored plain or transposed.
// Returns 0 on success, 1 on a launch error, 2 if the grid cannot be co-resident (the caller then runs the two
// products separately).  The kernel's hand-offs need every block resident: one 128-KiB-LDS block per CU, so the grid
// must not exceed the CU count and the occupancy query must admit the block (checked once; a cooperative launch
// would repeat that check on every call for +15-19 us of host time).
static int launch_tri_pair(const HWs& k, const uint16_t* Q, const uint16_t* Qt, const uint16_t* B1, uint16_t* T3, void* out,
                           long ldo, int out_trans, int Mk, int Nk, hipStream_t st, int flag_set = 0) {
  static int blocks_per_cu = -1;
  if (blocks_per_cu < 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&k_hgemm_tri_pair_256<0>), kThreads2, 0) !=
        hipSuccess) n = 0;
    blocks_per_cu = n;
  }
  const int tiles_m = Mk / T2, tiles_n = Nk / T2;
  if (blocks_per_cu < 1 || tiles_m > device_cu_count()) return 2;
  int cpl = device_cu_count() / tiles_m;                 // tile columns per launch
  if (cpl >= 8) cpl = cpl / 8 * 8;                       // whole columns per XCD
  if (cpl > tiles_n) cpl = tiles_n;
  // the hand-off words were zeroed by k_factors_to_bf16 earlier on this stream
  for (int c0 = 0; c0 < tiles_n; c0 += cpl) {
    const int cc = (tiles_n - c0 < cpl) ? tiles_n - c0 : cpl;
    HPairArgs p = {Q, Mk, B1, Mk, T3, Mk, Qt, Mk, out, ldo, 1, out_trans, Mk, Nk, c0, cc,
                   k.flags + 4 + flag_set * tiles_m * tiles_n, k.flags, g_spin_limit,
                   g_pair_xcd_patch >= 0 ? g_pair_xcd_patch : (tiles_m >= 32 ? 8 : 0)};
    hipLaunchKernelGGL(k_hgemm_tri_pair_256<0>, dim3(tiles_m * cc), dim3(kThreads2), 0, st, p);
    if (hipGetLastError() != hipSuccess) return 1;
  }
  return 0;
}

static int launch_cvt(const void* src, int src_bf16, long lds_, uint16_t* dst, long ldd, int rows, int cols,
                      int transpose, hipStream_t st, uint16_t* dstT = nullptr, long lddT = 0) {      // dstT: also the transposed copy (transpose = 0)
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (src_bf16) hipLaunchKernelGGL((k_to_bf16<true>), grid, dim3(kThreads), 0, st, src, lds_, dst, ldd, rows, cols, transpose, dstT, lddT);
  else hipLaunchKernelGGL((k_to_bf16<false>), grid, dim3(kThreads), 0, st, src, lds_, dst, ldd, rows, cols, transpose, dstT, lddT);
  return (int)hipGetLastError();
}

// dst (fp32) = src (bf16), contiguous; n multiple of 8
// part (or null): block b leaves max|x| of what it converted in part[b] (NaN propagates) -- the partial maxima the split launch of the
// inverse-route solves wants (psgd_kron.hip k_absmax), without a second pass over the fp32 copy
__global__ __launch_bounds__(kThreads) void k_bf16_to_f32(const uint16_t* __restrict__ src, float* __restrict__ dst, long n8,
                                                          float* __restrict__ part) {
  __shared__ float red[kThreads / 64];
  float m = 0.0f;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n8; i += (long)gridDim.x * kThreads) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + i * 8);
    f32x4 lo, hi;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      lo[2 * e] = __uint_as_float(v[e] << 16);         lo[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
      hi[2 * e] = __uint_as_float(v[2 + e] << 16);     hi[2 * e + 1] = __uint_as_float(v[2 + e] & 0xffff0000u);
    }
    *reinterpret_cast<f32x4*>(dst + i * 8) = lo;
    *reinterpret_cast<f32x4*>(dst + i * 8 + 4) = hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) m = amaxf(amaxf(m, fabsf(lo[e])), fabsf(hi[e]));
  }
  if (!part) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = amaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) m = amaxf(m, red[w]);
    part[blockIdx.x] = m;
  }
}

// Workspace of the bf16-operand update.  fp32: the balanced factors, the solve chain, the two max words.
// bf16: both orientations of the balanced factors, T' = (dG QrS')', the K-concatenated Gram operands
// W1 = [Bt | 0 | A] (M rows) and W2 = [Bt' | 0 | A'] (N rows), and the two triangular gradients.
struct HUpdWs {
  float *scal, *QlS, *QrS, *X0, *X1, *Bt, *dinv;
  void* inv_ws;             // the solves through explicit inverses (kron_shared.h), null when the route does not apply
  uint16_t *Qlb, *QlTb, *Qrb, *QrTb, *Tt, *W1, *W2, *g1, *g2;
  float* skp;               // stream-K partial tiles (null when M or N is not a multiple of 256)
  int n64, m64;             // column offset of the A part in W1 (N rounded up to the K tile) / of A' in W2
  int ld1, ld2;             // row strides of W1 / W2: n64 + N (m64 + M) plus a pad that keeps them off multiples of 4 KiB
  int64_t w1_bytes, w2_bytes, total;
};

static HUpdWs hupd_layout(char* base, int M, int N) {
  HUpdWs k;
  const int64_t mm = (int64_t)M * M, nn = (int64_t)N * N, mn = (int64_t)M * N;
  k.n64 = (N + TK - 1) / TK * TK; k.m64 = (M + TK - 1) / TK * TK;
  int64_t off = 0;
  auto takef = [&](int64_t elems) { float* p = reinterpret_cast<float*>(base + off); off = align256(off + elems * 4); return p; };
  auto takeh = [&](int64_t elems) { uint16_t* p = reinterpret_cast<uint16_t*>(base + off); off = align256(off + elems * 2); return p; };
  k.scal = takef(64);
  k.QlS = takef(mm); k.QrS = takef(nn); k.X0 = takef(mn); k.X1 = takef(mn); k.Bt = takef(mn);
  k.dinv = takef((int64_t)((M + 31) / 32 + (N + 31) / 32) * 1024);
  k.Qlb = takeh(mm); k.QlTb = takeh(mm); k.Qrb = takeh(nn); k.QrTb = takeh(nn);
  k.Tt = takeh(mn);
  // (a row stride that is a multiple of 4 KiB puts the rows a DMA instruction touches on few memory channels: +64 elements)
  static const int pad = getenv("PSGD_SK_PAD") ? atoi(getenv("PSGD_SK_PAD")) : 64;
  k.ld1 = k.n64 + N; k.ld2 = k.m64 + M;
  if (k.ld1 % 2048 == 0) k.ld1 += pad;
  if (k.ld2 % 2048 == 0) k.ld2 += pad;
  k.w1_bytes = (int64_t)M * k.ld1 * 2; k.w2_bytes = (int64_t)N * k.ld2 * 2;
  k.W1 = takeh(k.w1_bytes / 2); k.W2 = takeh(k.w2_bytes / 2);
  k.g1 = takeh(mm); k.g2 = takeh(nn);
  k.skp = nullptr;
  if (M % T2 == 0 && N % T2 == 0) {
    // two partial-tile slots per workgroup of the gradient launch: at most kSkMaxGrid workgroups, never more than half its units
    const int64_t tm = M / T2, tn = N / T2;
    const int64_t units = tm * (tm + 1) / 2 * ((k.n64 + N) / TK) + tn * (tn + 1) / 2 * ((k.m64 + M) / TK);
    const int64_t g = units / 2 < kSkMaxGrid ? (units / 2 > 0 ? units / 2 : 1) : kSkMaxGrid;
    k.skp = takef(2 * g * 65536);
  }
  const int64_t ib = psgdk::kron_inv_solves_bytes(M, N);       // (a function of the shape)
  k.inv_ws = ib > 0 ? static_cast<void*>(base + off) : nullptr;
  off = align256(off + ib);
  k.total = off;
  return k;
}

// zero_flags: the launch also zeroes the hand-off flags of the fused pairs (the apply behind it then needs no memset launch)
static int launch_factors_cvt(const HWs& k, FactorJob j0, FactorJob j1, hipStream_t st, bool zero_flags = false) {
  const int n = j0.n > j1.n ? j0.n : j1.n;
  dim3 grid((n + 63) / 64, (n + 63) / 64, 2);
  hipLaunchKernelGGL(k_factors_to_bf16, grid, dim3(kThreads), 0, st, j0, j1, zero_flags ? k.flags + 4 : static_cast<unsigned*>(nullptr),
                     zero_flags ? (int)((k.flag_bytes - 16) / 4) : 0);
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
  if (key == 1) { g_two_pairs = value; return PSGD_OK; }
  if (key == 3) { g_trsm_lite = value; return PSGD_OK; }
  if (key == 4) { g_streamk = value; return PSGD_OK; }
  if (key == 5) { g_pair_patch = value; return PSGD_OK; }
  if (key == 6) { g_pair_merge_whatif = value; return PSGD_OK; }
  if (key == 7) { g_pair_xcd_patch = (value < 0 || value > 16) ? -1 : value; return PSGD_OK; }
  if (key == 2) { g_spin_limit = (value < 0 || value > 30) ? (1u << 22) : (1u << value); return PSGD_OK; }
  return PSGD_ERR_BAD_ARG;
}

int psgd_kron_bf16_handoff_timeouts(const void* ws, int M, int N) {
  if (!ws || M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  HWs k = hws_layout(static_cast<char*>(const_cast<void*>(ws)), M, N);
  unsigned v = 0;
  if (hipMemcpy(&v, k.flags, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return PSGD_ERR_LAUNCH;
  return (int)v;
}

int64_t psgd_kron_bf16_handoff_counter_offset(int M, int N) {
  if (M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  HWs k = hws_layout(static_cast<char*>(nullptr), M, N);
  return (int64_t)(reinterpret_cast<char*>(k.flags) - static_cast<char*>(nullptr));
}

int psgd_kron_bf16_handoff_reset(void* ws, int M, int N, void* stream) {
  if (!ws || M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  HWs k = hws_layout(static_cast<char*>(ws), M, N);
  if (hipMemsetAsync(k.flags, 0, 16, static_cast<hipStream_t>(stream)) != hipSuccess) return PSGD_ERR_LAUNCH;
  return PSGD_OK;
}

int64_t psgd_kron_dd_workspace_bytes_bf16(int M, int N) {
  if (M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  return hws_layout(nullptr, M, N).total;
}

static int bf16_apply_check(int M, int N, void* ws, int64_t ws_bytes) {
  if (M <= 0 || N <= 0 || (M % 8) || (N % 8)) return PSGD_ERR_SHAPE;   // 16-byte bf16 chunks along K
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < hws_layout(nullptr, M, N).total)
    return PSGD_ERR_WORKSPACE;
  return PSGD_OK;
}

/* bf16 copies of the fp32 master factors, plain and transposed, into the workspace (one launch; only the 256-blocks on
 * or above the diagonal are touched).  They change only when the factors do, so a caller that applies the same
 * factors repeatedly prepares once and then calls psgd_kron_dd_apply_bf16_prepared. */
int psgd_kron_bf16_prepare_factors(const float* Ql, const float* Qr, int M, int N, void* ws, int64_t ws_bytes,
                                   void* stream) {
  if (!Ql || !Qr) return PSGD_ERR_BAD_ARG;
  const int rc = bf16_apply_check(M, N, ws, ws_bytes);
  if (rc) return rc;
  HWs k = hws_layout(static_cast<char*>(ws), M, N);
  HK(launch_factors_cvt(k, FactorJob{Qr, k.Qr, k.QrT, N}, FactorJob{Ql, k.Ql, k.QlT, M}, static_cast<hipStream_t>(stream)));
  return PSGD_OK;
}

static int apply_prepared_impl(const void* G, void* out, int M, int N, void* ws, int64_t ws_bytes, void* stream, bool flags_zeroed);
int psgd_kron_dd_apply_bf16_prepared(const void* G, void* out, int M, int N, void* ws, int64_t ws_bytes, void* stream) {
  return apply_prepared_impl(G, out, M, N, ws, ws_bytes, stream, false);
}
static int apply_prepared_impl(const void* G, void* out, int M, int N, void* ws, int64_t ws_bytes, void* stream, bool flags_zeroed) {
  if (!G || !out) return PSGD_ERR_BAD_ARG;
  const int rc = bf16_apply_check(M, N, ws, ws_bytes);
  if (rc) return rc;
  if ((reinterpret_cast<uintptr_t>(G) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  HWs k = hws_layout(static_cast<char*>(ws), M, N);
  const uint16_t* Gb = static_cast<const uint16_t*>(G);
  // hand-off flags of the fused pairs: zero before every call (the sticky word in front of them stays)
  // (flags_zeroed: the conversion launch of psgd_kron_dd_apply_bf16 did it)
  if (!flags_zeroed && hipMemsetAsync(k.flags + 4, 0, (size_t)k.flag_bytes - 16, st) != hipSuccess) return PSGD_ERR_LAUNCH;
  if (pair_legal(M, N) && g_two_pairs) {
    // Two fused triangular pairs and no Gram:  out = Ql' (Ql ((G Qr') Qr)).  Same product as psgd.py:189-192 with the
    // Gram Qr'Qr (resp. Ql'Ql) re-associated into the chain: (G Qr') Qr costs the flops of the dense G (Qr'Qr) alone,
    // and both halves have the complementary-K structure the wavefront kernel balances.  bf16 path only (one bf16
    // rounding moves from the Gram to G Qr'); the fp32 path keeps the reference's association order.
    if (g_pair_merge_whatif && (N / T2) * (M / T2) <= device_cu_count()) {      // WHAT-IF (timing only): see k_hgemm_tri_pair2_256
      const int tm0 = N / T2, tn0 = M / T2, tm1 = M / T2, tn1 = N / T2;
      HPairArgs p0 = {k.Qr, N, Gb, N, k.T3, N, k.QrT, N, k.T2, M, 1, 0, N, M, 0, tn0, k.flags + 4, k.flags, g_spin_limit};
      HPairArgs p1 = {k.Ql, M, k.T2, M, k.GT, M, k.QlT, M, out, N, 1, 0, M, N, 0, tn1, k.flags + 4 + tm0 * tn0, k.flags, g_spin_limit};    // (GT: free on this route)
      hipLaunchKernelGGL(k_hgemm_tri_pair2_256<0>, dim3(tm0 * tn0 + tm1 * tn1), dim3(kThreads2), 0, st, p0, p1, tm0 * tn0);
      return hipGetLastError() == hipSuccess ? PSGD_OK : PSGD_ERR_LAUNCH;
    }
    // right pair: Y' [N][M] = ((G Qr') Qr)'   -- triangular factor Qr as the A operand, G [M][N] the K-contiguous B operand
    int rc1 = launch_tri_pair(k, k.Qr, k.QrT, Gb, k.T3, k.T2, M, 0, N, M, st, 0);
    // left pair: out [M][N] = Ql' (Ql Y)      -- B operand Y' [N][M]
    int rc2 = rc1 ? rc1 : launch_tri_pair(k, k.Ql, k.QlT, k.T2, k.T3, out, N, 0, M, N, st, 1);
    if (rc1 == 0 && rc2 == 0) return PSGD_OK;
    if (rc1 == 1 || rc2 == 1) return PSGD_ERR_LAUNCH;
    // not resident-able (2): fall through to the staged chain below, which recomputes everything from G (a first pair
    // that did run only wrote the scratch buffers T2, T3)
  }
  if (M < N) {                                                                     // psgd.py:189-190
    HK(launch_cvt(G, 1, N, k.GT, M, M, N, 1, st));
    // T1 = Ql'Ql              A = Ql' [M][K=M], Bt = Ql' ; k <= min(m, n); symmetric: upper tiles computed, stored twice
    HK(launch_hgemm(k.QlT, M, k.QlT, M, k.T1, M, 1, 0, M, M, M, KHI_M | KHI_N, st, 1));
    // T2 = T1 G               A = T1 [M][K=M], Bt = G' [N][M]
    HK(launch_hgemm(k.T1, M, k.GT, M, k.T2, N, 1, 0, M, N, M, 0, st));
    // fused pair with the roles of the operands swapped: (T2 Qr')' = Qr T2' and (T3 Qr)' = Qr' T3', i.e. the
    // triangular factor is the A operand again, T2 [M][N] and T3 [M][N] are the K-contiguous B operands, and the
    // result tile is stored transposed
    int fused = pair_legal(N, M) ? launch_tri_pair(k, k.Qr, k.QrT, k.T2, k.T3, out, N, 1, N, M, st) : 2;
    if (fused == 1) return PSGD_ERR_LAUNCH;
    if (fused == 2) {
      // T3 = T2 Qr'             A = T2 [M][K=N], Bt[n][k] = Qr'[k][n] = Qr[n][k] ; k >= n
      HK(launch_hgemm(k.T2, N, k.Qr, N, k.T3, N, 1, 0, M, N, N, KLO_N, st));
      // out = T3 Qr             Bt[n][k] = Qr[k][n] = Qr'[n][k] ; k <= n
      HK(launch_hgemm(k.T3, N, k.QrT, N, out, N, 1, 0, M, N, N, KHI_N, st));
    }
  } else {                                                                         // psgd.py:191-192
    // T1 = Qr'Qr  (symmetric, so it is its own Bt layout; upper tiles computed, stored twice)
    HK(launch_hgemm(k.QrT, N, k.QrT, N, k.T1, N, 1, 0, N, N, N, KHI_M | KHI_N, st, 1));
    // T2 = G T1               A = G [M][K=N], Bt = T1' = T1 ; stored transposed: T2' [N][M]
    HK(launch_hgemm(Gb, N, k.T1, N, k.T2, M, 1, 1, M, N, N, 0, st));
    int fused = pair_legal(M, N) ? launch_tri_pair(k, k.Ql, k.QlT, k.T2, k.T3, out, N, 0, M, N, st) : 2;
    if (fused == 1) return PSGD_ERR_LAUNCH;
    if (fused == 2) {                                      // not fused: the two products as separate launches
      // T3 = Ql T2              A = Ql [M][K=M] (k >= m), Bt = T2' ; stored transposed: T3' [N][M]
      HK(launch_hgemm(k.Ql, M, k.T2, M, k.T3, M, 1, 1, M, N, M, KLO_M, st));
      // out = Ql' T3            A = Ql' [M][K=M] (k <= m), Bt = T3'
      HK(launch_hgemm(k.QlT, M, k.T3, M, out, N, 1, 0, M, N, M, KHI_M, st));
    }
  }
  return PSGD_OK;
}

/* _precond_grad_dense_dense with bf16 operands: prepare the factor copies, then apply. */
int psgd_kron_dd_apply_bf16(const float* Ql, const float* Qr, const void* G, void* out, int M, int N, void* ws,
                            int64_t ws_bytes, void* stream) {
  if (!Ql || !Qr || !G || !out) return PSGD_ERR_BAD_ARG;
  // new factors on every call (what BASELINE's second metric times): the conversion launch also zeroes the pairs' hand-off flags
  const int rc = bf16_apply_check(M, N, ws, ws_bytes);
  if (rc) return rc;
  HWs k = hws_layout(static_cast<char*>(ws), M, N);
  HK(launch_factors_cvt(k, FactorJob{Qr, k.Qr, k.QrT, N}, FactorJob{Ql, k.Ql, k.QlT, M}, static_cast<hipStream_t>(stream), true));
  return apply_prepared_impl(G, out, M, N, ws, ws_bytes, stream, true);
}

int64_t psgd_kron_dd_update_workspace_bytes_bf16(int M, int N) {
  if (M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  return hupd_layout(nullptr, M, N).total;
}

/* update_precond_kron dense (x) dense (psgd.py:160-180) with bf16 matrix-core operands.  fp32 master factors in and
 * out; dX, dG arrive in bf16.  What stays fp32: the balance (:166-170), the triangular solves of :174 (fp32 strip
 * substitutions on fp32 inverted diagonal blocks; the trailing products BETWEEN strips run on bf16 x 3 splits of fp32
 * operands with the three leading terms kept -- 2^-16 relative per product, 128x finer than the bf16 rounding of dX
 * itself; psgd_kron_bf16_set_tuning(3, 0) keeps all six terms), the max-norms, the step sizes and the final
 * subtraction Q - step * grad * Q.
 * What runs on bf16 operands with fp32 accumulation: A = QlS dG QrS' (:173), the four Grams of :175-176 (two launches:
 * each gradient is ONE symmetric product over a concatenated K axis, [Bt | A][Bt | A]' with the accumulators negated
 * between the halves) and grad * Q of :179-180. */
int psgd_kron_dd_update_bf16(const float* Ql, const float* Qr, const void* dX, const void* dG, float* QlOut,
                             float* QrOut, int M, int N, float step, float tiny, void* ws, int64_t ws_bytes,
                             void* stream) {
  if (!Ql || !Qr || !dX || !dG || !QlOut || !QrOut) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0 || (M % 8) || (N % 8)) return PSGD_ERR_SHAPE;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < hupd_layout(nullptr, M, N).total)
    return PSGD_ERR_WORKSPACE;
  if ((reinterpret_cast<uintptr_t>(dX) & 15) || (reinterpret_cast<uintptr_t>(dG) & 15)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  HUpdWs k = hupd_layout(static_cast<char*>(ws), M, N);
  const uint16_t* dGb = static_cast<const uint16_t*>(dG);
  const int ld1 = k.ld1, ld2 = k.ld2;
  // the zero columns between the two halves of the concatenated K axis (only when N or M is not a K-tile multiple)
  if (k.n64 != N && hipMemsetAsync(k.W1, 0, (size_t)k.w1_bytes, st) != hipSuccess) return PSGD_ERR_LAUNCH;
  if (k.m64 != M && hipMemsetAsync(k.W2, 0, (size_t)k.w2_bytes, st) != hipSuccess) return PSGD_ERR_LAUNCH;
  const bool inv_route = k.inv_ws && psgdk::kron_inv_solves_on(M, N);
  const bool fused = inv_route && psgdk::kron_fused_prologue_on(M, N);
  // (the two max words: zeroed by the fused prologue's first launch, else by a memset -- a 12-us launch of its own on this stream)
  if (!fused && hipMemsetAsync(k.scal, 0, 256, st) != hipSuccess) return PSGD_ERR_LAUNCH;
  // :166-170 (+ the solves' inverted diagonal blocks; on the inverse route also the zeroed plane metas and the factors' partial maxima)
  // (round 6) on the tile-scale inverse route: rho + ONE sweep that also writes the factors' column-form planes for the solves
  if (fused) HK(psgdk::kron_balance_planes(Ql, Qr, M, N, k.QlS, k.QrS, st, k.dinv, k.inv_ws, k.scal));
  else HK(psgdk::kron_balance(Ql, Qr, M, N, k.QlS, k.QrS, st, nullptr, k.dinv, inv_route ? k.inv_ws : nullptr));
  // the bf16 products of :173 go to the side stream (kron_shared.h), the fp32 solves of :174 stay on the caller's
  psgdk::KronFork* fk = psgdk::kron_overlap_chains(M, N) ? psgdk::kron_fork(st) : nullptr;
  psgdk::KronForkScope fork_scope(fk, st);   // joins on every exit path, early error returns included
  hipStream_t sf = fk ? fk->side : st;
  // the bf16 products of psgd.py:173 (side stream)
  auto products = [&]() -> int {
    {
      dim3 grid((max(M, N) + 63) / 64, (max(M, N) + 63) / 64, 2);
      hipLaunchKernelGGL(k_factors_to_bf16, grid, dim3(kThreads), 0, sf, FactorJob{k.QlS, k.Qlb, k.QlTb, M},
                         FactorJob{k.QrS, k.Qrb, k.QrTb, N}, static_cast<unsigned*>(nullptr), 0);
      HK((int)hipGetLastError());
    }
    // T' [N][M] = (dG QrS')'      A = dG [M][K=N], Bt[n][k] = QrS[n][k], k >= n                      (:173)
    // (these two run BESIDE the fp32 solves' plane products: a 128-KiB-LDS workgroup only starts on a CU the other chain has left
    //  entirely and then keeps it for its whole range -- as stream-K launches they took 291 + 340 us instead of 178 + 166 and
    //  stretched the solves' splits from 70-90 to 180 us: they stay on the 64-KiB one-tile kernel, profiles/r04_streamk_ab.txt.
    //  Also measured: the same launches on 64 / 128 / 192 workgroups only, on a third stream from the start of the call -- half
    //  the CUs stay free for the inversions' launch-bound chains and the solves then run alone: 1.75 / 1.72 / 1.73 ms against
    //  1.70 -- the chains stretch by what the solves gain, 470 us of inversions instead of 270)
    HK(launch_hgemm(dGb, N, k.Qrb, N, k.Tt, M, 1, 1, M, N, N, KLO_N, sf));
    // A = QlS T  -> second half of W1     A operand QlS [M][K=M], k >= m; Bt = T'
    HK(launch_hgemm(k.Qlb, M, k.Tt, M, k.W1 + k.n64, ld1, 1, 0, M, N, M, KLO_M, sf));
    HK(launch_cvt(k.W1 + k.n64, 1, ld1, k.W2 + k.m64, ld2, M, N, 1, sf));              // A' -> second half of W2
    return PSGD_OK;
  };
  const bool inv_first = inv_route && fk && psgdk::kron_inv_first(M, N);
  if (!inv_first) { const int rc = products(); if (rc) return rc; }
  // Bt = QlS^-T dX QrS^-1 in fp32                                                                   (:174)
  int x0_parts = 0;
  // (round 6, fused prologue) dX's conversion and planes on the third stream: X1 needs them only behind Qr's inversion
  const bool x0_bg = fused && fk && fk->bg && inv_first;
  hipStream_t sx = st;
  if (x0_bg) { HK(psgdk::kron_fork_bg(fk)); sx = fk->bg; }
  {
    const long n8 = (long)M * N / 8;
    int grid = (int)((n8 + kThreads - 1) / kThreads);
    x0_parts = (inv_route && !fused) ? psgdk::kron_inv_part_max() : 0;   // (the inverse route: dX's partial maxima from this launch)
    if (grid > (x0_parts ? x0_parts : 4096)) grid = x0_parts ? x0_parts : 4096;
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(grid), dim3(kThreads), 0, sx, static_cast<const uint16_t*>(dX), k.X0, n8,
                       x0_parts ? psgdk::kron_inv_part(k.inv_ws, M, N) : static_cast<float*>(nullptr));
    HK((int)hipGetLastError());
    if (x0_parts) x0_parts = grid;
  }
  if (inv_route) {
    // factors from 2048 on: the solves as products with the inverses of the diagonal 2048-blocks (fp32-accurate f16 x 2 plane
    // products; psgd_kron.hip blk_solves_*).  Ql's inversion goes to the side stream: behind the bf16 products, or -- both factors
    // from 4096 on -- ahead of them (kron_inv_first), the products then running beside X1 and Bt.
    const float* dinv_l = k.dinv + (long)((N + 31) / 32) * 1024;
    if (inv_first) {
      HK(psgdk::kron_inv_solves_front(k.QlS, k.QrS, k.dinv, dinv_l, k.X0, k.X1, k.Bt, M, N, k.inv_ws, st, sf, fk->mid, true, x0_parts, fused, x0_bg ? fk->aux : nullptr,
                                      x0_bg ? fk->bg : nullptr));
      { const int rc = products(); if (rc) return rc; }
      if (hipStreamWaitEvent(st, fk->mid, 0) != hipSuccess) return PSGD_ERR_LAUNCH;
      HK(psgdk::kron_inv_solves_back(k.QlS, k.X1, k.Bt, M, N, k.inv_ws, st, fused));
      HK(fork_scope.join());
    } else {
      HK(psgdk::kron_inv_solves_front(k.QlS, k.QrS, k.dinv, dinv_l, k.X0, k.X1, k.Bt, M, N, k.inv_ws, st, sf, nullptr, true, x0_parts, fused));
      HK(fork_scope.join());
      HK(psgdk::kron_inv_solves_back(k.QlS, k.X1, k.Bt, M, N, k.inv_ws, st, fused));
    }
  } else {
    HK(psgdk::kron_trsm_ut(k.QrS, N, k.X0, k.X1, M, (long)N, 1L, k.dinv, st, g_trsm_lite, true));
    HK(psgdk::kron_trsm_ut(k.QlS, M, k.X1, k.Bt, N, 1L, (long)N, k.dinv + (long)((N + 31) / 32) * 1024, st, g_trsm_lite, true));
    HK(fork_scope.join());
  }
  HK(launch_cvt(k.Bt, 0, N, k.W1, ld1, M, N, 0, st, k.W2, ld2));                      // Bt -> first half of W1, Bt' -> first half of W2
  // grad1 = triu(A A' - Bt Bt'), grad2 = triu(A'A - Bt'Bt)                                          (:175-176)
  {
    HGemmArgs g = {k.W1, ld1, k.W1, ld1, k.g1, M, 1, 0, M, M, k.n64 + N, 0, 1};
    g.epi = HEPI_TRIU_MAX; g.kflip = k.n64 / TK; g.maxout = k.scal + 0;
    HGemmArgs h = {k.W2, ld2, k.W2, ld2, k.g2, N, 1, 0, N, N, k.m64 + M, 0, 1};
    h.epi = HEPI_TRIU_MAX; h.kflip = k.m64 / TK; h.maxout = k.scal + 1;
    const HGemmArgs two[2] = {g, h};
    const int rc = launch_hgemm_sk(two, 2, k.skp, st);
    if (rc == 2) HK(launch_hgemm_two(g, h, st)); else HK(rc);
  }
  // Ql_new = QlS - step / (max|grad1| + tiny) grad1 QlS, same for Qr                                (:177-180)
  {
    HGemmArgs g = {k.g1, M, k.QlTb, M, QlOut, M, 0, 0, M, M, M, KLO_M | KHI_N, 2};      // sym = 2: upper tiles + D copy below
    g.epi = HEPI_D_MINUS; g.D = k.QlS; g.ldd = M; g.scale_max = k.scal + 0; g.step = step; g.tiny = tiny;
    HGemmArgs h = {k.g2, N, k.QrTb, N, QrOut, N, 0, 0, N, N, N, KLO_M | KHI_N, 2};
    h.epi = HEPI_D_MINUS; h.D = k.QrS; h.ldd = N; h.scale_max = k.scal + 1; h.step = step; h.tiny = tiny;
    g.patch_order = h.patch_order = g_pair_patch;
    g.lower_zero = h.lower_zero = fused ? 1 : 0;
    // (as a stream-K launch, twice: 136 + 70 and, with the final fix-up launch, 136 + 53 us against 137 -- short K ranges, ~5 pieces per
    //  workgroup, nearly every tile cut, a D tile read per epilogue)
    HK(launch_hgemm_two(g, h, st));
  }
  return PSGD_OK;
}

}  // extern "C"
