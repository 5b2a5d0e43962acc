// psgd_kron.hip -- Kronecker dense (x) dense preconditioner on gfx950 (psgd.py:156-192).
//
//   update:  rho balance (:166-170), A = Ql (dG Qr') (:173), Bt = Ql^-T dX Qr^-1 (:174),
//            grad1 = triu(A A' - Bt Bt') (:175), grad2 = triu(A'A - Bt'Bt) (:176),
//            step_i = step / (max|grad_i| + tiny) (:177-178), Q_i - (step_i grad_i) Q_i (:179)
//   apply:   M <  N: (((Ql'Ql) G) Qr') Qr     (:190)
//            M >= N: Ql' (Ql (G (Qr'Qr)))     (:192)
//
// All products run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: bit-exact fp32 fma
// chains, so the numerics are those of an fp32 GEMM), with the elementwise steps of the
// reference fused into operand loads and epilogues:
//   * A A' - Bt Bt' is one launch (two K loops into one accumulator, second one negated),
//     its epilogue applies triu and reduces max|.| with an integer atomicMax on the
//     non-negative float bits (order independent, hence deterministic);
//   * Q - (step grad) Q scales the grad operand by step/(max+tiny) while staging it.
// The triangular solves are one blocked kernel: independent vectors across workgroups,
// 32-wide diagonal blocks solved by substitution in registers, trailing updates on MFMA
// with the K range split over the four waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include "psgd_hip.h"
#include "kron_shared.h"
#include "nanmax.h"
using psgd::amaxf;
using psgd::nmaxf;

namespace psgdk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

enum { EPI_STORE = 0, EPI_TRIU_MAX = 1, EPI_D_MINUS = 2 };

struct GemmArgs {
  const float* A; long a_rs, a_cs;      // A(m,k) = A[m*a_rs + k*a_cs]
  const float* B; long b_rs, b_cs;      // B(k,n) = B[k*b_rs + n*b_cs]
  const float* A2; long a2_rs, a2_cs;   // optional second pair, subtracted
  const float* B2; long b2_rs, b2_cs;
  int K2;
  float* C; long ldc;                   // C(m,n) = C[m*ldc + n*c_cs]  (c_cs = 0 means 1)
  int M, N, K;
  const float* D; long ldd;             // EPI_D_MINUS: C = D - acc, D(m,n) = D[m*ldd + n*c_cs]
  int epi;
  const float* scale_max;               // if set: A operand scaled by step / (scale_max[0] + tiny)
  float step, tiny;
  float* maxout;                        // EPI_TRIU_MAX: max|C| (atomicMax on int bits; zeroed by caller)
  // Triangular operands: products with an upper-triangular factor only need part of the K range.
  // klo/khi (per pass) restrict k to [lo, hi) with lo = row/col offset of the tile (rounded to BK):
  //   KLO_M: k >= m0 (A upper-triangular, A(m,k) = 0 for k < m)      KHI_M: k < m0 + BM (A = U', zero for k > m)
  //   KLO_N: k >= n0 (B(k,n) = 0 for k < n, e.g. B = U')             KHI_N: k < n0 + BN (B upper-triangular)
  int kmode, kmode2;
  int kblk;                             // KBLK_*: the size 2b of the diagonal blocks whose halves bound the K range (first pair only)
  long c_cs;
  const float* colv; int colsq;         // EPI_STORE: C(m,n) *= colv[n] (colsq: *= colv[n]^2)
  int sym;                              // C is symmetric (Gram X'X): tiles below the diagonal are skipped, the others stored twice
  int lite;                             // split GEMM: keep only h h' + h m' + m h' (2^-16 instead of 2^-24 relative per product)
};

// KBLK_HI_M: k < start of the second half of the kblk-block that holds row m0;  KBLK_LO_N: k >= start of the second half
// of the kblk-block that holds column n0  (the two products of one doubling level of a triangular inverse, see tri_inverse)
enum { KLO_M = 1, KHI_M = 2, KLO_N = 4, KHI_N = 8, KBLK_HI_M = 16, KBLK_LO_N = 32,
       KORD_PATCH = 64 };      // (not a K range: the tile order of gemm_tile_from_id's case 3)

// ---------------------------------------------------------------------------------------------
// fp32 GEMM on the matrix cores, one template for two square tile sizes T (64 for small problems
// and the batched launches, 128 when there are enough tiles to fill the chip):
//   T x T x 16 block tile, 4 waves (2 x 2), wave tile T/2 x T/2 = (T/32)^2 MFMA 16x16x4 tiles;
//   LDS double-buffered, the next K tile is prefetched into registers while the current one is
//   multiplied (one barrier per K tile); 16-byte global loads on interior tiles whenever the
//   operand's contiguous dimension allows it, guarded scalar loads on edges / odd strides;
//   LDS pitch T + 16 floats: the 4 k-rows of an MFMA operand read land 16 banks apart.

struct TileSrc {
  const float* P; long rs, cs;   // element (x, k) at P[x*rs + k*cs], x = m (A) or n (B)
  int X;                         // extent along x
};

// Fetch modes of an operand tile (x, k), compile-time per K loop (see f32_pass; the same reasons as in the split GEMM further
// down: a branch between load variants inside the K loop makes the compiler wait for every outstanding load at the join,
// which put A's memory latency in front of B's loads and both in front of the MFMAs -- ~3 us per K tile):
//   F32_KVEC  K-contiguous (cs == 1, 16-byte aligned rows): float4 along k; rows past the operand are CLAMPED to the last
//             one and zeroed at the commit, so blocks at the x edge stay on this mode;
//   F32_XVEC  X-contiguous (rs == 1), block inside the operand: float4 along x;
//   F32_ANY   anything else and ragged K tails: scalar loads from clamped addresses, zeroed at the commit.
// Nothing in g2r_* uses a loaded value.
enum { F32_ANY = 0, F32_KVEC = 1, F32_XVEC = 2 };

template <int T, int GK, int MODE>
__device__ __forceinline__ void g2r_tile(const TileSrc& t, int x0, int k0, int khi, float (&r)[T * GK / kThreads]) {
  const int tid = threadIdx.x;
  constexpr int NV = T * GK / (4 * kThreads);   // float4 per thread
  if (MODE == F32_KVEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int f = tid + kThreads * u, row = min(x0 + f / (GK / 4), t.X - 1), k4 = f % (GK / 4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(t.P + (long)row * t.rs + k0 + 4 * k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[4 * u + e] = v[e];
    }
  } else if (MODE == F32_XVEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int f = tid + kThreads * u, k = f / (T / 4), x4 = f % (T / 4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(t.P + (long)(k0 + k) * t.cs + x0 + 4 * x4);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[4 * u + e] = v[e];
    }
  } else {
#pragma unroll
    for (int u = 0; u < T * GK / kThreads; ++u) {
      const int e = tid + kThreads * u;
      int x, k;
      if (t.cs == 1) { k = e % GK; x = e / GK; } else { x = e % T; k = e / T; }
      r[u] = t.P[(long)min(x0 + x, t.X - 1) * t.rs + (long)min(k0 + k, khi - 1) * t.cs];
    }
  }
}

template <int T, int GK, int MODE>
__device__ __forceinline__ void r2s_tile(const TileSrc& t, int x0, int k0, int khi, float mul,
                                         const float (&r)[T * GK / kThreads], float (*S)[T + 16]) {
  const int tid = threadIdx.x;
  constexpr int NV = T * GK / (4 * kThreads);
  if (MODE == F32_KVEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int f = tid + kThreads * u, row = f / (GK / 4), k4 = f % (GK / 4);
      const float m = (x0 + row < t.X) ? mul : 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) S[4 * k4 + e][row] = r[4 * u + e] * m;
    }
  } else if (MODE == F32_XVEC) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int f = tid + kThreads * u, k = f / (T / 4), x4 = f % (T / 4);
      *reinterpret_cast<f32x4*>(&S[k][4 * x4]) = f32x4{r[4 * u] * mul, r[4 * u + 1] * mul, r[4 * u + 2] * mul, r[4 * u + 3] * mul};
    }
  } else {
#pragma unroll
    for (int u = 0; u < T * GK / kThreads; ++u) {
      const int e = tid + kThreads * u;
      int x, k;
      if (t.cs == 1) { k = e % GK; x = e / GK; } else { x = e % T; k = e / T; }
      S[k][x] = (x0 + x < t.X && k0 + k < khi) ? r[u] * mul : 0.0f;
    }
  }
}

template <int T>
__device__ __forceinline__ int f32_mode(const TileSrc& t, int x0) {
  const bool aligned = (reinterpret_cast<uintptr_t>(t.P) & 15) == 0;
  if (aligned && t.cs == 1 && (t.rs & 3) == 0) return F32_KVEC;
  if (aligned && t.rs == 1 && (t.cs & 3) == 0 && x0 + T <= t.X) return F32_XVEC;
  return F32_ANY;
}

template <int T, int GK>
struct GemmLds {
  float A[2][GK][T + 16];
  float B[2][GK][T + 16];
};

// Linear block id -> output tile.
// (1) L2 locality: blocks id, id + 8, ... run on one XCD (own 4 MiB L2).  When the tile grid is a multiple of 8 x 8,
//     every XCD works through 8 x 8 tile PATCHES: the 64 tiles that run together share 8 A panels and 8 B panels.
//     The large products are bound by L2-miss traffic (fp32 operands, 32 flop/B per 128^2 tile), not by the matrix
//     cores, so this is worth more than anything inside the tile.  Patches are dealt to the XCDs in serpentine order
//     (0..7, 7..0, ...): an XCD that got a long-K (or an all-zero, skipped) patch in one round gets the opposite in
//     the next, which balances the triangular K ranges and the triu outputs.
// (2) Longest K first otherwise: a triangular K range makes a tile's work depend on its row (KLO_M / KHI_M) or its
//     column (KLO_N / KHI_N); row-dependent work is sorted by the row-major order (reversed for the upper bounds),
//     column-dependent work walks the grid column-major (dG Qr' at 4096^2: 823 -> 485 us).
__device__ __forceinline__ void gemm_tile_from_id(int id, int ty, int tx, int kmode, int& by, int& bx) {
  const int nt = ty * tx;
  if ((kmode & KORD_PATCH) && (kmode & (KLO_M | KHI_N)) == (KLO_M | KHI_N) && ty == tx && ty % 4 == 0 && ty >= 32) {
    // (3') the same product in 4 x 4 tile PATCHES (tuning key 27).  Row order shares a row's A panel in the XCD's L2 but every
    // tile reads a B panel of its own: 30 % L2 hits on the 4096^2 factor updates, and the launch is bound by its longest tile's
    // chain of K steps, each a load that misses (profiles/r04_p3_update_pmc.txt).  The tiles of a patch share four A and four B
    // panels at K offsets a few steps apart.  Upper tiles first: the patches by decreasing distance from the diagonal (= work),
    // the tiles of a patch by decreasing K length; that list is dealt to the XCDs in runs of 16 tiles, serpentine per round of
    // 128 (within 6 % of the mean work at 32 x 32 tiles, 3 % at 48 x 48); then the tiles below the diagonal (copies of D).
    const int T = ty, P = T >> 2, ntu = T * (T + 1) / 2;
    if (id >= ntu) {
      const int q = id - ntu;                                                     // q-th tile below the diagonal, row by row
      int r = (int)((1.0f + sqrtf(1.0f + 8.0f * q)) * 0.5f);
      while (r * (r - 1) / 2 > q) --r;
      while ((r + 1) * r / 2 <= q) ++r;
      by = r; bx = q - r * (r - 1) / 2;
      return;
    }
    const int full = ntu & ~127, x = id & 7, j = id >> 3;
    int pos;
    if (j < (full >> 3)) {
      const int m = j >> 4, xx = (m & 1) ? 7 - x : x;
      pos = m * 128 + 16 * xx + (j & 15);
    } else {
      pos = id;                                                                   // (the last ntu % 128 tiles: one by one)
    }
    const int noff = 16 * (P * (P - 1) / 2);                                      // tiles of the patches off the diagonal
    // (r', c') of a patch's tiles by decreasing c' - r': (0,3) (0,2) (1,3) (0,1) (1,2) (2,3) (0,0) (1,1) (2,2) (3,3), then the six below
    const unsigned long long ord = 0xcd8e94fa50b61723ULL;                         // nibble t = (r' << 2) | c' of the t-th tile
    int pr, pc, t;
    if (pos < noff) {
      const int sidx = pos >> 4;
      t = pos & 15;
      int k = (int)((sqrtf(8.0f * sidx + 1.0f) - 1.0f) * 0.5f);                   // diagonal P - 1 - k: k (k + 1) / 2 patches before it
      while (k * (k + 1) / 2 > sidx) --k;
      while ((k + 1) * (k + 2) / 2 <= sidx) ++k;
      const int d = P - 1 - k;
      pr = sidx - k * (k + 1) / 2; pc = pr + d;
    } else {
      const int u = pos - noff;
      pr = pc = u / 10; t = u % 10;
    }
    const int nib = (int)((ord >> (4 * t)) & 15ULL);
    by = 4 * pr + (nib >> 2); bx = 4 * pc + (nib & 3);
    return;
  }
  if ((kmode & (KLO_M | KHI_N)) == (KLO_M | KHI_N) && !(kmode & (KLO_N | KHI_M)) && ty == tx && ty % 16 == 0) {
    // (3) K = [m0, n0 + T): a tile's work is its distance from the diagonal (the two factor updates, psgd.py:179).  8 x 8
    // patches give the XCD that owns the top-right corner 2.1x the mean work (1600 of 5984 K chunks at 32 x 32 tiles, and
    // blocks are bound to XCDs by id % 8).  Instead every XCD gets whole tile ROWS, dealt in serpentine order (rows x,
    // 15 - x, 16 + x, 31 - x, ...: within 9 % of the mean), and walks its rows from the right-hand column inwards, longest
    // K first; the tiles of a row share its A panel in the XCD's L2.
    const int xcd = id % 8, j = id / 8, rpx = ty / 8;
    const int q = j % rpx, jj = j / rpx;
    by = (q & 1) ? q * 8 + 7 - xcd : q * 8 + xcd;
    bx = tx - 1 - jj;
    return;
  }
  const bool by_col = (kmode & (KLO_N | KHI_N)) && !(kmode & (KLO_M | KHI_M));
  if (ty % 8 == 0 && tx % 8 == 0 && nt % 512 == 0) {
    const int xcd = id % 8, j = id / 8, pn = (by_col ? ty : tx) / 8;
    const int pl = j / 64, e = j % 64;
    const int pid = (pl & 1) ? pl * 8 + 7 - xcd : pl * 8 + xcd;
    const int major = (pid / pn) * 8 + e / 8, minor = (pid % pn) * 8 + e % 8;   // patches walk the work-sorted axis first
    by = by_col ? minor : major;
    bx = by_col ? major : minor;
  } else if (by_col) {
    bx = id / ty; by = id % ty;
  } else {
    by = id / tx; bx = id % tx;
  }
  if (kmode & (KHI_M | KHI_N)) { by = ty - 1 - by; bx = tx - 1 - bx; }
}

__device__ __forceinline__ void gemm_tile_order(int kmode, int& by, int& bx) {
  gemm_tile_from_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.y, gridDim.x, kmode, by, bx);
}

// C-tile epilogue shared by the GEMM bodies: acc[i][j][e] is C[m0 + wm*W + i*16 + (lane>>4)*4 + e][n0 + wn*W + j*16 + (lane&15)]
// (the 16x16 C/D register layout is the same for the f32 and the bf16 MFMA forms)
template <int T>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x4 (&acc)[T / 32][T / 32], int m0, int n0) {
  constexpr int W = T / 2, NT = T / 32;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;
  float vmax = 0.0f;
  const long ccs = g.c_cs ? g.c_cs : 1;
  // EPI_D_MINUS: every D element of the tile is requested BEFORE the first one is used (clamped addresses, no branches):
  // a load inside the store loop below costs one full memory latency per element -- 16 .. 64 of them in a row per block,
  // as much as the whole K loop of a K = 512 product.
  f32x4 dv[NT][NT];
  if (g.epi == EPI_D_MINUS) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = min(m0 + wm * W + i * 16 + (lane >> 4) * 4 + e, g.M - 1);
          const int col = min(n0 + wn * W + j * 16 + (lane & 15), g.N - 1);
          dv[i][j][e] = g.D[(long)row * g.ldd + col * ccs];
        }
  }
  float cvj[NT];                        // column scales (EPI_STORE with colv): one load per column of the wave tile, up front
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    cvj[j] = 1.0f;
    if (g.epi == EPI_STORE && g.colv) {
      const float cv = g.colv[min(n0 + wn * W + j * 16 + (lane & 15), g.N - 1)];
      cvj[j] = g.colsq ? cv * cv : cv;
    }
  }
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = m0 + wm * W + i * 16 + (lane >> 4) * 4 + e;
        const int col = n0 + wn * W + j * 16 + (lane & 15);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][e];
          if (g.epi == EPI_TRIU_MAX) {
            v = (col >= row) ? v : 0.0f;
            vmax = amaxf(vmax, fabsf(v));
          } else if (g.epi == EPI_D_MINUS) {
            v = dv[i][j][e] - v;
          } else if (g.colv) {
            v *= cvj[j];
          }
          g.C[(long)row * g.ldc + col * ccs] = v;
          if (g.sym && n0 > m0) g.C[(long)col * g.ldc + row * ccs] = v;      // the mirror tile is not computed
        }
      }
  if (g.epi == EPI_TRIU_MAX) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
    // integer max on the bits of a non-negative float: order-independent, and NaN (above +inf) propagates
    if (lane == 0 && __float_as_uint(vmax) != 0u) atomicMax(reinterpret_cast<int*>(g.maxout), __float_as_int(vmax));
  }
}

// n K tiles [k0 + GK t, +GK) of one operand pair into acc (LDS double-buffered: one barrier per K tile).  Enters and leaves
// with both LDS buffers free.
template <int T, int GK, int MA, int MB>
__device__ __forceinline__ void f32_pass(const TileSrc& ta, const TileSrc& tb, int m0, int n0, int k0, int khi, int n, float mul,
                                         GemmLds<T, GK>& L, f32x4 (&acc)[T / 32][T / 32]) {
  constexpr int W = T / 2, NT = T / 32;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;
  float ra[T * GK / kThreads], rb[T * GK / kThreads];
  g2r_tile<T, GK, MA>(ta, m0, k0, khi, ra);
  g2r_tile<T, GK, MB>(tb, n0, k0, khi, rb);
  r2s_tile<T, GK, MA>(ta, m0, k0, khi, mul, ra, L.A[0]);
  r2s_tile<T, GK, MB>(tb, n0, k0, khi, 1.0f, rb, L.B[0]);
  __syncthreads();
  for (int t = 0; t < n; ++t) {
    const int buf = t & 1;
    const int kn = k0 + min(t + 1, n - 1) * GK;          // the last iteration re-reads its own tile: no branch round the loads
    g2r_tile<T, GK, MA>(ta, m0, kn, khi, ra);
    g2r_tile<T, GK, MB>(tb, n0, kn, khi, rb);
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      const int kr = kk * 4 + (lane >> 4);
      float a[NT], b[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) a[i] = L.A[buf][kr][wm * W + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = L.B[buf][kr][wn * W + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < n) {
      r2s_tile<T, GK, MA>(ta, m0, kn, khi, mul, ra, L.A[buf ^ 1]);
      r2s_tile<T, GK, MB>(tb, n0, kn, khi, 1.0f, rb, L.B[buf ^ 1]);
    }
    __syncthreads();
  }
}

template <int T, int GK>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, int m0, int n0, GemmLds<T, GK>& L) {
  constexpr int NT = T / 32;   // MFMA tiles per wave edge
  // upper-triangular outputs: tiles strictly below the diagonal are all zero
  const bool tri_skip = (g.epi == EPI_TRIU_MAX || g.sym) && (m0 >= n0 + T);
  if (g.sym && tri_skip) return;        // written by the mirror tile's epilogue (a triu tile below the diagonal stores zeros instead)

  f32x4 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float a_mul = 1.0f;
  if (g.scale_max) a_mul = g.step / (g.scale_max[0] + g.tiny);

  // one operand pair at a time; the fetch modes are chosen once per block and pair, outside the K loops
#pragma unroll 1
  for (int p = 0; p < ((g.A2 && !tri_skip) ? 2 : (tri_skip ? 0 : 1)); ++p) {
    const TileSrc ta = p ? TileSrc{g.A2, g.a2_rs, g.a2_cs, g.M} : TileSrc{g.A, g.a_rs, g.a_cs, g.M};
    const TileSrc tb = p ? TileSrc{g.B2, g.b2_cs, g.b2_rs, g.N} : TileSrc{g.B, g.b_cs, g.b_rs, g.N};   // (n, k) view of B
    const int K = p ? g.K2 : g.K, km = p ? g.kmode2 : g.kmode;
    int lo = 0, hi = K;
    if (km & KLO_M) lo = max(lo, m0);
    if (km & KLO_N) lo = max(lo, n0);
    if (km & KHI_M) hi = min(hi, m0 + T);
    if (km & KHI_N) hi = min(hi, n0 + T);
    lo = (lo / GK) * GK;
    if (hi <= lo) continue;
    const int ntot = (hi - lo + GK - 1) / GK, nfull = (hi - lo) / GK;
    const float mul = p ? -a_mul : a_mul;
    int ma = f32_mode<T>(ta, m0), mb = f32_mode<T>(tb, n0);
    if (ma == F32_ANY || mb == F32_ANY) ma = mb = F32_ANY;
    const int e0 = (ma == F32_ANY) ? 0 : nfull;         // first K tile of the scalar loop (everything, or the ragged tail)
    if (e0 > 0) {
      if (ma == F32_KVEC && mb == F32_KVEC) f32_pass<T, GK, F32_KVEC, F32_KVEC>(ta, tb, m0, n0, lo, hi, e0, mul, L, acc);
      else if (ma == F32_KVEC) f32_pass<T, GK, F32_KVEC, F32_XVEC>(ta, tb, m0, n0, lo, hi, e0, mul, L, acc);
      else if (mb == F32_KVEC) f32_pass<T, GK, F32_XVEC, F32_KVEC>(ta, tb, m0, n0, lo, hi, e0, mul, L, acc);
      else f32_pass<T, GK, F32_XVEC, F32_XVEC>(ta, tb, m0, n0, lo, hi, e0, mul, L, acc);
    }
    if (ntot > e0) f32_pass<T, GK, F32_ANY, F32_ANY>(ta, tb, m0, n0, lo + e0 * GK, hi, ntot - e0, mul, L, acc);
  }
  gemm_epilogue<T>(g, acc, m0, n0);
}

// ---------------------------------------------------------------------------------------------
// fp32-accurate GEMM on the bf16 matrix cores ("bf16 x 3").  Every fp32 operand element is split on the way into
// LDS into three bf16 terms x = h + m + l (exactly: a truncating split, bf16 keeps fp32's exponent range), and a
// product a*b is accumulated as  h h' + h m' + m h' + h l' + l h' + m m'  with fp32 accumulation: six
// v_mfma_f32_16x16x32_bf16 at 16x the rate of v_mfma_f32_16x16x4_f32, i.e. up to 2.6x the fp32 matrix-core peak at
// fp32-level accuracy (the dropped terms are below 2^-24 relative, like one fp32 rounding).  Same operand views,
// K-range clipping, second (subtracted) operand pair and epilogues as gemm_body; 128 x 128 x 32 tile, one LDS
// buffer of 3 planes per operand (48 KiB, 2 blocks per CU), next K tile prefetched into registers.
typedef __bf16 bf16x8_k __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_k __attribute__((ext_vector_type(4)));
constexpr int kX3K = 32;

// Chunk swizzle of the 64-byte-row LDS images below (four 16-byte chunks per row, four rows per 256-byte bank line).
// ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), and
// a fragment read has row = lane & 15, chunk = lane >> 4: a group holds rows 0-3 and 12-15 at chunk c and rows 4-11 at
// chunk c + 1, so the XOR terms t[(row >> 2) & 3] must make t0, t3, t1 ^ 1, t2 ^ 1 distinct: {0, 3, 2, 1} = -(row >> 2) & 3.
// (Rounds 1-4 used (row >> 2) & 3, conflict-free for CONTIGUOUS 16-lane groups only: rows 0-3 / chunk 0 and rows 4-7 /
// chunk 1 shared their slots -- 8 LDS cycles per read instead of 4, SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE.)
#ifndef PSGD_LDS_SWZ_OLD
__device__ __forceinline__ int lds_swz(int row) { return (0 - (row >> 2)) & 3; }
#else
__device__ __forceinline__ int lds_swz(int row) { return (row >> 2) & 3; }
#endif

struct GemmLdsX3 {
  u32x4_k P[2][3][128 * 4];    // [A|B][plane][row*4 + (chunk ^ lds_swz(row))], a chunk = 8 consecutive k as bf16
};

// Two fp32 values -> three packed bf16 pairs (plane h, m, l; low half = x0).  The split TRUNCATES: h = the top 16 bits
// of x, r1 = x - h (exact), m = the top 16 bits of r1, l = r1 - m, which has at most 8 significant bits and is a bf16
// exactly -- so x = h + m + l holds exactly (8 + 8 + 8 mantissa bits), the dropped product terms are the same
// 2^-24-relative ones as with a round-to-nearest split, and the whole thing is 4 full-rate VALU ops per element plus
// one v_perm_b32 per pair and plane (the first version used three quarter-rate v_cvt_pk_bf16_f32 per element).
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned (&pk)[3]) {
  const unsigned top = 0xFFFF0000u;
  const float h0 = __uint_as_float(__float_as_uint(x0) & top), h1 = __uint_as_float(__float_as_uint(x1) & top);
  const float r0 = x0 - h0, r1 = x1 - h1;
  const float m0 = __uint_as_float(__float_as_uint(r0) & top), m1 = __uint_as_float(__float_as_uint(r1) & top);
  const float s0 = r0 - m0, s1 = r1 - m1;
  pk[0] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);   // (x0 >> 16) | (x1 & top)
  pk[1] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
  pk[2] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// Operand tile -> registers -> LDS planes.  Whatever the memory layout, a thread ends up with 16 consecutive k of ONE
// row of the (x, k) tile, so the LDS side is always two 16-byte chunk stores per plane.  The layout is a COMPILE-TIME
// mode of the K loop (x3_pass below is instantiated per mode pair and chosen once per block and operand pair):
//   X3_KVEC  K-contiguous operands (cs == 1): row = tid / 2, k half = tid % 2, four float4 loads;
//   X3_XROW  X-contiguous operands (rs == 1): row = tid % 128, k half = tid / 128, sixteen 4-byte loads, each coalesced
//            across the wave (lane -> row);
//   X3_EDGE  anything else and edge tiles: the first mapping with scalar loads from CLAMPED addresses; the elements
//            outside the operand are zeroed when the tile is committed to LDS.
// The fetch is straight-line code and never USES a loaded value (no scaling, no select): the loads of K tile t + 1 are
// issued ahead of the MFMAs of tile t, and any use -- or a branch between load variants, which makes the compiler wait
// at the join -- would put a full memory latency in front of the matrix work.  (Round 1 scaled the A operand in the
// fetch and chose the variant inside the loop: 34 % matrix-pipe occupancy)
enum { X3_EDGE = 0, X3_KVEC = 1, X3_XROW = 2 };

// Addresses are a UNIFORM base (scalar registers, advanced per K tile) plus a 32-bit per-lane byte offset that does not
// depend on the K tile: one VGPR per operand instead of a 64-bit address per load kept live across the loop.
// Blocks at the x edge of an operand (x0 + 128 > X) stay on the fast modes: the lane's ROW is clamped to the last one
// (a duplicate load), and the commit zeroes what lies outside -- only a ragged K tail needs the scalar mode.  (Edge blocks
// used to run every K step on clamped scalar loads: shapes that are not tile multiples cost up to 2x, e.g. the trailing
// products of the solves at 2500 vectors.)
template <int MODE>
__device__ __forceinline__ unsigned x3_lane_offset(const TileSrc& t, int x0) {
  const int tid = threadIdx.x, last = t.X - 1 - x0;
  if (MODE == X3_XROW) return 4u * ((unsigned)(tid >> 7) * 16u * (unsigned)t.cs + (unsigned)min(tid & 127, last));
  if (MODE == X3_KVEC) return 4u * ((unsigned)min(tid >> 1, last) * (unsigned)t.rs + (unsigned)(tid & 1) * 16u);
  return 0u;
}

template <int MODE>
__device__ __forceinline__ void g2r_x3(const TileSrc& t, int x0, int k0, int khi, unsigned voff, float (&r)[16]) {
  const int tid = threadIdx.x;
  if (MODE == X3_XROW) {
    const char* sb = reinterpret_cast<const char*>(t.P + (long)k0 * t.cs + x0);
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = *reinterpret_cast<const float*>(sb + (long)u * t.cs * 4 + voff);
  } else if (MODE == X3_KVEC) {
    const char* sb = reinterpret_cast<const char*>(t.P + (long)x0 * t.rs + k0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sb + 16 * q + voff);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[4 * q + e] = v[e];
    }
  } else {
    const long xo = (long)min(x0 + (tid >> 1), t.X - 1) * t.rs;
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = t.P[xo + (long)min(k0 + (tid & 1) * 16 + u, khi - 1) * t.cs];
  }
}

// (plane kernels, p3_pass: 1 = no DMA after the first tiles, 8 = every lane reads slot 0 of LDS after the first step: no bank
// conflicts, one broadcast line; 16 = no barriers)
// X3_DBG: what-if switches of tools/micro/x3_gemm_bench.hip (wrong results; 0 in the library): 1 = no global loads in
// the K loop, 2 = no operand split (raw bits), 4 = no LDS commit in the K loop.
#ifndef X3_DBG
#define X3_DBG 0
#endif

template <int MODE>
__device__ __forceinline__ void r2s_x3(const TileSrc& t, int x0, int k0, int khi, const float (&r)[16], float mul,
                                       u32x4_k (*P)[128 * 4]) {
  const int tid = threadIdx.x;
  const int xr = MODE == X3_XROW ? (tid & 127) : (tid >> 1);
  const int kb = MODE == X3_XROW ? (tid >> 7) * 16 : (tid & 1) * 16;
  unsigned pk[3][8];
#pragma unroll
  for (int u = 0; u < 16; u += 2) {
    float v0 = r[u] * mul, v1 = r[u + 1] * mul;
    const bool xin = x0 + xr < t.X;
    if (MODE == X3_EDGE) {
      v0 = (xin && k0 + kb + u < khi) ? v0 : 0.0f;
      v1 = (xin && k0 + kb + u + 1 < khi) ? v1 : 0.0f;
    } else {
      v0 = xin ? v0 : 0.0f;
      v1 = xin ? v1 : 0.0f;
    }
    unsigned q[3];
    if (X3_DBG & 2) { q[0] = __float_as_uint(v0); q[1] = __float_as_uint(v1); q[2] = q[0] ^ q[1]; }
    else split3_pair(v0, v1, q);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) pk[pl][u >> 1] = q[pl];
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = (kb >> 3) + h;
    const int sl = xr * 4 + (c ^ lds_swz(xr));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) P[pl][sl] = u32x4_k{pk[pl][4 * h], pk[pl][4 * h + 1], pk[pl][4 * h + 2], pk[pl][4 * h + 3]};
  }
}

// n K tiles [k0 + 32 t, +32) of one operand pair, accumulated into acc.  Enters and leaves with the LDS buffer free
// (every wave past its last fragment read).
template <int MA, int MB, bool LITE>
__device__ __forceinline__ void x3_pass(const TileSrc& ta, const TileSrc& tb, int m0, int n0, int k0, int khi, int n,
                                        float mul, GemmLdsX3& L, f32x4 (&acc)[4][4]) {
  constexpr int GK = kX3K, W = 64, NT = 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1, c = lane >> 4;
  float ra[16], rb[16];
  const unsigned va = x3_lane_offset<MA>(ta, m0), vb = x3_lane_offset<MB>(tb, n0);
  g2r_x3<MA>(ta, m0, k0, khi, va, ra);
  g2r_x3<MB>(tb, n0, k0, khi, vb, rb);
  r2s_x3<MA>(ta, m0, k0, khi, ra, mul, L.P[0]);
  r2s_x3<MB>(tb, n0, k0, khi, rb, 1.0f, L.P[1]);
  __syncthreads();
  for (int t = 0; t < n; ++t) {
    const int kn = k0 + min(t + 1, n - 1) * GK;        // the last iteration re-reads its own tile: no branch round the loads
    if (!(X3_DBG & 1)) {
      g2r_x3<MA>(ta, m0, kn, khi, va, ra);
      g2r_x3<MB>(tb, n0, kn, khi, vb, rb);
    }
    __builtin_amdgcn_sched_barrier(0);                 // the loads are issued here, ahead of the matrix work, not sunk into it
    bf16x8_k a[NT][3], b[NT][3];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = wm * W + i * 16 + (lane & 15);
      const int sl = row * 4 + (c ^ lds_swz(row));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[i][pl] = __builtin_bit_cast(bf16x8_k, L.P[0][pl][sl]);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * W + j * 16 + (lane & 15);
      const int sl = row * 4 + (c ^ lds_swz(row));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[j][pl] = __builtin_bit_cast(bf16x8_k, L.P[1][pl][sl]);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v = acc[i][j];
        if (!LITE) {
          v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][1], v, 0, 0, 0);   // m m'
          v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][2], v, 0, 0, 0);   // h l'
          v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][2], b[j][0], v, 0, 0, 0);   // l h'
        }
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][1], v, 0, 0, 0);   // h m'
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][0], v, 0, 0, 0);   // m h'
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][0], v, 0, 0, 0);   // h h'
        acc[i][j] = v;
      }
    __builtin_amdgcn_sched_barrier(0);                 // ... and nothing that uses them is hoisted into it
    __syncthreads();                      // every wave is done reading this K tile
    if (t + 1 < n && !(X3_DBG & 4)) {
      r2s_x3<MA>(ta, m0, kn, khi, ra, mul, L.P[0]);
      r2s_x3<MB>(tb, n0, kn, khi, rb, 1.0f, L.P[1]);
    }
    __syncthreads();
  }
}

// The fetch modes (MA, MB) are chosen by the HOST from the operand strides (x3_host_mode below; both operand pairs of a
// dual product must agree) and are template parameters of the kernel, so that a kernel holds two K loops only: the
// fast one (blocks at the x edge included: clamped rows, masked at the commit) and the scalar one for a ragged K tail.
template <int MA, int MB, bool LITE>
__device__ __forceinline__ void gemm_body_x3(const GemmArgs& g, int m0, int n0, GemmLdsX3& L) {
  constexpr int T = 128, GK = kX3K, NT = 4;
  const bool tri_skip = (g.epi == EPI_TRIU_MAX || g.sym) && (m0 >= n0 + T);
  if (g.sym && tri_skip) return;        // written by the mirror tile's epilogue (a triu tile below the diagonal stores zeros instead)

  f32x4 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float a_mul = 1.0f;
  if (g.scale_max) a_mul = g.step / (g.scale_max[0] + g.tiny);
  const bool fast = MA != X3_EDGE && MB != X3_EDGE;

  // one operand pair at a time: the second pair of a dual product restarts the pipeline (one more latency per block)
#pragma unroll 1
  for (int p = 0; p < ((g.A2 && !tri_skip) ? 2 : (tri_skip ? 0 : 1)); ++p) {
    const TileSrc ta = p ? TileSrc{g.A2, g.a2_rs, g.a2_cs, g.M} : TileSrc{g.A, g.a_rs, g.a_cs, g.M};
    const TileSrc tb = p ? TileSrc{g.B2, g.b2_cs, g.b2_rs, g.N} : TileSrc{g.B, g.b_cs, g.b_rs, g.N};   // (n, k) view of B
    const int K = p ? g.K2 : g.K, km = p ? g.kmode2 : g.kmode;
    int lo = 0, hi = K;
    if (km & KLO_M) lo = max(lo, m0);
    if (km & KLO_N) lo = max(lo, n0);
    if (km & KHI_M) hi = min(hi, m0 + T);
    if (km & KHI_N) hi = min(hi, n0 + T);
    lo = (lo / GK) * GK;
    if (hi <= lo) continue;
    const int ntot = (hi - lo + GK - 1) / GK, nfull = (hi - lo) / GK;
    const int e0 = fast ? nfull : 0;                    // first K tile of the clamped loop (ragged tail / edge blocks)
    const float mul = p ? -a_mul : a_mul;
    if (MA != X3_EDGE && MB != X3_EDGE && e0 > 0) x3_pass<MA, MB, LITE>(ta, tb, m0, n0, lo, hi, e0, mul, L, acc);
    if (ntot > e0) x3_pass<X3_EDGE, X3_EDGE, LITE>(ta, tb, m0, n0, lo + e0 * GK, hi, ntot - e0, mul, L, acc);
  }
  gemm_epilogue<T>(g, acc, m0, n0);
}

template <int MA, int MB, bool LITE>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_x3(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) GemmLdsX3 L;
  int by, bx;
  gemm_tile_order(g.kmode, by, bx);
  gemm_body_x3<MA, MB, LITE>(g, by * 128, bx * 128, L);
}

// Two independent large products in one launch (the two gradient products of the update, psgd.py:175-176, then its
// two factor updates, :179): with 528 upper tiles per 4096^2 triu product and 512 resident blocks, a launch of its own
// ends with a nearly empty second wave of blocks; back to back in one grid the tail is paid once.
struct GemmPair { GemmArgs g[2]; int tiles0, tx0, tx1, tiles1; };

// Block -> (product, tile id) of a two-product grid.  The products are INTERLEAVED in groups of 8 + 8 blocks (a group of 8
// keeps id % 8 = XCD): blocks are dispatched in id order, so with product 1 queued behind all of product 0 its long
// tiles found the block slots held by product 0's long tiles and ran as a second wave -- the two factor updates
// (work = distance from the diagonal) took 0.49 ms in one grid against 0.26 ms for one of them alone.
__device__ __forceinline__ void pair_block(int b, int tiles0, int tiles1, int& which, int& id) {
  const int nint = min(tiles0, tiles1) & ~7;
  if (b < 2 * nint) {
    const int t = b & 15;
    which = t >> 3;
    id = (b >> 4) * 8 + (t & 7);
  } else {
    const int r = b - 2 * nint, rem0 = tiles0 - nint;
    which = r >= rem0 ? 1 : 0;
    id = nint + (which ? r - rem0 : r);
  }
}

template <int MA0, int MB0, int MA1, int MB1>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_x3_pair(GemmPair p) {
  __shared__ __attribute__((aligned(16))) GemmLdsX3 L;
  int which, id;
  pair_block(blockIdx.x, p.tiles0, p.tiles1, which, id);
  const GemmArgs& g = p.g[which];
  const int tx = which ? p.tx1 : p.tx0;
  const int ty = ((g.M + 127) / 128);
  int by, bx;
  gemm_tile_from_id(id, ty, tx, g.kmode, by, bx);       // tiles0 is a multiple of 8 whenever the patch map applies
  if (which) gemm_body_x3<MA1, MB1, false>(g, by * 128, bx * 128, L);
  else gemm_body_x3<MA0, MB0, false>(g, by * 128, bx * 128, L);
}

// ---------------------------------------------------------------------------------------------
// The same fp32-accurate product on PRE-SPLIT operands ("planes").  k_gemm_x3 splits every operand element once per
// output tile column/row it meets -- 32 times at 4096^2 -- and stages it through registers; measured (tools/micro/
// x3_gemm_bench.hip) the split + LDS commit and the register-staged loads cost 35-45 % on top of the MFMA + fragment-
// read loop.  Here an operand is split ONCE into three bf16 planes in HBM (x = h + m + l exactly, 6 B/element), K-
// contiguous and zero-padded to whole tiles, and the K loop is DMA (global_load ... lds, no VGPR staging, no VALU) +
// fragment reads + MFMAs.  Producers write planes directly: k_split3 for caller data and factors, and the epilogue of
// this kernel for chained products (row-major and/or transposed planes), so an intermediate is never re-read as fp32.
// Plane layout: K-tile-major.  Element (x, k) of plane pl is p[pl * ps + (k / 32) * ts + x * 32 + k % 32] with
// ts = 32 * (padded x extent): the 128 rows x 32 k a block fetches per K tile and plane are ONE contiguous 8 KiB run, so
// every DMA instruction moves whole 128-byte lines (with row-major planes a K tile touched half of each line and every
// line was fetched twice through the 64 B/clk L1 -- the whole K loop was bound by it).  x padded to 128, k to 32, zeros.
struct PlaneMeta { float scale, inv, amax, bound; };      // f16 x 2 format (below): 2^e, 2^-e, max|X|, the bound e was chosen from
struct P3 {
  const __bf16* p;
  long ts, ps;
  const PlaneMeta* meta;   // f16 x 2 format only: of the matrix these planes hold (device memory)
  const int* te;           // f16 x 2, TILE scales (round 5; see "tile scales" below): te[xb * kTeLd + kb] = e with scale 2^e of the
                           // 128 x 128 tile (x block xb, k block kb) of this view, kTeAny for an all-zero tile; nullptr = one scale (meta)
};
constexpr int kTeLd = 64;                 // tiles per table row: extents up to 8192
constexpr int kTeAny = -(1 << 30);        // an all-zero tile: any scale
constexpr int kTeUnset = -(1 << 30) + 1;  // (p3_pass: no tile seen yet)
#ifndef PSGD_TE_QUANT
#define PSGD_TE_QUANT 8
#endif
constexpr int kTeQuant = PSGD_TE_QUANT;   // tile exponents are multiples of this

// ---- the second plane format (round 3): x 2^e = h + 2^-11 M with h, M in fp16 ("f16 x 2") -----------------------------
// Two planes and THREE matrix-core products per term instead of three planes and six: h h' into the accumulators, M h' and
// h M' into a second set that is folded in with 2^-11 after the K loop (206 registers, still two blocks per CU).  h =
// fp16(x s) keeps 11 bits, the residual another 11 (x s = h + m to 2^-23; the dropped m m' is 2^-24), so a product is at
// least as accurate as with the bf16 x 3 split, which drops three cross terms of 2^-24 (CPU emulation before it was
// built: profiles/r03_f16x2_planes_study.txt; on the device, whole apply against fp64: 5.7e-7 against 2.6e-6 at 4096^2,
// profiles/r03_f16x2_planes_ab.txt) at half the matrix-core work and two thirds of the LDS and DMA traffic (4096^2 apply
// 1.22 -> 0.81 ms).  What fp16 lacks is range, so every matrix carries ONE power-of-two scale s = 2^e that puts a bound of
// its max|x| at 2^14 (h never overflows), and the residual plane is stored pre-scaled (M = 2^11 m) so that it stays a
// normal number wherever h is one -- without that, elements more than ~2^11 below the bound lose residual bits to fp16's
// subnormal range.  Scales come from ACTUAL maxima: a matrix split from fp32 data (caller data, factors, solve results) is
// preceded by a reduction (k_absmax, or the balance launch for the balanced factors: per-block partial maxima that the split
// kernel reduces -- thousands of atomics on one address serialise in L2); a product that feeds the next product writes
// fp32 and accumulates max|C| in its epilogue, and a split launch makes its planes (p3_chain).  The first version let the
// epilogue write the planes itself with a scale from the bound K max|A| max|B| (still there: tuning key 16 = 0): bounds
// compounded over a chain cost 1e-4 .. 5e-4, and even a single bound is loose by the conditioning of the factors -- products
// like QlS (dG QrS') cancel by orders of magnitude -- which cost the update's increments two decimal digits on factors with
// cond 1e4 (tools/illcond_increment_probe.py, profiles/r03_f16x2_illcond_probe.txt).  The accumulators are brought back to
// real values (x 2^-(eA + eB), exact) before any epilogue logic.  NaN / Inf: a maximum that is not finite gives s = 1 (NaN)
// or the smallest scale (Inf) and the values themselves carry the NaN / Inf through the products.
typedef _Float16 f16x8_k __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float plane_scale_of_bound(float b) {      // 2^e with b 2^e in [2^13, 2^14)
  if (!(b > 0.0f)) return 1.0f;                                       // zero matrix, NaN: any scale (the values decide)
  if (!(b < 1.7e38f)) return 1.1754944e-38f * 32768.0f;               // a bound at the top of the range or beyond: 2^-111
  int k;
  (void)frexpf(b, &k);                                                // b = f 2^k, f in [0.5, 1)
  return ldexpf(1.0f, min(14 - k, 126));                              // (subnormal-sized data: 2^e and 2^-e stay normal)
}
// *addr = max(*addr, v) for non-negative v (NaN above Inf, see amaxf).  Thousands of waves report to ONE address and atomics
// on one address serialise in L2 (measured: 8192 of them turned the 16 us balance launch into 98 us), so a wave first looks:
// the value only grows, a stale look costs a redundant atomic at worst.
__device__ __forceinline__ void atomic_amax(float* addr, float v) {
  const unsigned u = __float_as_uint(v);
  if (u == 0u || u <= *reinterpret_cast<volatile unsigned*>(addr)) return;
  atomicMax(reinterpret_cast<int*>(addr), (int)u);
}
// two fp32 values (already scaled) -> packed fp16 pairs of the planes h and M = 2^11 (x - h)
__device__ __forceinline__ void split2h_pair(float x0, float x1, unsigned (&q)[2]) {
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
  const _Float16 m0 = (_Float16)((x0 - (float)h0) * 2048.0f), m1 = (_Float16)((x1 - (float)h1) * 2048.0f);
  q[0] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  q[1] = (unsigned)__builtin_bit_cast(unsigned short, m0) | ((unsigned)__builtin_bit_cast(unsigned short, m1) << 16);
}
__host__ __device__ __forceinline__ long p3_index(long ts, long x, long k) { return (k >> 5) * ts + x * 32 + (k & 31); }

struct P3Args {
  P3 A, B;             // (m, k) and (n, k) views
  P3 A2, B2;           // optional second pair, subtracted (K2, kmode2 in e)
  GemmArgs e;          // M, N, K, kmode and every epilogue field of the fp32 kernels (e.C may be null; e.A .. e.B2 unused,
                       // except e.A2 != nullptr <=> the second pair is present)
  __bf16* Crow; long crow_ts, crow_ps;   // planes of C  (x = row, k = column)  (optional)
  __bf16* Ccol; long ccol_ts, ccol_ps;   // planes of C' (x = column, k = row)  (optional)
  int fmt;                               // 0: bf16 x 3 planes, 1: f16 x 2 planes (operands and plane outputs alike)
  // f16 x 2 plane outputs: their scale comes from bound = okmul * oa->amax * ob->amax (+ okmul2 * oa2->amax * ob2->amax);
  // ometa receives {scale, 1 / scale, max|C| (atomic; zeroed by the host beforehand), bound}
  PlaneMeta* ometa;
  const PlaneMeta *oa, *ob, *oa2, *ob2;
  float okmul, okmul2;
  // f16 x 2 plane outputs with TILE scales: every 128 x 128 tile of C at the scale of its own maximum, written by the epilogue that
  // holds it -- no fp32 round trip, no max|C| over the grid, no split launch.  te_row[(m0 / 128) * kTeLd + n0 / 128] / te_col[(n0 / 128) *
  // kTeLd + m0 / 128] receive the tile's exponent (the tables of Crow / Ccol).  Null: the output's scale comes from ometa as before.
  int *te_row, *te_col;
  int neg;               // (tile-scale outputs only) the result is -(A B ...): the inverse levels' -T = -(A^-1 B)
  int lower_zero;        // C tiles strictly below the diagonal are written as zeros, D is not read there (the factor updates when the
                         // fused prologue left the balanced factors' lower tiles unwritten: k_kron_balance_planes)
};

typedef __attribute__((address_space(3))) void* lds_ptr3_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr3_t;
#define P3_FENCE() asm volatile("" ::: "memory")
#ifndef P3_EARLY
#define P3_EARLY 2      // MFMA columns (of 4) issued before the buffer-free barrier; 0 = all fragments first, then all MFMAs
#endif
// The f16 x 2 kernels exist in both forms (template parameter ER, chosen by the launcher): grids of at least a full round of
// block slots run with none -- with half the MFMAs per step the next tile's DMA wants to go out as early as it can (tools/micro:
// 4096^3 0.305-0.324 -> 0.282 ms, triangular K ranges 0.166 -> 0.157, the gradient grid at 6144^2 2.53 -> 2.33, at 2048^2
// 0.151 -> 0.128) -- smaller ones keep two (a lone workgroup per CU loses 3-5 % without: 1024^3 0.036 -> 0.037, 1000^2 apply
// 0.124 -> 0.131).  A run-time switch inside the loop instead of two instantiations cost 35 % (4096^2 apply 0.78 -> 1.11 ms).
// Per kernel and size (tuning key 19 forces a form; micro bench, ms, none / two): plain product 2048^3 0.065 / 0.063, 2560^3 0.087 /
// 0.098, 4096^3 0.285 / 0.310; gradient grid 1536^2 0.095 / 0.093, 2048^2 0.128 / 0.151, 2560^2 0.173 / 0.193, 4096^2 0.760 / 0.761;
// factor-update pair 2048^2 0.063 / 0.063, 2560^2 0.089 / 0.098, 4096^2 0.188 / 0.208 -- hence "none" from 3/4 of the block slots on
// (products, pair) and from half of them on (gradient grid).
constexpr int p3_early(int FMT, int ER) { return FMT ? ER : P3_EARLY; }

// LDS of the plane kernels, ONE object (a second __shared__ object makes hipcc guard its accesses with vmcnt(0) while DMA
// into the first is in flight).  bf16 x 3: one 48 KiB stage, the layout of GemmLdsX3.  f16 x 2: TWO 32 KiB stages -- with
// half the MFMAs per K step a step no longer covers the latency of the next step's DMA, so tiles are requested two steps
// ahead and awaited with a counted vmcnt (64 KiB per block, still two blocks per CU).
template <int FMT>
struct P3Lds {
  u32x4_k P[FMT ? 2 : 1][2][FMT ? 2 : 3][128 * 4];    // [stage][A|B][plane][row * 4 + (chunk ^ lds_swz(row))]
  unsigned ticket[4];                                  // (split-K: the arrival number of this block)
  float red[4];                                        // (tile scales: the waves' maxima of the finished tile)
};

// ---- tile scales (round 5) ------------------------------------------------------------------------------------------------------
// A matrix-wide scale needs max|C| over the whole grid before the first plane element can be written: every chained product was
// "product (fp32 out + atomic max) -> split launch", and those launches were a quarter of the large update's critical path (657 of
// 2616 us at 4096^2, two of them 140 us each beside a full-chip product of the other stream; profiles/r04_kron_update_trace_f32.txt).
// A scale only has to be constant along the K range ONE accumulation runs over -- and fp32 accumulators can be moved from one
// power-of-two scale to another exactly (v_ldexp_f32).  So a producer tile writes its planes at the scale of ITS OWN maximum, straight
// from the accumulators, and leaves the exponent in a table; a consumer's K loop looks up the exponents of its A and B tiles at every
// 128-k boundary (lane-held: one v_readlane each) and, when their sum differs from the scale its accumulators are at, shifts the
// accumulators (128 v_ldexp_f32, only when the scale changes).  Tighter scales than one per matrix, no extra launch, no fp32 round trip.
// Accumulators are at the scale of the tile in hand: they overflow only if a row of tiles spans > 2^80 in magnitude.
__device__ __forceinline__ int p3_exp_of_scale(float s) { return (int)((__float_as_uint(s) >> 23) & 255u) - 127; }
template <int FMT>
__device__ __forceinline__ int p3_tile_exps(const P3& X, int x0, int lo, int hi) {      // lane l: exponent of K tile (lo / 128 + l)
  if constexpr (FMT != 1) return 0;
  const int lane = threadIdx.x & 63;
  if (!X.te) return p3_exp_of_scale(X.meta->scale);
  const int kb0 = lo >> 7, nkb = ((hi + 127) >> 7) - kb0;
  return X.te[(x0 >> 7) * kTeLd + kb0 + min(lane, nkb - 1)];
}

// K tiles [lo, hi) of one operand pair.  LDS image as in k_gemm_x3: [A|B][plane][row * 4 + (chunk ^ swz(row))].  The DMA
// writes linearly (wave base + 16 B * lane), so the swizzle is applied to the per-lane SOURCE address: the lane that
// fills slot s = row * 4 + cpos fetches chunk cpos ^ swz(row) of that row.
// (tile scales) cur_exp: the unit the accumulators of an output tile are in, carried from pass to pass -- the exponent sum of the K
// tile in hand
template <int FMT, int ER, bool TS = false>
__device__ __forceinline__ void p3_pass(const P3& A, const P3& B, int m0, int n0, int lo, int hi, P3Lds<FMT>& L,
                                        f32x4 (&acc)[4][4], int* cur_exp = nullptr) {
  constexpr int EARLY = p3_early(FMT, ER);
  constexpr int NP = FMT ? 2 : 3;
  constexpr int W = 64, NT = 4;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1, c = lane >> 4;
  unsigned offA[2], offB[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int s = (2 * w + q) * 64 + lane, row = s >> 2, chunk = (s & 3) ^ lds_swz(row);
    offA[q] = offB[q] = (unsigned)(row * 32 + chunk * 8);
  }
  const __bf16* baseA = A.p + (long)m0 * 32;
  const __bf16* baseB = B.p + (long)n0 * 32;
  constexpr int NS = FMT ? 2 : 1;           // stages
  auto issue = [&](int k0, int st) {
    const long ka = (long)(k0 >> 5) * A.ts, kb = (long)(k0 >> 5) * B.ts;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        __builtin_amdgcn_global_load_lds((gbl_ptr3_t)(baseA + pl * A.ps + ka + offA[q]), (lds_ptr3_t)&L.P[st][0][pl][(2 * w + q) * 64], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_ptr3_t)(baseB + pl * B.ps + kb + offB[q]), (lds_ptr3_t)&L.P[st][1][pl][(2 * w + q) * 64], 16, 0, 0);
      }
  };
  if (hi <= lo) return;
  f32x4 cross[FMT == 1 ? NT : 1][FMT == 1 ? NT : 1];
  if constexpr (FMT == 1) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) cross[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  // tile scales: the exponents of this row of A tiles / B tiles, one K tile per lane (requested ahead of the DMA: older in vmcnt order)
  int vexpA = 0, vexpB = 0, cur = kTeUnset;
  bool live = true;                         // (TS) false while a 128-k tile is being dropped (overflow guard below)
  if constexpr (TS) {
    vexpA = p3_tile_exps<FMT>(A, m0, lo, hi);
    vexpB = p3_tile_exps<FMT>(B, n0, lo, hi);
    cur = *cur_exp;
  }
  issue(lo, 0);
  if (NS == 2 && lo + kX3K < hi) issue(lo + kX3K, 1);
  int st = 0;
  for (int k0 = lo; k0 < hi; k0 += kX3K, st ^= (NS - 1)) {
    // this step's tile has landed: everything but the (2 x 2 planes x 2 halves =) 8 DMA instructions of the next step's
    if (NS == 2 && k0 + kX3K < hi) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P3_FENCE();
    if (!(X3_DBG & 16)) __builtin_amdgcn_s_barrier();           // every wave's part of this K tile has landed
    P3_FENCE();
    if constexpr (TS && FMT == 1) {
      if ((k0 & 127) == 0 || k0 == lo) {                        // a new 128-k tile: are the accumulators at its scale?
        const int kb = __builtin_amdgcn_readfirstlane((k0 >> 7) - (lo >> 7));
        const int ea = __builtin_amdgcn_readlane(vexpA, kb), eb = __builtin_amdgcn_readlane(vexpB, kb);
        live = true;
        if (ea != kTeAny && eb != kTeAny) {                     // (an all-zero tile adds zeros at any scale)
          const int c = ea + eb;
          const int d = (cur == kTeUnset) ? 0 : c - cur;
          bool drop = false;
          if (d > 0) {
            // Moving UP to the unit of a finer tile multiplies what is already there by 2^d: look at what IS there (this wave's
            // largest |acc|, |cross|; upward moves are rare and the matrix cores are busy meanwhile).  A tile that would push it
            // past 2^120 is dropped instead: its whole contribution is below 2^-78 of the sum so far, and the matrix-wide
            // scale of round 4 flushed such a tile to zero before it was ever multiplied.  Per wave: every wave
            // carries its own accumulators and its own unit.
            unsigned mx = 0u;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  mx = max(mx, __float_as_uint(acc[i][j][e]) & 0x7fffffffu);
                  mx = max(mx, __float_as_uint(cross[i][j][e]) & 0x7fffffffu);
                }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
            mx = __builtin_amdgcn_readfirstlane(mx);
            drop = mx != 0u && (int)(mx >> 23) - 127 + d > 120;
          }
          if (drop) {
            live = false;                                       // (the products of this 128-k tile are not issued)
          } else {
            if (d != 0) {
#pragma unroll
              for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    acc[i][j][e] = ldexpf(acc[i][j][e], d);
                    cross[i][j][e] = ldexpf(cross[i][j][e], d);
                  }
            }
            cur = c;
          }
        }
      }
    }
    bf16x8_k a[NT][NP], b[NT][NP];
    const bool skip_reads = (X3_DBG & 8) && k0 != lo;           // (what-if: fragments of the first step reused)
    auto read_b = [&](int j) {
      const int row = wn * W + j * 16 + (lane & 15);
      const int sl = row * 4 + (c ^ lds_swz(row));
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) b[j][pl] = __builtin_bit_cast(bf16x8_k, L.P[st][1][pl][skip_reads ? 0 : sl]);
    };
    auto mfma6 = [&](int i, int j) {
      f32x4 v = acc[i][j];
      if constexpr (FMT == 1) {               // f16 x 2: h h' into acc, M h' + h M' into cross (folded in with 2^-11 at the end)
        const f16x8_k ah = __builtin_bit_cast(f16x8_k, a[i][0]), aM = __builtin_bit_cast(f16x8_k, a[i][1]);
        const f16x8_k bh = __builtin_bit_cast(f16x8_k, b[j][0]), bM = __builtin_bit_cast(f16x8_k, b[j][1]);
        f32x4 x = cross[i][j];
        x = __builtin_amdgcn_mfma_f32_16x16x32_f16(aM, bh, x, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, v, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bM, x, 0, 0, 0);
        cross[i][j] = x;
        acc[i][j] = v;
      } else {
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][1], v, 0, 0, 0);   // m m'
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][NP - 1], v, 0, 0, 0);   // h l'
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][NP - 1], b[j][0], v, 0, 0, 0);   // l h'
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][1], v, 0, 0, 0);   // h m'
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][1], b[j][0], v, 0, 0, 0);   // m h'
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][0], b[j][0], v, 0, 0, 0);   // h h'
        acc[i][j] = v;
      }
    };
    // The first column of MFMAs starts as soon as ITS fragments are there (the compiler's counted lgkmcnt waits); the rest
    // of the fragment reads return under it.  Only then: every wave holds its fragments, the buffer is free.
    read_b(0);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int row = wm * W + i * 16 + (lane & 15);
      const int sl = row * 4 + (c ^ lds_swz(row));
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) a[i][pl] = __builtin_bit_cast(bf16x8_k, L.P[st][0][pl][skip_reads ? 0 : sl]);
    }
#pragma unroll
    for (int j = 1; j < NT; ++j) read_b(j);
    if constexpr (EARLY > 0) {
      // (TS: a dropped tile skips its products with a wave-uniform branch around each of the two groups -- in the common path one
      // scalar branch, and the explicit lgkmcnt(0) below is the join, so the counted waits of the fragment reads stay as they were)
      if (!TS || live) {
#pragma unroll
        for (int j = 0; j < EARLY; ++j)
#pragma unroll
          for (int i = 0; i < NT; ++i) mfma6(i, j);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // (measured and not kept: ONE barrier per step with the next tile requested right behind it into the other stage --
    // 4096^3 0.317 -> 0.302 ms, but the triangular K ranges 0.173 -> 0.194)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    P3_FENCE();
    if (!(X3_DBG & 16)) __builtin_amdgcn_s_barrier();           // every wave holds its fragments: the buffer is free
    P3_FENCE();
    if (k0 + NS * kX3K < hi && !(X3_DBG & 1)) issue(k0 + NS * kX3K, st);   // the next tile for this stage streams in under the MFMAs
    if (!TS || live) {
#pragma unroll
      for (int j = EARLY; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) mfma6(i, j);
    }
  }
  if constexpr (FMT == 1) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] += cross[i][j] * 0.00048828125f;
  }
  if constexpr (TS) *cur_exp = cur;
}
// plane outputs of a C tile (pads inside the padded extents are written as zeros)
template <int FMT>
__device__ __forceinline__ void p3_store_planes(const P3Args& g, const f32x4 (&acc)[4][4], int m0, int n0, bool do_row,
                                                bool do_col, float tile_sc = 0.0f) {      // tile_sc != 0: this tile's own scale (tile scales)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;
  const bool odd = lane & 1;
  if constexpr (FMT == 1) {
    // the output's scale from the bound its host described (the same in every block), its actual maximum for its consumers
    float sc = tile_sc;
    if (tile_sc == 0.0f) {
      float bound = g.okmul * g.oa->amax * g.ob->amax;
      if (g.oa2) bound += g.okmul2 * g.oa2->amax * g.ob2->amax;
      sc = plane_scale_of_bound(bound);
      if (threadIdx.x == 0) {            // (every block that gets here: under a K split any block may be a tile's last one)
        g.ometa->scale = sc; g.ometa->inv = 1.0f / sc; g.ometa->bound = bound;
      }
    }
    float vmax = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
        const int col = n0 + wn * 64 + j * 16 + (lane & 15);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (row0 + e < g.e.M && col < g.e.N) ? acc[i][j][e] : 0.0f;
          vmax = amaxf(vmax, fabsf(v[e]));
          v[e] *= sc;
        }
        if (do_col) {
          unsigned q0[2], q1[2];
          split2h_pair(v[0], v[1], q0);
          split2h_pair(v[2], v[3], q1);
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
            *reinterpret_cast<uint2*>(g.Ccol + pl * g.ccol_ps + p3_index(g.ccol_ts, col, row0)) = make_uint2(q0[pl], q1[pl]);
        }
        if (do_row) {
          const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];
          const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
          const float x00 = odd ? r0 : v[0], x01 = odd ? v[2] : r0;
          const float x10 = odd ? r1 : v[1], x11 = odd ? v[3] : r1;
          unsigned q0[2], q1[2];
          split2h_pair(x00, x01, q0);
          split2h_pair(x10, x11, q1);
          const int rr = row0 + (odd ? 2 : 0), cc = col - (odd ? 1 : 0);
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            *reinterpret_cast<unsigned*>(g.Crow + pl * g.crow_ps + p3_index(g.crow_ts, rr, cc)) = q0[pl];
            *reinterpret_cast<unsigned*>(g.Crow + pl * g.crow_ps + p3_index(g.crow_ts, rr + 1, cc)) = q1[pl];
          }
        }
      }
    if (tile_sc != 0.0f) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
    if (lane == 0) atomic_amax(&g.ometa->amax, vmax);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
      const int col = n0 + wn * 64 + j * 16 + (lane & 15);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (row0 + e < g.e.M && col < g.e.N) ? acc[i][j][e] : 0.0f;
      if (do_col) {                                       // C'[col][row0 .. row0 + 3]: 4 consecutive bf16 per plane
        unsigned q0[3], q1[3];
        split3_pair(v[0], v[1], q0);
        split3_pair(v[2], v[3], q1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          *reinterpret_cast<uint2*>(g.Ccol + pl * g.ccol_ps + p3_index(g.ccol_ts, col, row0)) = make_uint2(q0[pl], q1[pl]);
      }
      if (do_row) {                                       // C[row][col]: lane pairs exchange so that a lane stores (col, col + 1)
        const float s0 = odd ? v[0] : v[2], s1 = odd ? v[1] : v[3];
        const float r0 = __shfl_xor(s0, 1, 64), r1 = __shfl_xor(s1, 1, 64);
        // even lane: rows row0, row0 + 1 at columns (col, col + 1);  odd lane: rows row0 + 2, row0 + 3 at (col - 1, col)
        const float x00 = odd ? r0 : v[0], x01 = odd ? v[2] : r0;
        const float x10 = odd ? r1 : v[1], x11 = odd ? v[3] : r1;
        unsigned q0[3], q1[3];
        split3_pair(x00, x01, q0);
        split3_pair(x10, x11, q1);
        const int rr = row0 + (odd ? 2 : 0), cc = col - (odd ? 1 : 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          *reinterpret_cast<unsigned*>(g.Crow + pl * g.crow_ps + p3_index(g.crow_ts, rr, cc)) = q0[pl];
          *reinterpret_cast<unsigned*>(g.Crow + pl * g.crow_ps + p3_index(g.crow_ts, rr + 1, cc)) = q1[pl];
        }
      }
    }
}

// 4 waves, one 48 KiB stage, two blocks per CU.  Measured and not kept: a 256 x 128 tile with 8 waves, one block per CU and
// two stages (DMA a whole K step ahead, one barrier per step, 25 % less L1 traffic): 0.519 against 0.532 ms at 4096^3 and
// far worse on the triangular K ranges (512 uneven tiles on 256 CUs); three blocks per CU with the A fragments read in
// two halves (164 registers): 0.514-0.537.  Under this kernel the device sits at 1.98-2.13 GHz and 1260-1335 W of its
// 1400 W (tools/clock_probe.sh): 1540 TFLOP/s issued is 72 % of the matrix peak AT THAT CLOCK, the K loop without its DMA
// (the what-if floor, 0.42 ms) 91 % -- what is left is mostly not schedule.
// K split of a tile (chunk >= 0): the tile's K steps are dealt to `nchunk` blocks (the subtracted pair's to the first
// half); every block leaves its partial tile in `scratch`, and the LAST one to arrive (ticket counter) sums the partials
// in chunk order -- a fixed order, whoever arrives last -- and runs the epilogue.  The counter is left at zero again.
struct P3Split { int chunk, nchunk; float* scratch; unsigned* cnt; };     // scratch, cnt: of THIS tile

__device__ __forceinline__ void pow2_halves(double v, float& m1, float& m2) {   // v = 2^e (> 0)  ->  m1 m2 = v, exponents e/2 each
  int e;
  (void)frexp(v, &e);                                   // v = 0.5 * 2^e
  e -= 1;
  const int h = e / 2;
  m1 = ldexpf(1.0f, h); m2 = ldexpf(1.0f, e - h);
}

template <int FMT = 0, int ER = 2, bool TS = false>
__device__ __forceinline__ void p3_body(const P3Args& g, int by, int bx, P3Lds<FMT>& L, const P3Split sp = P3Split{-1, 0, nullptr, nullptr}) {
  constexpr int TM = 128, TN = 128, GK = kX3K;
  static_assert(!TS || FMT == 1, "tile scales are a property of the f16 x 2 format");
  int cur = kTeUnset;                   // (TS) the power-of-two scale the accumulators are at
  // f16 x 2: what brings a pair's accumulators back to real values (2^-(eA + eB), exact)
  // (2^-(eA + eB) can leave the fp32 range where the real result does not: kept in double, applied as two balanced
  // power-of-two factors -- the intermediate lies between the accumulator and the result, so it is in range when they are)
  float ia1 = 1.0f, ib1 = 1.0f, ia2 = 1.0f, ib2 = 1.0f;
  double inv1 = 1.0, inv2 = 1.0;
  if constexpr (FMT == 1 && !TS) {
    inv1 = (double)g.A.meta->inv * g.B.meta->inv;
    pow2_halves(inv1, ia1, ib1);
    if (g.e.A2) { inv2 = (double)g.A2.meta->inv * g.B2.meta->inv; pow2_halves(inv2, ia2, ib2); }
  }
  const int m0 = by * TM, n0 = bx * TN;
  const bool tri_skip = (g.e.epi == EPI_TRIU_MAX || g.e.sym) && (m0 >= n0 + TN);
  if (g.e.sym && tri_skip) return;      // written by the mirror tile's epilogue
  if (g.lower_zero && m0 >= n0 + TN) {
    if (g.e.C) {
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + w * 32 + i * 2 + (lane >> 5), col = n0 + (lane & 31) * 4;
        if (row < g.e.M) {
          if (col + 3 < g.e.N && !g.e.c_cs && (g.e.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.e.C) & 15) == 0)
            *reinterpret_cast<float4*>(g.e.C + (long)row * g.e.ldc + col) = make_float4(0.f, 0.f, 0.f, 0.f);
          else
            for (int e = 0; e < 4; ++e)
              if (col + e < g.e.N) g.e.C[(long)row * g.e.ldc + (long)(col + e) * (g.e.c_cs ? g.e.c_cs : 1)] = 0.0f;
        }
      }
    }
    return;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (sp.chunk >= 0) {
    const int npair = g.e.A2 ? 2 : 1, per_pair = sp.nchunk / npair;
    const int p = npair - 1 - sp.chunk / per_pair, sub = sp.chunk % per_pair;     // the subtracted pair first, as below
    const int K = p ? g.e.K2 : g.e.K, km = p ? g.e.kmode2 : g.e.kmode;
    int lo = 0, hi = K;
    if (km & KLO_M) lo = max(lo, m0);
    if (km & KLO_N) lo = max(lo, n0);
    if (km & KHI_M) hi = min(hi, m0 + TM);
    if (km & KHI_N) hi = min(hi, n0 + TN);
    if (km & KBLK_HI_M) hi = min(hi, (m0 / g.e.kblk) * g.e.kblk + g.e.kblk / 2);
    if (km & KBLK_LO_N) lo = max(lo, (n0 / g.e.kblk) * g.e.kblk + g.e.kblk / 2);
    lo = (lo / GK) * GK;
    hi = ((hi + GK - 1) / GK) * GK;
    const int steps = (hi - lo) / GK, per = (steps + per_pair - 1) / per_pair;
    const int clo = lo + sub * per * GK, chi = min(hi, clo + per * GK);
    p3_pass<FMT, ER, TS>(p ? g.A2 : g.A, p ? g.B2 : g.B, m0, n0, clo, chi, L, acc, &cur);
    f32x4* mine = reinterpret_cast<f32x4*>(sp.scratch) + (long)sp.chunk * (16 * kThreads);
    float pa = p ? -ia2 : ia1, pb = p ? ib2 : ib1;   // partials are stored as real values (f16 x 2: each pair has its own scale)
    if constexpr (TS) {                                // (tile scales: back from wherever the chunk's last tile left the accumulators)
      const int back = cur == kTeUnset ? 0 : -cur;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = ldexpf(acc[i][j][e], back);
      pa = p ? -1.0f : 1.0f; pb = 1.0f;
    }
    // Hand-off (the fused bf16 pair's recipe): the partial goes out write-through (8-byte agent-scope stores), every wave
    // drains its stores, the block's barrier, one lane takes the ticket; the last block to arrive does one agent-scope
    // acquire.  A __threadfence() here instead is a release of the WHOLE L2 (buffer_wbl2) per block: it made a split item
    // cost ~100 us more than its K steps (3072^2 gradient grid with 1408 items: 0.745 ms against 0.363 unsplit).
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = (acc[i][j] * pa) * pb;
        unsigned long long* q = reinterpret_cast<unsigned long long*>(mine + (i * 4 + j) * kThreads + threadIdx.x);
        __hip_atomic_store(q, (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) L.ticket[0] = __hip_atomic_fetch_add(sp.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (L.ticket[0] != (unsigned)(sp.nchunk - 1)) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other blocks' partials
    if (threadIdx.x == 0) __hip_atomic_store(sp.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4* q = reinterpret_cast<const f32x4*>(sp.scratch) + (i * 4 + j) * kThreads + threadIdx.x;
        f32x4 v = __builtin_nontemporal_load(q);
        for (int c = 1; c < sp.nchunk; ++c) v += __builtin_nontemporal_load(q + (long)c * (16 * kThreads));
        acc[i][j] = v;
      }
  } else if (!tri_skip) {
    // the subtracted pair first, then one sign flip of the accumulators.  (f16 x 2: the pairs have their own units,
    // 2^-(eA + eB); the pair with the FINER unit goes first and its sums are carried over to the coarser unit -- a factor
    // <= 1, so nothing overflows however far apart the two products are)
    auto k_range = [&](int p, int& lo, int& hi) {
      const int K = p ? g.e.K2 : g.e.K, km = p ? g.e.kmode2 : g.e.kmode;
      lo = 0; hi = K;
      if (km & KLO_M) lo = max(lo, m0);
      if (km & KLO_N) lo = max(lo, n0);
      if (km & KHI_M) hi = min(hi, m0 + TM);
      if (km & KHI_N) hi = min(hi, n0 + TN);
      if (km & KBLK_HI_M) hi = min(hi, (m0 / g.e.kblk) * g.e.kblk + g.e.kblk / 2);
      if (km & KBLK_LO_N) lo = max(lo, (n0 / g.e.kblk) * g.e.kblk + g.e.kblk / 2);
      lo = (lo / GK) * GK;
      hi = ((hi + GK - 1) / GK) * GK;                 // the planes are zero-padded to whole K tiles
    };
    bool swap = FMT == 1 && !TS && g.e.A2 && inv2 > inv1;
    if constexpr (TS) {
      // tile scales: the same rule per output tile -- the pair whose FINEST tile scale is finer goes first, so the accumulators are only
      // ever carried to coarser units between the pairs (dX ~ 1e12 against dG ~ 1e-13: the other order is a shift by 2^+166)
      if (g.e.A2) {
        int fin[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          int lo, hi;
          k_range(p, lo, hi);
          int c = kTeAny;
          if (hi > lo) {
            const int ea = p3_tile_exps<FMT>(p ? g.A2 : g.A, m0, lo, hi), eb = p3_tile_exps<FMT>(p ? g.B2 : g.B, n0, lo, hi);
            c = (ea == kTeAny || eb == kTeAny) ? kTeAny : ea + eb;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c = max(c, __shfl_xor(c, off, 64));
          }
          fin[p] = __builtin_amdgcn_readfirstlane(c);
        }
        swap = fin[1] < fin[0];
      }
    }
#pragma unroll 1
    for (int it = g.e.A2 ? 1 : 0; it >= 0; --it) {
      const int p = swap ? 1 - it : it;
      int lo, hi;
      k_range(p, lo, hi);
      p3_pass<FMT, ER, TS>(p ? g.A2 : g.A, p ? g.B2 : g.B, m0, n0, lo, hi, L, acc, &cur);
      if (it) {                                        // (f16 x 2: and over to the other pair's unit, a power of two)
        // (tile scales: only the sign -- the second pass moves the accumulators to its own tiles' scales like any change of tile)
        const float flip = (FMT == 1 && !TS) ? -(float)((p ? inv2 : inv1) / (p ? inv1 : inv2)) : -1.0f;     // (<= 1 in magnitude)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] *= flip;
      }
    }
    if constexpr (FMT == 1 && !TS) {                   // back to real values (the last pair's unit; A - B either way)
      const float fa = swap ? -ia2 : ia1, fb = swap ? ib2 : ib1;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc[i][j] * fa) * fb;
    }
    if constexpr (TS) {
      const int back = cur == kTeUnset ? 0 : -cur;
      const float sg = swap ? -1.0f : 1.0f;            // (A - B either way)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] = ldexpf(acc[i][j][e] * sg, back);
    }
  }
  if (g.e.scale_max) {                                // (step / max) A B = step / max (A B): applied to the finished sums
    const float mul = g.e.step / (g.e.scale_max[0] + g.e.tiny);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] *= mul;
  }
  if (g.e.epi == EPI_TRIU_MAX && !g.e.C) {            // triu and max|.| on the registers: the planes below are the only output
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float vmax = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = m0 + (w >> 1) * 64 + i * 16 + (lane >> 4) * 4 + e, col = n0 + (w & 1) * 64 + j * 16 + (lane & 15);
          const float v = (col >= row && row < g.e.M && col < g.e.N) ? acc[i][j][e] : 0.0f;
          acc[i][j][e] = v;
          vmax = amaxf(vmax, fabsf(v));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
    if (lane == 0) atomic_amax(g.e.maxout, vmax);
  }
  if constexpr (FMT == 1) {
    // an fp32 result that a split launch turns into f16 x 2 planes afterwards: its exact max|C| for that launch
    if (g.ometa && !g.Crow && !g.Ccol) {
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      float vmax = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = m0 + (w >> 1) * 64 + i * 16 + (lane >> 4) * 4 + e, col = n0 + (w & 1) * 64 + j * 16 + (lane & 15);
            vmax = amaxf(vmax, (row < g.e.M && col < g.e.N) ? fabsf(acc[i][j][e]) : 0.0f);
          }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
      if (lane == 0) atomic_amax(&g.ometa->amax, vmax);
    }
  }
  if constexpr (TS) {
    if (g.te_row || g.te_col) {
      // tile scales: this tile's planes at the scale of its own maximum, from the registers that hold it.  The fp32 store (when the
      // caller wants one) happens here too, because the planes are of the FINAL values: D - acc for a D-minus epilogue, col >= row
      // for a triu epilogue, -acc under g.neg.  (No column scales, no mirrored store: the launchers keep such products off this path.)
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      const long ccs = g.e.c_cs ? g.e.c_cs : 1;
      if (g.e.epi == EPI_D_MINUS) {                      // every D element requested before the first use (see gemm_epilogue)
        f32x4 dv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = min(m0 + (w >> 1) * 64 + i * 16 + (lane >> 4) * 4 + e, g.e.M - 1);
              const int col = min(n0 + (w & 1) * 64 + j * 16 + (lane & 15), g.e.N - 1);
              dv[i][j][e] = g.e.D[(long)row * g.e.ldd + col * ccs];
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = dv[i][j] - acc[i][j];
      }
      float vmax = 0.0f;
      const float sgn = g.neg ? -1.0f : 1.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = m0 + (w >> 1) * 64 + i * 16 + (lane >> 4) * 4 + e, col = n0 + (w & 1) * 64 + j * 16 + (lane & 15);
            const bool in = row < g.e.M && col < g.e.N;
            const float v = (in && (g.e.epi != EPI_TRIU_MAX || col >= row)) ? acc[i][j][e] * sgn : 0.0f;
            acc[i][j][e] = v;
            vmax = amaxf(vmax, fabsf(v));
            if (g.e.C && in) g.e.C[(long)row * g.e.ldc + col * ccs] = v;
          }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
      if (g.e.epi == EPI_TRIU_MAX && g.e.C && lane == 0) atomic_amax(g.e.maxout, vmax);     // (C == nullptr: done on the registers above)
      __syncthreads();
      if (lane == 0) L.red[w] = vmax;
      __syncthreads();
      vmax = amaxf(amaxf(L.red[0], L.red[1]), amaxf(L.red[2], L.red[3]));
      // Exponents in steps of kTeQuant (rounded towards the smaller scale: the maximum lands in [2^(14 - kTeQuant), 2^14) instead of
      // [2^13, 2^14), which costs kTeQuant - 1 of fp16's ~28 binades below the maximum).  Neighbouring tiles of ordinary data differ by
      // a binade or two: unquantised, a consumer shifted its accumulators at nearly every 128-k boundary (the gradient grid 715 -> 764 us).
      float sc = plane_scale_of_bound(vmax);
      {
        const int e0 = p3_exp_of_scale(sc), eq = e0 - (((e0 % kTeQuant) + kTeQuant) % kTeQuant);
        sc = ldexpf(1.0f, max(eq, -126));
      }
      const bool do_col = g.Ccol != nullptr && (!g.e.sym || n0 > m0);
      if (threadIdx.x == 0) {
        const int ex = (__float_as_uint(vmax) == 0u) ? kTeAny : p3_exp_of_scale(sc);
        if (g.te_row) g.te_row[(m0 >> 7) * kTeLd + (n0 >> 7)] = ex;
        if (g.te_col && do_col) g.te_col[(n0 >> 7) * kTeLd + (m0 >> 7)] = ex;
      }
      p3_store_planes<FMT>(g, acc, m0, n0, g.Crow != nullptr, do_col, sc);
      return;
    }
  }
  if (g.e.C) gemm_epilogue<128>(g.e, acc, m0, n0);
  // a symmetric product (Gram) names the same buffer twice: the mirror image is the transposed store of the tiles above
  // the diagonal (a diagonal tile holds both halves itself)
  if (g.Crow || g.Ccol) p3_store_planes<FMT>(g, acc, m0, n0, g.Crow != nullptr, g.Ccol != nullptr && (!g.e.sym || n0 > m0));
}

template <int FMT, int ER = 2, bool TS = false>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_p3(P3Args g) {
  __shared__ __attribute__((aligned(16))) P3Lds<FMT> L;
  int by, bx;
  gemm_tile_order(g.e.kmode, by, bx);
  p3_body<FMT, ER, TS>(g, by, bx, L);
}

// The off-diagonal b x b blocks (rows of the first half, columns of the second half) of every 2b-block on the diagonal of
// an n x n product: the tiles of one doubling level of a triangular inverse (tri_inverse), b a multiple of 128.
template <int FMT, int ER = 2, bool TS = false>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_p3_blk(P3Args g, int b) {
  __shared__ __attribute__((aligned(16))) P3Lds<FMT> L;
  const int tb = b / 128, per = tb * tb;
  const int p = blockIdx.x / per, r = blockIdx.x % per;
  const int by = (p * 2 * b) / 128 + r / tb, bx = (p * 2 * b + b) / 128 + r % tb;
  if (by * 128 >= g.e.M || bx * 128 >= g.e.N) return;
  p3_body<FMT, ER, TS>(g, by, bx, L);
}

// two independent products in one grid (see k_gemm_x3_pair)
struct P3Pair { P3Args g[2]; int tiles0, tx0, tx1, tiles1; };

template <int FMT, int ER = 2, bool TS = false>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_p3_pair(P3Pair p) {
  __shared__ __attribute__((aligned(16))) P3Lds<FMT> L;
  int which, id;
  pair_block(blockIdx.x, p.tiles0, p.tiles1, which, id);
  const P3Args& g = p.g[which];
  int by, bx;
  gemm_tile_from_id(id, (g.e.M + 127) / 128, which ? p.tx1 : p.tx0, g.e.kmode, by, bx);
  p3_body<FMT, ER, TS>(g, by, bx, L);
}

// The two gradient products of the update (psgd.py:175-176) in one grid: upper tiles only (the planes of a triu result are
// never read below the diagonal), in row-major order of the upper triangle, and with the LAST tiles of the second product
// split along K (P3Split).  All these tiles cost the same 2 x K / 32 steps, so without the split 1056 tiles on 512 block
// slots are two full rounds and a third with 32 tiles (half a millisecond of idle CUs at 4096^2); with the last 64 tiles
// as 512 eighth-size items the tail is one round of 32 steps.
struct P3Grad { P3Args g[2]; int T0, T1, n0, n1, nsplit, nchunk; float* scratch; unsigned* cnt; int order; };

__device__ __forceinline__ void upper_tile(int idx, int T, int& r, int& c) {      // idx-th tile (r <= c) of a T x T upper triangle
  const float b = 2.0f * T + 1.0f;
  r = (int)((b - sqrtf(b * b - 8.0f * idx)) * 0.5f);
  while (r > 0 && idx < r * T - (r * (r - 1)) / 2) --r;                           // (float rounding)
  while (idx >= (r + 1) * T - ((r + 1) * r) / 2) ++r;
  c = r + idx - (r * T - (r * (r - 1)) / 2);
}

// idx-th tile (r <= c) of a T x T upper triangle in PATCH order: 4 x 4 tile patches in row-major order of the patch triangle
// (T a multiple of 4), the tiles of a patch in row-major order (10 in a diagonal patch, 16 elsewhere)
__device__ __forceinline__ void upper_tile_patched(int idx, int T, int& r, int& c) {
  const int P = T >> 2;
  int pr = 0;
  for (;; ++pr) {                                       // patch row: 10 + 16 (P - pr - 1) tiles
    const int in_row = 10 + 16 * (P - pr - 1);
    if (idx < in_row) break;
    idx -= in_row;
  }
  int pc = pr, tr, tc;
  if (idx < 10) {
    upper_tile(idx, 4, tr, tc);
  } else {
    idx -= 10;
    pc = pr + 1 + (idx >> 4);
    tr = (idx & 15) >> 2; tc = idx & 3;
  }
  r = 4 * pr + tr; c = 4 * pc + tc;
}

template <int FMT, int ER = 2, bool TS = false>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_p3_grad(P3Grad p) {
  __shared__ __attribute__((aligned(16))) P3Lds<FMT> L;
  const int whole1 = p.n1 - p.nsplit;
  int id = blockIdx.x;
  // order (tuning key 17): 1 = every XCD (blocks b, b + 8, ...) works through a CONTIGUOUS run of the whole tiles' list, so
  // the tiles in flight on one L2 are neighbours of one or two tile rows; 2 = the same over 4 x 4 tile patches of the triangle
  const int whole = p.n0 + whole1;
  if (p.order && id < whole) {
    const int x = id & 7, j = id >> 3, q = whole >> 3, rem = whole & 7;
    id = x * q + min(x, rem) + j;
  }
  int r, c;
  const bool patched = p.order == 2 && !(p.T0 & 3) && !(p.T1 & 3);
  // ONE instance of the body for the three kinds of work item (round 6: three inlined copies were past the reach of a short
  // branch, and the branch relaxation's emergency spill slot gave the kernel a private segment it never touches)
  const int which = id < p.n0 ? 0 : 1;
  int tile = id - (which ? p.n0 : 0);
  P3Split sp{-1, 0, nullptr, nullptr};
  if (id >= p.n0 + whole1) {
    const int s = id - p.n0 - whole1, t = s / p.nchunk;
    tile = whole1 + t;
    sp = P3Split{s % p.nchunk, p.nchunk, p.scratch + (long)t * p.nchunk * (64 * kThreads), p.cnt + t};
  }
  const int T = which ? p.T1 : p.T0;
  if (patched) upper_tile_patched(tile, T, r, c); else upper_tile(tile, T, r, c);
  p3_body<FMT, ER, TS>(p.g[which], r, c, L, sp);
}

// One triu product with few output tiles and a long K (the gradient of the dense factor of a sparse Kron format at
// embedding shapes: 1000 x 1000 outputs, K = 30000, twice): upper tiles only, every tile's K steps dealt to `nchunk`
// blocks (P3Split).  On the in-GEMM split kernel this product ran on 64 workgroups for 1.8 ms.
template <int FMT>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_p3_splitk(P3Args g, int T, int nchunk, float* scratch, unsigned* cnt) {
  __shared__ __attribute__((aligned(16))) P3Lds<FMT> L;
  const int t = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
  int r, c;
  upper_tile(t, T, r, c);
  p3_body<FMT, FMT ? 0 : 2>(g, r, c, L, P3Split{chunk, nchunk, scratch + (long)t * nchunk * (64 * kThreads), cnt + t});
}

// The same for any product (all tiles, in the usual tile order): shapes with few output tiles -- a 128 x 4096 apply is one
// row of 32 tiles, each a chain of 128 K steps -- leave most of the chip idle and are bound by that chain.
template <int FMT, bool TS = false>
__global__ __launch_bounds__(kThreads, 2) void k_gemm_p3_splitk_rect(P3Args g, int ty, int tx, int nchunk, float* scratch,
                                                                     unsigned* cnt) {
  __shared__ __attribute__((aligned(16))) P3Lds<FMT> L;
  const int t = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk;
  int by, bx;
  gemm_tile_from_id(t, ty, tx, g.e.kmode, by, bx);
  p3_body<FMT, 2, TS>(g, by, bx, L, P3Split{chunk, nchunk, scratch + (long)t * nchunk * (64 * kThreads), cnt + t});
}

// fp32 view X(r, c) = X[r * rs + c * cs], r < R, c < C  ->  planes with x = r, k = c (zeros outside R x C; the grid covers
// the padded extents) and, optionally, the planes of the transposed view (x = c, k = r) from the same read.  64 x 64
// tiles through LDS so that the read (along the view's contiguous dimension) and both writes are coalesced.
// FMT = 1: the f16 x 2 planes of X 2^e, e from meta->amax (k_absmax ran before); the first block completes *meta.
// SplitOpt (the triangular inverse): tri = 1 keeps c >= r of the view, 2 keeps c <= r (zeros elsewhere); blk > 0 with
// off = 0: only the tiles inside the blk x blk blocks on the diagonal are made, off = 1: only the (first half, second half)
// off-diagonal quarter of every such block; neg: the planes of -X.
struct SplitOpt { int tri, blk, off, neg; };
// At most 48 registers per lane (40 as compiled: keep it so).  Split launches sit inside chains that run BESIDE full-chip plane products of the
// other stream (232 registers, two workgroups per CU = 464 of a SIMD's 512): at 54 registers a split workgroup could not join a CU
// until one of the product's workgroups had finished its whole K loop, and a 15-us split took 105-160 us (profiles/r04_kron_update_trace_f32.txt).
template <int FMT>
__device__ __forceinline__ void split3_body(const float* __restrict__ X, long rs, long cs, int R, int C,
                                            __bf16* __restrict__ P, long ts, long ps, __bf16* __restrict__ Pt,
                                            long tts, long tps, PlaneMeta* meta, const float* __restrict__ part,
                                            int npart, SplitOpt opt, int bx, int by, int bz, float (*S)[65],
                                            int* te = nullptr, int* tet = nullptr) {       // te / tet: tile-exponent tables of P / Pt to fill
  int r0 = by * 64, c0 = bx * 64;
  const int tid = threadIdx.x;
  if (opt.blk) {
    // a COMPACT grid: x, y run over the 64-tiles of one blk x blk diagonal block (off: of its first-half x second-half quarter),
    // z over the blocks -- with one workgroup per tile of the whole matrix (n = 4096, blk = 512: 4096 workgroups, 7/8 of which
    // return at once) this launch took 157 us instead of 16 beside a full-chip product: every workgroup queues for a slot
    r0 += bz * opt.blk;
    c0 += bz * opt.blk + (opt.off ? opt.blk / 2 : 0);
    const long xp = P ? ts / 32 : tps / (tts / 32), kp = P ? ps / (ts / 32) : tts / 32;     // padded extents of the view (x = r, k = c)
    if (r0 >= xp || c0 >= kp) return;                            // (the last block of a ragged matrix)
  }
  constexpr int NPL = FMT ? 2 : 3;
  float sc = 1.0f;
  if constexpr (FMT == 1) {
    // max|X| = the maximum of the partial maxima the launch ahead left (k_absmax, the balance): every block reduces them itself
    float amax = 0.0f;
    for (int i = tid; i < npart; i += kThreads) amax = amaxf(amax, part[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = amaxf(amax, __shfl_down(amax, off, 64));
    if ((tid & 63) == 0) S[0][tid >> 6] = amax;
    __syncthreads();
    amax = amaxf(amaxf(S[0][0], S[0][1]), amaxf(S[0][2], S[0][3]));
    __syncthreads();
    sc = plane_scale_of_bound(amax);
    // (the first block that exists under a block filter: tile (0, 0), or the first off-diagonal quarter's first tile)
    const bool first = opt.off ? (r0 == 0 && c0 == opt.blk / 2) : (r0 == 0 && c0 == 0);
    if (first && tid == 0) { meta->scale = sc; meta->inv = 1.0f / sc; meta->amax = amax; meta->bound = amax; }
    // (planes that products with tile scales read: every 128-tile this launch writes carries the matrix's one exponent)
    if (tid == 0 && !(r0 & 127) && !(c0 & 127)) {
      if (te) te[(r0 >> 7) * kTeLd + (c0 >> 7)] = p3_exp_of_scale(sc);
      if (tet) tet[(c0 >> 7) * kTeLd + (r0 >> 7)] = p3_exp_of_scale(sc);
    }
    if (opt.neg) sc = -sc;
  }
  auto split_pair = [&](float x0, float x1, unsigned (&q)[3]) {
    if constexpr (FMT == 1) {
      unsigned t[2];
      split2h_pair(x0 * sc, x1 * sc, t);
      q[0] = t[0]; q[1] = t[1]; q[2] = 0u;
    } else {
      split3_pair(x0, x1, q);
    }
  };
  // (clamped addresses, every load of a half issued before its first use: a guarded load compiles to load-then-wait.  Two halves of
  //  eight loads, not sixteen at once: the kernel has to stay within 48 registers -- see the launch bounds)
  const bool masked = (opt.tri == 1 && c0 + 63 < r0) || (opt.tri == 2 && c0 > r0 + 63);      // nothing of this tile is kept: zeros, no loads
  if (masked) {
    for (int e = tid; e < 64 * 64; e += kThreads) S[e >> 6][e & 63] = 0.0f;
  } else if (cs == 1 || rs != 1) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        x[i] = X[(long)min(r0 + (tid >> 6) + 4 * (i + 8 * hh), R - 1) * rs + (long)min(c0 + (tid & 63), C - 1) * cs];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = (tid >> 6) + 4 * (i + 8 * hh), c = tid & 63;
        const bool keep = opt.tri == 0 || (opt.tri == 1 ? c0 + c >= r0 + r : c0 + c <= r0 + r);
        S[r][c] = (r0 + r < R && c0 + c < C && keep) ? x[i] : 0.0f;
      }
    }
  } else {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        x[i] = X[(long)min(r0 + (tid & 63), R - 1) + (long)min(c0 + (tid >> 6) + 4 * (i + 8 * hh), C - 1) * cs];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = (tid >> 6) + 4 * (i + 8 * hh), r = tid & 63;
        const bool keep = opt.tri == 0 || (opt.tri == 1 ? c0 + c >= r0 + r : c0 + c <= r0 + r);
        S[r][c] = (r0 + r < R && c0 + c < C && keep) ? x[i] : 0.0f;
      }
    }
  }
  __syncthreads();
  if (P) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 4) + 16 * i, c = (tid & 15) * 4;
      unsigned q0[3], q1[3];
      split_pair(S[r][c], S[r][c + 1], q0);
      split_pair(S[r][c + 2], S[r][c + 3], q1);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<uint2*>(P + pl * ps + p3_index(ts, r0 + r, c0 + c)) = make_uint2(q0[pl], q1[pl]);
    }
  }
  if (Pt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (tid >> 4) + 16 * i, r = (tid & 15) * 4;
      unsigned q0[3], q1[3];
      split_pair(S[r][c], S[r + 1][c], q0);
      split_pair(S[r + 2][c], S[r + 3][c], q1);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<uint2*>(Pt + pl * tps + p3_index(tts, c0 + c, r0 + r)) = make_uint2(q0[pl], q1[pl]);
    }
  }
}

template <int FMT>
__global__ __launch_bounds__(kThreads) void k_split3(const float* __restrict__ X, long rs, long cs, int R, int C,
                                                     __bf16* __restrict__ P, long ts, long ps, __bf16* __restrict__ Pt,
                                                     long tts, long tps, PlaneMeta* meta, const float* __restrict__ part,
                                                     int npart, SplitOpt opt, int* te = nullptr, int* tet = nullptr) {
  __shared__ float S[64][65];
  split3_body<FMT>(X, rs, cs, R, C, P, ts, ps, Pt, tts, tps, meta, part, npart, opt, blockIdx.x, blockIdx.y, blockIdx.z, S, te, tet);
}
// two independent splits in one launch (the two gradients of the large update): job 0 on the rows y < y0 of the grid, job 1 behind
struct SplitJob { const float* X; long rs, cs; int R, C; __bf16* P; long ts, ps; __bf16* Pt; long tts, tps; PlaneMeta* meta; const float* part; int npart; SplitOpt opt; int gx; };
template <int FMT>
__global__ __launch_bounds__(kThreads) void k_split3_two(SplitJob a, SplitJob b, int y0) {
  __shared__ float S[64][65];
  const bool second = (int)blockIdx.y >= y0;
  const SplitJob j = second ? b : a;
  if ((int)blockIdx.x >= j.gx) return;
  split3_body<FMT>(j.X, j.rs, j.cs, j.R, j.C, j.P, j.ts, j.ps, j.Pt, j.tts, j.tps, j.meta, j.part, j.npart, j.opt, blockIdx.x,
                   second ? blockIdx.y - y0 : blockIdx.y, 0, S);
}

// The inverse of every 128 x 128 diagonal block of an upper-triangular Q [n x n] from its inverted 32 x 32 diagonal blocks
// (`dinv`: k_tri_inv32 / the balance launch): levels 32 and 64 of the doubling  [A B; 0 C]^-1 = [A^-1, -A^-1 B C^-1; 0, C^-1]
// inside one workgroup, in LDS with plain fp32 fma chains (the matrix-core levels start at 128, tri_inverse).  Rows and
// columns past n count as identity.  Writes the block of Inv [n x n] (zeros below the diagonal) and max|.| into *amax.
template <int N> struct IntK { static constexpr int value = N; };
__global__ __launch_bounds__(kThreads) void k_tri_inv128(const float* __restrict__ Q, int n, const float* __restrict__ dinv,
                                                         float* __restrict__ Inv, float* amax) {
  extern __shared__ __attribute__((aligned(16))) float inv128_sm[];
  constexpr int PT = 129;
  float (*Tb)[PT] = reinterpret_cast<float (*)[PT]>(inv128_sm);
  float (*Ib)[PT] = reinterpret_cast<float (*)[PT]>(inv128_sm + 128 * PT);
  const int j0 = 128 * blockIdx.x, tid = threadIdx.x;
  // (clamped addresses, eight loads in flight: a guarded load compiles to load-then-wait, 64 L2 round trips in a row per thread
  //  -- 47 us for a launch whose arithmetic is ~10)
#pragma unroll 8
  for (int it = 0; it < 128 * 128 / kThreads; ++it) {
    const int e = tid + it * kThreads, r = e >> 7, c = e & 127;
    const float q = Q[(long)min(j0 + r, n - 1) * n + min(j0 + c, n - 1)];
    Tb[r][c] = (j0 + r < n && j0 + c < n && c >= r) ? q : ((r == c) ? 1.0f : 0.0f);
    Ib[r][c] = 0.0f;
  }
  __syncthreads();
  const int nb32 = (n + 31) / 32;
#pragma unroll 8
  for (int it = 0; it < 4 * 1024 / kThreads; ++it) {                // the four inverted 32-blocks
    const int e = tid + it * kThreads, t = e >> 10, r = (e >> 5) & 31, c = e & 31;
    const float d = dinv[(long)min(j0 / 32 + t, nb32 - 1) * 1024 + r * 32 + c];
    Ib[32 * t + r][32 * t + c] = (j0 + 32 * t < n) ? d : ((r == c) ? 1.0f : 0.0f);
  }
  __syncthreads();
  // merge the pair of inverted b-blocks at a0: W = B C^-1 (kept below the diagonal meanwhile), X12 = -A^-1 W
  auto merge = [&](int a0, int b, int m0, int n0, auto TMc, auto TNc) {
    constexpr int TM = decltype(TMc)::value, TN = decltype(TNc)::value;
    float acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) acc[i][jn] = 0.0f;
    for (int k = 0; k < b; ++k) {
      float a[TM], c[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Tb[a0 + m0 + i][a0 + b + k];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) c[jn] = Ib[a0 + b + k][a0 + b + n0 + jn];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[i][jn] = fmaf(a[i], c[jn], acc[i][jn]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) Ib[a0 + b + m0 + i][a0 + n0 + jn] = acc[i][jn];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) acc[i][jn] = 0.0f;
    for (int k = 0; k < b; ++k) {
      float a[TM], c[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ib[a0 + m0 + i][a0 + k];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) c[jn] = Ib[a0 + b + k][a0 + n0 + jn];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[i][jn] = fmaf(a[i], c[jn], acc[i][jn]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) Ib[a0 + m0 + i][a0 + b + n0 + jn] = -acc[i][jn];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) Ib[a0 + b + m0 + i][a0 + n0 + jn] = 0.0f;
    __syncthreads();
  };
  {                                                                  // level 32: pairs (0,1) and (2,3); 2 x 4 outputs per thread
    const int pr = tid >> 7, u = tid & 127;
    merge(64 * pr, 32, (u >> 3) * 2, (u & 7) * 4, IntK<2>{}, IntK<4>{});
  }
  merge(0, 64, (tid >> 4) * 4, (tid & 15) * 4, IntK<4>{}, IntK<4>{});             // level 64: 4 x 4 outputs per thread
  float m = 0.0f;
  for (int e = tid; e < 128 * 128; e += kThreads) {
    const int r = e >> 7, c = e & 127;
    if (j0 + r < n && j0 + c < n) {
      Inv[(long)(j0 + r) * n + j0 + c] = Ib[r][c];
      m = amaxf(m, fabsf(Ib[r][c]));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = amaxf(m, __shfl_down(m, off, 64));
  if ((tid & 63) == 0) atomic_amax(amax, m);
}

// max |X(o, i)|, X(o, i) = X[o * os + i], o < O, i < L (NaN propagates): block b leaves ITS maximum in part[b] and the
// split kernel behind reduces them.  A dense matrix is one run (O = 1); a column block of a row-major matrix is O runs of
// L.  The first block also clears zero[0 .. nzero): the maxima that the epilogues of the products behind accumulate.
__global__ __launch_bounds__(kThreads) void k_absmax(const float* __restrict__ X, long os, long O, long L, float* __restrict__ part,
                                                     float* __restrict__ zero, int nzero) {
  __shared__ float red[kThreads / 64];
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < nzero; i += kThreads) zero[i] = 0.0f;
  float m = 0.0f;
  if (O == 1) {
    const long n = L, stride = (long)gridDim.x * kThreads, t0 = (long)blockIdx.x * kThreads + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(X) & 15) == 0) {
      const float4* X4 = reinterpret_cast<const float4*>(X);
      const long n4 = n >> 2;
      for (long i = t0; i < n4; i += stride) {
        const float4 v = X4[i];
        m = amaxf(amaxf(m, fabsf(v.x)), amaxf(amaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
      }
      for (long i = (n4 << 2) + t0; i < n; i += stride) m = amaxf(m, fabsf(X[i]));
    } else {
      for (long i = t0; i < n; i += stride) m = amaxf(m, fabsf(X[i]));
    }
  } else {
    const bool vec = (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (os & 3) == 0 && (L & 3) == 0;
    for (long o = blockIdx.x; o < O; o += gridDim.x) {
      const float* __restrict__ row = X + o * os;
      if (vec) {
        for (long i = threadIdx.x * 4L; i < L; i += kThreads * 4L) {
          const float4 v = *reinterpret_cast<const float4*>(row + i);
          m = amaxf(amaxf(m, fabsf(v.x)), amaxf(amaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
        }
      } else {
        for (long i = threadIdx.x; i < L; i += kThreads) m = amaxf(m, fabsf(row[i]));
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = amaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) m = amaxf(m, red[w]);
    part[blockIdx.x] = m;
  }
}

// max|.| of the UPPER triangles of two row-major square matrices in one launch (the factors of a Kron apply: upper triangular by
// contract): blocks [0, ba) leave the partial maxima of A in partA[0 .. ba), the rest those of B in partB; a block reads rows
// b, b + nb, ... of its matrix from the (4-aligned) column of the diagonal on -- half the bytes of two flat scans, one launch.
// zero[0 .. nzero) as in k_absmax.  Contract: both base addresses 16-byte aligned, nA % 4 == nB % 4 == 0 (launcher: else two k_absmax).
__global__ __launch_bounds__(kThreads) void k_absmax_tri2(const float* __restrict__ A, int nA, float* __restrict__ partA, int ba,
                                                          const float* __restrict__ B, int nB, float* __restrict__ partB,
                                                          float* __restrict__ zero, int nzero) {
  __shared__ float red[kThreads / 64];
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < nzero; i += kThreads) zero[i] = 0.0f;
  const bool second = (int)blockIdx.x >= ba;
  const float* __restrict__ X = second ? B : A;
  const int n = second ? nB : nA, b = second ? blockIdx.x - ba : blockIdx.x, nb = second ? gridDim.x - ba : ba;
  float m = 0.0f;
  for (int r = b; r < n; r += nb) {
    const float* __restrict__ row = X + (long)r * n;
    for (int c = (r & ~3) + threadIdx.x * 4; c < n; c += kThreads * 4) {
      const float4 v = *reinterpret_cast<const float4*>(row + c);
      m = amaxf(amaxf(m, c >= r ? fabsf(v.x) : 0.0f), amaxf(amaxf(c + 1 >= r ? fabsf(v.y) : 0.0f, c + 2 >= r ? fabsf(v.z) : 0.0f), fabsf(v.w)));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = amaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) (second ? partB : partA)[b] = amaxf(amaxf(red[0], red[1]), amaxf(red[2], red[3]));
}

// (T = 128 is the exact-fp32 alternative to the split GEMM, psgd_kron_set_tuning(1, 0): two resident blocks per CU; with the
// K loops per fetch mode three blocks -- 168 registers -- would spill)
template <int T, int GK>
__global__ __launch_bounds__(kThreads, (T == 128 ? 2 : 1)) void k_gemm_f32(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) GemmLds<T, GK> L;
  // longest-K tiles first (see the bf16 kernel): an upper K bound grows with the tile index
  int by, bx;
  gemm_tile_order(g.kmode, by, bx);
  gemm_body<T, GK>(g, by * T, bx * T, L);
}

// tile configurations: 128 x 128 x 16 for large problems; 64 x 64 x 64 for small ones, where the
// grid cannot fill the chip and every K step costs a full (unhidden) load latency
constexpr int kSmallK = 64, kBigK = 16;

// One launch for the same stage of several independent problems (the layers of a small network):
// blockIdx.x -> (problem, tile) through the prefix sums of the tile counts.  64 x 64 tiles.
constexpr int kMaxBatch = 16;     // problems per batched launch (16 x 216-byte GemmArgs = 3.4 KiB of kernel arguments)
constexpr int kMaxLayers = 8;     // layers per group of the batched calls (two independent products per layer share a launch)
struct GemmBatch {
  int count;
  int tile_end[kMaxBatch];     // inclusive prefix sums of tiles
  GemmArgs g[kMaxBatch];
};

template <int T>
__global__ __launch_bounds__(kThreads) void k_gemm_f32_batched(GemmBatch b) {
  __shared__ __attribute__((aligned(16))) GemmLds<T, kSmallK> L;
  int p = 0;
  while (p + 1 < b.count && (int)blockIdx.x >= b.tile_end[p]) ++p;
  const int t = blockIdx.x - (p ? b.tile_end[p - 1] : 0);
  const GemmArgs& g = b.g[p];
  const int tn = (g.N + T - 1) / T;
  gemm_body<T, kSmallK>(g, (t / tn) * T, (t % tn) * T, L);
}

// ---------------------------------------------------------------------------------------------
// 32 x 32 tile body for problems that cannot fill the chip (LeNet5-size layers).  The block is alone on its CU, one
// wave per SIMD: every instruction of the K loop is exposed, and so is one memory latency per K step in gemm_body
// (one K tile of register prefetch).  Here (a) everything about a thread's 8 elements per operand tile that does not
// change with the K step is computed once (element offset, k index, LDS slot, row guard), (b) kSmallD K tiles are in
// flight in a ring of register stages, and (c) every global load is unconditional -- clamped address, value selected
// afterwards -- so the waits are counted vmcnt(N) instead of vmcnt(0).
constexpr int kSmallD = 4;

struct SmallSrc {
  const float* P; long cs; int khi;
  unsigned off[8];          // (x0 + x) * rs + kfix * cs   (32-bit: these problems are small)
  int kfix[8], sidx[8];     // k inside the tile; LDS index k * 48 + x
  unsigned xok;             // bit u: x0 + x < X
};

__device__ __forceinline__ void small_src_init(SmallSrc& s, const TileSrc& t, int x0, int khi) {
  s.P = t.P; s.cs = t.cs; s.khi = khi; s.xok = 0u;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = threadIdx.x + kThreads * u;
    int x, k;
    if (t.cs == 1) { k = e % kSmallK; x = e / kSmallK; } else { x = e % 32; k = e / 32; }
    const bool ok = x0 + x < t.X;
    s.off[u] = ok ? (unsigned)((long)(x0 + x) * t.rs + (long)k * t.cs) : 0u;
    s.kfix[u] = k; s.sidx[u] = k * 48 + x;
    s.xok |= ok ? (1u << u) : 0u;
  }
}

// Loads only (masked elements read a valid address): the values are first USED by small_commit D stages later, so that
// no s_waitcnt for them sits between a stage's loads and its MFMAs.
__device__ __forceinline__ void small_fetch(const SmallSrc& s, int k0, bool live, float (&r)[8]) {
  const float* Pk = s.P + (long)(live ? k0 : 0) * s.cs;              // uniform; a dead tile reads from the base
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const bool ok = live && ((s.xok >> u) & 1u) && (k0 + s.kfix[u] < s.khi);
    r[u] = Pk[ok ? s.off[u] : 0u];
  }
}

__device__ __forceinline__ void small_commit(const SmallSrc& s, int k0, float mul, const float (&r)[8], float* S) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const bool ok = ((s.xok >> u) & 1u) && (k0 + s.kfix[u] < s.khi);
    S[s.sidx[u]] = ok ? r[u] * mul : 0.0f;
  }
}

__device__ __forceinline__ void gemm_body_small(const GemmArgs& g, int m0, int n0, GemmLds<32, kSmallK>& L) {
  constexpr int T = 32, GK = kSmallK, D = kSmallD;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;
  const bool tri_skip = (g.epi == EPI_TRIU_MAX || g.sym) && (m0 >= n0 + T);
  if (g.sym && tri_skip) return;

  f32x4 acc[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
  float a_mul = 1.0f;
  if (g.scale_max) a_mul = g.step / (g.scale_max[0] + g.tiny);

  int klo[2] = {0, 0}, khi[2] = {0, 0}, nk[2] = {0, 0};
  if (!tri_skip) {
    for (int p = 0; p < 2; ++p) {
      if (p == 1 && !g.A2) break;
      const int K = p ? g.K2 : g.K, km = p ? g.kmode2 : g.kmode;
      int lo = 0, hi = K;
      if (km & KLO_M) lo = max(lo, m0);
      if (km & KLO_N) lo = max(lo, n0);
      if (km & KHI_M) hi = min(hi, m0 + T);
      if (km & KHI_N) hi = min(hi, n0 + T);
      lo = (lo / GK) * GK;
      klo[p] = lo; khi[p] = hi; nk[p] = hi > lo ? (hi - lo + GK - 1) / GK : 0;
    }
  }
  float* LA = &L.A[0][0][0];
  float* LB = &L.B[0][0][0];
  constexpr int kBuf = GK * (T + 16);
  // one operand pair at a time (the second pair of a dual product restarts the ring: one more latency, no branches
  // around the loads inside the loop)
  auto run_pair = [&](const TileSrc& ta, const TileSrc& tb, int lo, int hi, int n, float mul) {
    SmallSrc sa, sb;
    small_src_init(sa, ta, m0, hi);
    small_src_init(sb, tb, n0, hi);
    float ra[D][8], rb[D][8];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      small_fetch(sa, lo + d * GK, d < n, ra[d]);
      small_fetch(sb, lo + d * GK, d < n, rb[d]);
    }
    for (int t0 = 0; t0 < n; t0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int t = t0 + d, buf = d & 1;             // D is even: the LDS buffer alternates with d
        if (t < n) {                                    // uniform; LDS traffic and MFMAs only (no global access inside)
          small_commit(sa, lo + t * GK, mul, ra[d], LA + buf * kBuf);
          small_commit(sb, lo + t * GK, 1.0f, rb[d], LB + buf * kBuf);
        }
        small_fetch(sa, lo + (t + D) * GK, t + D < n, ra[d]);      // unconditional refill of the stage just consumed
        small_fetch(sb, lo + (t + D) * GK, t + D < n, rb[d]);
        if (t < n) {
          __syncthreads();
#pragma unroll
          for (int kk = 0; kk < GK / 4; ++kk) {
            const int kr = kk * 4 + (lane >> 4);
            const float a = L.A[buf][kr][wm * 16 + (lane & 15)];
            const float b = L.B[buf][kr][wn * 16 + (lane & 15)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0][0], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();                                   // the next pair may overwrite buffer 0
  };
  if (nk[0] > 0) run_pair(TileSrc{g.A, g.a_rs, g.a_cs, g.M}, TileSrc{g.B, g.b_cs, g.b_rs, g.N}, klo[0], khi[0], nk[0], a_mul);
  if (nk[1] > 0) run_pair(TileSrc{g.A2, g.a2_rs, g.a2_cs, g.M}, TileSrc{g.B2, g.b2_cs, g.b2_rs, g.N}, klo[1], khi[1], nk[1], -a_mul);
  gemm_epilogue<32>(g, acc, m0, n0);
}

__global__ __launch_bounds__(kThreads) void k_gemm_small(GemmBatch b) {
  __shared__ __attribute__((aligned(16))) GemmLds<32, kSmallK> L;
  int p = 0;
  while (p + 1 < b.count && (int)blockIdx.x >= b.tile_end[p]) ++p;
  const int t = blockIdx.x - (p ? b.tile_end[p - 1] : 0);
  const GemmArgs& g = b.g[p];
  const int tn = (g.N + 31) / 32;
  gemm_body_small(g, (t / tn) * 32, (t % tn) * 32, L);
}

// one problem: the kernel arguments are one GemmArgs, not the 8-problem table
__global__ __launch_bounds__(kThreads) void k_gemm_small_one(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) GemmLds<32, kSmallK> L;
  gemm_body_small(g, blockIdx.y * 32, blockIdx.x * 32, L);
}

// Solve  y[i,:] Q = x[i,:]  for nvec independent vectors i, Q upper-triangular [n,n] row-major:
//   y[i,j] = (x[i,j] - sum_{k<j} y[i,k] Q[k,j]) / Q[j,j]
// Element (i,j) of X / Y lives at  i*si + j*sj.  With (si,sj) = (ld,1) this is the right solve
// Y Q = X on row-major [nvec,n]; with (si,sj) = (1,ld) it is Q'Y = X on row-major [n,nvec]
// (tf.linalg.triangular_solve(Q, X, lower=False, adjoint=True), psgd.py:174).
// Q has leading dimension ldq (a diagonal block of a larger factor can be passed); X may alias Y.
// Grams Q'Q of up to 16 small upper-triangular factors in one launch (the factor-only half of the batched apply: two
// per layer).  Compact descriptors: sixteen full GemmArgs would not fit the kernel-argument segment.
constexpr int kMaxGrams = 16;
struct GramBatch {
  int count;
  int tile_end[kMaxGrams];
  const float* Q[kMaxGrams];
  float* P[kMaxGrams];
  int n[kMaxGrams];
};

__host__ __device__ inline GemmArgs gram_args(const float* Q, float* P, int n) {
  GemmArgs g = {};
  g.A = Q; g.a_rs = 1; g.a_cs = n;          // A = Q' (stored transposed)
  g.B = Q; g.b_rs = n; g.b_cs = 1;
  g.C = P; g.ldc = n; g.M = n; g.N = n; g.K = n;
  g.epi = EPI_STORE;
  g.kmode = KHI_M | KHI_N;                  // Q upper-triangular: k <= min(m, n)
  g.sym = 1;
  return g;
}

template <int T>
__global__ __launch_bounds__(kThreads) void k_gram_batched(GramBatch b) {
  int p = 0;
  while (p + 1 < b.count && (int)blockIdx.x >= b.tile_end[p]) ++p;
  const int t = blockIdx.x - (p ? b.tile_end[p - 1] : 0);
  const GemmArgs g = gram_args(b.Q[p], b.P[p], b.n[p]);
  const int tn = (g.N + T - 1) / T;
  if constexpr (T == 32) {
    __shared__ __attribute__((aligned(16))) GemmLds<32, kSmallK> L;
    gemm_body_small(g, (t / tn) * 32, (t % tn) * 32, L);
  } else {
    __shared__ __attribute__((aligned(16))) GemmLds<T, kSmallK> L;
    gemm_body<T, kSmallK>(g, (t / tn) * T, (t % tn) * T, L);
  }
}

struct TrsmArgs {
  const float* Q; int n, ldq;
  const float* X; float* Y;
  int nvec; long si, sj;        // strides of Y (and of X unless xi/xj are set)
  long xi, xj;                  // strides of X; 0,0 = same as Y
};

__device__ __forceinline__ void trsm_body(const float* __restrict__ Q, int n, int ldq, const float* X, float* Y,
                                          int nvec, long si, long sj, long xi, long xj, int v0,
                                          float (*red)[64][33], float (*Qd)[32]) {
  if (xi == 0 && xj == 0) { xi = si; xj = sj; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int j0 = 0; j0 < n; j0 += 32) {
    const int jw = (n - j0 < 32) ? (n - j0) : 32;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 4 * w; k < j0; k += 16) {
      const int kk = k + (lane >> 4);
      float a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int vi = v0 + i * 16 + (lane & 15);
        a[i] = (vi < nvec) ? Y[vi * si + kk * sj] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int jj = j0 + j * 16 + (lane & 15);
        b[j] = (jj < n) ? Q[(long)kk * ldq + jj] : 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[w][i * 16 + (lane >> 4) * 4 + e][j * 16 + (lane & 15)] = acc[i][j][e];
    for (int e = threadIdx.x; e < 1024; e += kThreads) {
      const int r = e >> 5, c = e & 31;
      Qd[r][c] = (j0 + r < n && j0 + c < n) ? Q[(long)(j0 + r) * ldq + j0 + c] : (r == c ? 1.0f : 0.0f);
    }
    __syncthreads();
    if (w == 0) {
      const int v = v0 + lane;
      const bool vok = v < nvec;
      float rr[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float x = (vok && j < jw) ? X[v * xi + (long)(j0 + j) * xj] : 0.0f;
        rr[j] = x - (((red[0][lane][j] + red[1][lane][j]) + red[2][lane][j]) + red[3][lane][j]);
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float y = rr[j] / Qd[j][j];
        rr[j] = y;
#pragma unroll
        for (int j2 = j + 1; j2 < 32; ++j2) rr[j2] = fmaf(-y, Qd[j][j2], rr[j2]);
      }
      if (vok) {
#pragma unroll
        for (int j = 0; j < 32; ++j)
          if (j < jw) Y[v * si + (long)(j0 + j) * sj] = rr[j];
      }
    }
    __syncthreads();   // workgroup-scope fence + barrier: the other waves may now read this Y block
  }
}

__global__ __launch_bounds__(kThreads) void k_trsm_ut(const float* __restrict__ Q, int n, int ldq, const float* X,
                                                      float* Y, int nvec, long si, long sj, long xi, long xj) {
  __shared__ float red[4][64][33];
  __shared__ float Qd[32][32];
  trsm_body(Q, n, ldq, X, Y, nvec, si, sj, xi, xj, blockIdx.x * 64, red, Qd);
}

// Inverses of the 32 x 32 diagonal sub-blocks of an upper-triangular Q (identity-padded at the ragged end):
// Dinv[b] = Q[32b:32b+32, 32b:32b+32]^-1, by substitution, lane = row of the inverse.  One wave per block; all
// blocks of a factor in parallel (~5 us), after which no solve below contains a serial per-vector chain.
__global__ __launch_bounds__(64) void k_tri_inv32(const float* __restrict__ Q, int n, int ldq, float* __restrict__ Dinv) {
  __shared__ float Qd[32][33];
  const int lane = threadIdx.x, b = blockIdx.x, j0 = b * 32;
  for (int e = lane; e < 1024; e += 64) {
    const int r = e >> 5, c = e & 31;
    Qd[r][c] = (j0 + r < n && j0 + c < n) ? Q[(long)(j0 + r) * ldq + j0 + c] : (r == c ? 1.0f : 0.0f);
  }
  __syncthreads();
  if (lane < 32) {
    float rr[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) rr[j] = (j == lane) ? 1.0f : 0.0f;      // row `lane` of I; solve x Qd = e_lane
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float y = rr[j] / Qd[j][j];
      rr[j] = y;
#pragma unroll
      for (int j2 = j + 1; j2 < 32; ++j2) rr[j2] = fmaf(-y, Qd[j][j2], rr[j2]);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) Dinv[(long)b * 1024 + lane * 32 + j] = rr[j];
  }
}

// Solve y Q = x for n <= 512 with the workgroup's strip of 64 vectors resident in LDS and the diagonal
// sub-blocks pre-inverted (k_tri_inv32): every 32-wide sub-step is  Y_s = R_s Dinv_s  followed by the
// trailing update  S[:, later] -= Y_s Q[s-rows, later],  both on MFMA with the A operand read from LDS.
// (Blocked TRSM via small triangular inverses + GEMM is the standard GPU formulation; the substitution
// kernel above is kept for the batched small-layer path.)  Dynamic LDS: 64 x pitch + 32 x 33 floats.
constexpr int kStripN = 512;

// VB = vectors per workgroup (64, or 16 so that a solve with few thousand vectors still covers every CU)
template <int VB>
__device__ __forceinline__ void trsm_inv_body(const TrsmArgs& t, const float* __restrict__ Dinv, int v0, float* S,
                                              int pitch, float (*Dd)[33]) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long xi = (t.xi == 0 && t.xj == 0) ? t.si : t.xi, xj = (t.xi == 0 && t.xj == 0) ? t.sj : t.xj;
  const int n = t.n;
  const int npad = (n + 31) & ~31;
  // strip in: 8 independent loads in flight per thread (the plain loop paid one memory latency per element: the load
  // and store phases together were ~3/4 of the kernel's time)
  for (int e0 = tid; e0 < VB * npad; e0 += kThreads * 8) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kThreads;
      int v, j;
      if (xj == 1) { j = e % npad; v = e / npad; } else { v = e % VB; j = e / VB; }
      x[u] = (e < VB * npad && v0 + v < t.nvec && j < n) ? t.X[(long)(v0 + v) * xi + (long)j * xj] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * kThreads;
      int v, j;
      if (xj == 1) { j = e % npad; v = e / npad; } else { v = e % VB; j = e / VB; }
      if (e < VB * npad) S[v * pitch + j] = x[u];
    }
  }
  float dnext[1024 / kThreads];          // next sub-step's inverted diagonal block, in flight during this sub-step
#pragma unroll
  for (int u = 0; u < 1024 / kThreads; ++u) dnext[u] = Dinv[tid + kThreads * u];
  for (int j0 = 0; j0 < n; j0 += 32) {
#pragma unroll
    for (int u = 0; u < 1024 / kThreads; ++u) { const int e = tid + kThreads * u; Dd[e >> 5][e & 31] = dnext[u]; }
    if (j0 + 32 < n) {
#pragma unroll
      for (int u = 0; u < 1024 / kThreads; ++u) dnext[u] = Dinv[(long)((j0 >> 5) + 1) * 1024 + tid + kThreads * u];
    }
    __syncthreads();
    if (w < VB / 16) {   // Y_s = R_s Dinv_s : wave w owns vector block w (rows 16w..16w+15), reads and rewrites only those rows
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = kk * 4 + (lane >> 4);
        const float a = S[(w * 16 + (lane & 15)) * pitch + j0 + k];
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Dd[k][c * 16 + (lane & 15)], acc[c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(w * 16 + (lane >> 4) * 4 + e) * pitch + j0 + c * 16 + (lane & 15)] = acc[c][e];
    }
    __syncthreads();
    const int c_begin = j0 + 32;
    const int ncb = (n > c_begin) ? (n - c_begin + 15) / 16 : 0;
    // trailing update of the strip: this wave's column blocks cb = w, w + 4, ... (at most kStripN / 64 = 8 of them).
    // All their Q panels are requested before the first one is used, so one memory latency is paid per 32-wide
    // sub-step instead of one per column block (the solve was a chain of ~100 exposed latencies per strip).
    constexpr int kMaxIt = kStripN / 64;
    float bq[kMaxIt][8];
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int cb = w + 4 * it;
      const int col = c_begin + cb * 16 + (lane & 15);
      const bool cok = (cb < ncb) && (col < n);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) bq[it][kk] = cok ? t.Q[(long)(j0 + kk * 4 + (lane >> 4)) * t.ldq + col] : 0.0f;
    }
#pragma unroll
    for (int it = 0; it < kMaxIt; ++it) {
      const int cb = w + 4 * it;
      if (cb >= ncb) break;
      const int col = c_begin + cb * 16 + (lane & 15);
      f32x4 acc[VB / 16];
#pragma unroll
      for (int i = 0; i < VB / 16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = S[(i * 16 + (lane >> 4) * 4 + e) * pitch + col];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = j0 + kk * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < VB / 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(-S[(i * 16 + (lane & 15)) * pitch + k], bq[it][kk], acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < VB / 16; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(i * 16 + (lane >> 4) * 4 + e) * pitch + col] = acc[i][e];
    }
    __syncthreads();
  }
#pragma unroll 8
  for (int e = tid; e < VB * n; e += kThreads) {
    int v, j;
    if (t.sj == 1) { j = e % n; v = e / n; } else { v = e % VB; j = e / VB; }
    if (v0 + v < t.nvec) t.Y[(long)(v0 + v) * t.si + (long)j * t.sj] = S[v * pitch + j];
  }
}

template <int VB>
__global__ __launch_bounds__(kThreads) void k_trsm_ut_inv(TrsmArgs t, const float* __restrict__ Dinv) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  const int pitch = ((t.n + 31) & ~31) + 2;          // == 2 (mod 32): MFMA A-operand reads (v, k) hit distinct banks
  trsm_inv_body<VB>(t, Dinv, blockIdx.x * VB, dyn_lds, pitch, reinterpret_cast<float (*)[33]>(dyn_lds + VB * pitch));
}

// Register-resident form of the strip solve (the default): the workgroup's 16 vectors x n <= 512 columns live in MFMA
// accumulators for the whole strip, TRANSPOSED -- tile (sb, h) of wave w = sb mod 4 holds
//   acc[e] = R[v = lane & 15][col = 32 sb + 16 h + 4 (lane >> 4) + e]
// i.e. the C layout of the tile R', rows = columns of the strip, columns = vectors.  In that layout an accumulator
// register IS a B operand of v_mfma_f32_16x16x4_f32 for the k order (16 h + 4 g + e), so
//   Y_s' = Dinv_s' R_s'            (8 MFMAs per 16-column half, A = Dinv_s read from global/L2)
//   R'[later] -= Q[s, later]' Y_s' (8 MFMAs per tile, A = the Q panel read from global/L2)
// chain through registers with no transposition through LDS.  Per 32-wide sub-step the owner wave (s mod 4) solves
// its block, stores it, and hands its 8 registers to the other waves through a 2-KiB LDS slot (double-buffered: one
// barrier per sub-step); every wave then updates the tiles it owns.  The Q panels and the inverted diagonal block of
// the next sub-step are requested before the barrier.
// The LDS-resident form above spends 68 us per 512-column strip of 4096 vectors (three barriers, a read-modify-write
// of the whole trailing strip through LDS and one exposed L2 latency per sub-step); this general body 52 us, the
// fully unrolled one below 40 us.
__device__ __forceinline__ void trsm_reg_body(const TrsmArgs& t, const float* __restrict__ Dinv, int v0, float (*Ybuf)[512]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, l = lane & 15;
  const long xi = (t.xi == 0 && t.xj == 0) ? t.si : t.xi, xj = (t.xi == 0 && t.xj == 0) ? t.sj : t.xj;
  const int n = t.n, nsb = (n + 31) >> 5;
  const int v = v0 + l;
  const bool vok = v < t.nvec;
  const bool coal = t.sj == 1 && !(t.si & 1) && !(reinterpret_cast<uintptr_t>(t.Y) & 7);   // coalesced stores of a solved block
  constexpr int kIt = kStripN / 128;            // 32-column blocks per wave
  f32x4 acc[kIt][2];
#pragma unroll
  for (int it = 0; it < kIt; ++it)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 32 * (w + 4 * it) + 16 * h + 4 * g + e;
        acc[it][h][e] = (vok && c < n) ? t.X[(long)v * xi + (long)c * xj] : 0.0f;
      }
  // q[it][h][mt][e] = Q[32 s + 16 mt + 4 g + e][32 sb + 16 h + l], sb = w + 4 it  (A operands of the trailing update)
  float q[kIt][2][2][4], dv[2][4][2];
  auto load_q = [&](int s, float (&qq)[kIt][2][2][4]) {
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int sb = w + 4 * it;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = 32 * sb + 16 * h + l;
        const bool ok = sb > s && col < n;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) qq[it][h][mt][e] = ok ? t.Q[(long)(32 * s + 16 * mt + 4 * g + e) * t.ldq + col] : 0.0f;
      }
    }
  };
  // dd[h][e][mt] = Dinv_s[16 h + 4 g + e][16 mt + l]
  auto load_d = [&](int s, float (&dd)[2][4][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) dd[h][e][mt] = Dinv[(long)s * 1024 + (16 * h + 4 * g + e) * 32 + 16 * mt + l];
  };
  load_q(0, q);
  if (w == 0) load_d(0, dv);
#pragma unroll
  for (int s = 0; s < kStripN / 32; ++s) {
    if (s >= nsb) continue;                          // uniform; `break` would keep the loop from being unrolled
    float qn[kIt][2][2][4], dn[2][4][2];
    const bool more = s + 1 < nsb;
    if (more) {
      load_q(s + 1, qn);
      if (w == ((s + 1) & 3)) load_d(s + 1, dn);
    }
    if (w == (s & 3)) {                              // owner: Y_s' = Dinv_s' R_s'
      f32x4 y[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            y[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[h][e][mt], acc[s >> 2][h][e], y[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Ybuf[s & 1][(mt * 4 + e) * 64 + lane] = -y[mt][e];
          const int c = 32 * s + 16 * mt + 4 * g + e;
          if (!coal && vok && c < n) t.Y[(long)v * t.si + (long)c * t.sj] = y[mt][e];
        }
    }
    __syncthreads();
    if (coal) {                                        // (vectors as rows: 16 lanes write one 128-byte run, see the full-strip body)
      const int sv = threadIdx.x >> 4, c2 = (threadIdx.x & 15) * 2;
      const int smt = c2 >> 4, sg = (c2 & 15) >> 2, se = c2 & 3;
      const float* src = &Ybuf[s & 1][(smt * 4 + se) * 64 + sg * 16 + sv];
      const int c = 32 * s + c2;
      if (v0 + sv < t.nvec) {
        float* dst = t.Y + (long)(v0 + sv) * t.si + c;
        if (c + 1 < n) *reinterpret_cast<float2*>(dst) = make_float2(-src[0], -src[64]);
        else if (c < n) dst[0] = -src[0];
      }
    }
    if (more) {
      float yn[2][4];                                // -Y_s in B-operand layout
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) yn[mt][e] = Ybuf[s & 1][(mt * 4 + e) * 64 + lane];
#pragma unroll
      for (int it = 0; it < kIt; ++it) {
        const int sb = w + 4 * it;
        if (sb > s && sb < nsb) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int h = 0; h < 2; ++h)
                acc[it][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[it][h][mt][e], yn[mt][e], acc[it][h], 0, 0, 0);
        }
      }
#pragma unroll
      for (int it = 0; it < kIt; ++it)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) q[it][h][mt][e] = qn[it][h][mt][e];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) dv[h][e][mt] = dn[h][e][mt];
    }
  }
}

// The same algorithm for the case every large solve consists of: a FULL strip (n == 512) and 16 valid vectors per
// workgroup.  Everything is unrolled (16 sub-steps) and every global access is unconditional and issued by all four
// waves at the same program points, so the compiler can wait with COUNTED s_waitcnt vmcnt(N): with loads under
// wave-dependent branches (the general body above) it falls back to vmcnt(0) and every sub-step pays the latency of the
// prefetch it has just issued.  Q panels are requested two sub-steps ahead, the inverted diagonal blocks a round ahead.
// Measured at 40 us per strip of 4096 vectors (2.5 us per sub-step; MFMA pipe 16 % busy, waves 54 % of their cycles in
// s_waitcnt).  Tried without gain: 8-byte A-operand loads over interleaved column tiles (41 us), an L2 warm-up pass
// over the strip (54), the round loop as a real loop with retired slots re-reading slot 0 (60), 32 vectors per
// workgroup (50).
// IDENT: the right-hand side is the identity (vector v = unit vector v: the strip's rows of the INVERSE of the 512-block, see
// k_tri_inv512); amax (optional) receives max|Y| of the workgroup's 16 vectors.
template <bool IDENT = false>
__device__ __forceinline__ void trsm_reg_full_body(const TrsmArgs& t, const float* __restrict__ Dinv, int v0, float (*Ybuf)[512],
                                                   float* amax = nullptr) {
  const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long xi = (t.xi == 0 && t.xj == 0) ? t.si : t.xi, xj = (t.xi == 0 && t.xj == 0) ? t.sj : t.xj;
  const int v = v0 + l;
  constexpr int kIt = 4, kSub = 16;
  f32x4 acc[kIt][2];
#pragma unroll
  for (int it = 0; it < kIt; ++it)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[it][h][e] = IDENT ? ((32 * (w + 4 * it) + 16 * h + 4 * g + e == v) ? 1.0f : 0.0f)
                              : t.X[(long)v * xi + (long)(32 * (w + 4 * it) + 16 * h + 4 * g + e) * xj];
  float ymax = 0.0f;
  const float* qlane = t.Q + (long)(4 * g) * t.ldq + 32 * w + l;
  float qb[4][kIt][2][2][4], dvb[2][2][4][2];
  // slot `it` of the round of sub-step s (= block w + 4 ((s >> 2) + it)); slot 0 is loaded even when it is not behind
  // s (w <= s mod 4): the access pattern must not depend on the wave
#ifndef TRSM_DBG
#define TRSM_DBG 0      // what-if switches of tools/micro/x3_gemm_bench.hip (wrong results): 1 = every Q panel load from one line, 2 = no stores of Y
#endif
#define TRSM_LOAD_Q(S, QQ)                                                                              \
  _Pragma("unroll") for (int it = 0; it < kIt; ++it) {                                                  \
    if (((S) >> 2) + it < 4) {                                                                          \
      _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                     \
      _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                  \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                     \
        QQ[it][h][mt][e] = (TRSM_DBG & 1) ? qlane[16 * h + e] : qlane[(long)(32 * (S) + 16 * mt + e) * t.ldq + 128 * (((S) >> 2) + it) + 16 * h]; \
    }                                                                                                   \
  }
#define TRSM_LOAD_D(R, DD)                                                                              \
  _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                         \
  _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                         \
  _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                      \
    DD[h][e][mt] = Dinv[(long)(4 * (R) + w) * 1024 + (16 * h + 4 * g + e) * 32 + 16 * mt + l];
  TRSM_LOAD_D(0, dvb[0]);
  TRSM_LOAD_Q(0, qb[0]);
  TRSM_LOAD_Q(1, qb[1]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r + 1 < 4) { TRSM_LOAD_D(r + 1, dvb[(r + 1) & 1]); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = 4 * r + u;
      if (s + 2 < kSub) { TRSM_LOAD_Q(s + 2, qb[(u + 2) & 3]); }
      if (w == u) {                                    // owner: Y_s' = Dinv_s' R_s'; only MFMAs and LDS writes in here
        f32x4 y[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              y[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dvb[r & 1][h][e][mt], acc[0][h][e], y[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) Ybuf[u & 1][(mt * 4 + e) * 64 + lane] = -y[mt][e];
      }
      __syncthreads();
      float yn[2][4];                                  // -Y_s in B-operand layout, for every wave
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) yn[mt][e] = Ybuf[u & 1][(mt * 4 + e) * 64 + lane];
      // the solved block (16 vectors x 32 columns) goes out from all four waves, two values per lane (same store count on every
      // path).  Vectors as rows (sj == 1): every lane takes two NEIGHBOURING columns of one vector, read back from the LDS slot, so
      // that 16 lanes write one 128-byte run -- with the register layout (lanes along the vectors) every store instruction of this
      // case scattered 64 single floats over 16 rows, which was 17 us of the 51 a strip of 4096 vectors took (tools/micro,
      // TRSM_DBG).  Vectors as columns (si == 1): the register layout is already the coalesced one (16 lanes = 64 bytes).
      if (t.sj == 1 && !(t.si & 1) && !(reinterpret_cast<uintptr_t>(t.Y) & 7)) {
        const int sv = threadIdx.x >> 4, c2 = (threadIdx.x & 15) * 2;                 // vector, first of its two columns
        const int smt = c2 >> 4, sg = (c2 & 15) >> 2, se = c2 & 3;                   // c2 = 16 mt + 4 g + e (e = 0 or 2)
        const float* src = &Ybuf[u & 1][(smt * 4 + se) * 64 + sg * 16 + sv];
        const float2 val = make_float2(-src[0], -src[64]);
        if (IDENT) ymax = amaxf(ymax, amaxf(fabsf(val.x), fabsf(val.y)));
        if (!(TRSM_DBG & 2)) *reinterpret_cast<float2*>(t.Y + (long)(v0 + sv) * t.si + 32 * s + c2) = val;
      } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int mt = k, e = w;                       // register (mt, e = w)
          float val = yn[mt][0];
          if (e == 1) val = yn[mt][1];
          if (e == 2) val = yn[mt][2];
          if (e == 3) val = yn[mt][3];
          if (!(TRSM_DBG & 2)) t.Y[(long)v * t.si + (long)(32 * s + 16 * mt + 4 * g + e) * t.sj] = -val;
        }
      }
      if (s + 1 < kSub) {
        if (w > u) {                                   // slot 0 is behind s only for the waves after the owner
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int h = 0; h < 2; ++h)
                acc[0][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[u][0][h][mt][e], yn[mt][e], acc[0][h], 0, 0, 0);
        }
#pragma unroll
        for (int it = 1; it < kIt; ++it) {
          if (r + it < 4) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                  acc[it][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[u][it][h][mt][e], yn[mt][e], acc[it][h], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int it = 0; it + 1 < kIt; ++it)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[it][h] = acc[it + 1][h];
  }
  if (IDENT && amax) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ymax = amaxf(ymax, __shfl_down(ymax, off, 64));
    if (lane == 0) atomic_amax(amax, ymax);
  }
#undef TRSM_LOAD_Q
#undef TRSM_LOAD_D
}

__global__ __launch_bounds__(kThreads) void k_trsm_ut_reg_full(TrsmArgs t, const float* __restrict__ Dinv) {
  __shared__ float Ybuf[2][512];
  trsm_reg_full_body(t, Dinv, blockIdx.x * 16, Ybuf);
}

// The inverse of every 512 x 512 diagonal block of an upper-triangular Q [n x n], n a multiple of 512, in ONE launch: the strip
// solve Y Q_d = I (row v of Y = row v of Q_d^-1), 32 workgroups of 16 rows per block, from the inverted 32-blocks `dinv` of the
// balance launch.  Replaces k_tri_inv128 and the doubling levels 128 and 256 of tri_inverse (nine launch-bound launches, ~160 us
// inside the update) by one of the strip kernel's ~40 us.  Same arithmetic as the substitution route's strips (fp32 fma chains).
// Writes the blocks of Inv (zeros below the diagonal come out of the substitution itself) and accumulates max|.| into *amax.
__global__ __launch_bounds__(kThreads) void k_tri_inv512(const float* __restrict__ Q, int n, const float* __restrict__ dinv,
                                                         float* __restrict__ Inv, float* amax) {
  __shared__ float Ybuf[2][512];
  const int d = blockIdx.x >> 5, v0 = (blockIdx.x & 31) * 16;
  const long o = (long)(512 * d) * n + 512 * d;
  TrsmArgs t = {Q + o, 512, n, nullptr, Inv + o, 512, (long)n, 1L, 0L, 0L};
  trsm_reg_full_body<true>(t, dinv + (long)(16 * d) * 1024, v0, Ybuf, amax);
}

__global__ __launch_bounds__(kThreads) void k_trsm_ut_reg(TrsmArgs t, const float* __restrict__ Dinv) {
  __shared__ float Ybuf[2][512];
  trsm_reg_body(t, Dinv, blockIdx.x * 16, Ybuf);
}

struct TrsmBatch {
  int count;
  int blk_end[kMaxBatch];      // inclusive prefix sums of 16-vector strips (solve) / of 32-blocks (inversion)
  TrsmArgs t[kMaxBatch];
  const float* dinv[kMaxBatch];
};

__global__ __launch_bounds__(kThreads) void k_trsm_ut_inv_batched(TrsmBatch b, int pitch_max) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  int p = 0;
  while (p + 1 < b.count && (int)blockIdx.x >= b.blk_end[p]) ++p;
  const int blk = blockIdx.x - (p ? b.blk_end[p - 1] : 0);
  const TrsmArgs& t = b.t[p];
  const int pitch = ((t.n + 31) & ~31) + 2;
  trsm_inv_body<16>(t, b.dinv[p], blk * 16, dyn_lds, pitch, reinterpret_cast<float (*)[33]>(dyn_lds + 16 * pitch_max));
}

__global__ __launch_bounds__(kThreads) void k_trsm_ut_reg_batched(TrsmBatch b) {
  __shared__ float Ybuf[2][512];
  int p = 0;
  while (p + 1 < b.count && (int)blockIdx.x >= b.blk_end[p]) ++p;
  const int blk = blockIdx.x - (p ? b.blk_end[p - 1] : 0);
  trsm_reg_body(b.t[p], b.dinv[p], blk * 16, Ybuf);
}


// One stage of each of the two independent chains of the small-layer update in a single launch: the products
// T = dG QrS' -> A = QlS T (:173) and the solves X1 = dX QrS^-1 -> Bt = QlS^-T X1 (:174) meet only at the gradient pair,
// and either stage alone leaves most of the chip idle.  The solve strips come first in the grid (the longer bodies).
struct MixBatch {
  int count, strips;           // layers; total solve strips (blocks below `strips` solve, the rest multiply)
  int blk_end[kMaxLayers];     // inclusive prefix sums of 16-vector strips
  int tile_end[kMaxLayers];    // inclusive prefix sums of 32 x 32 output tiles
  TrsmArgs t[kMaxLayers];
  const float* dinv[kMaxLayers];
  GemmArgs g[kMaxLayers];
};

__global__ __launch_bounds__(kThreads) void k_small_stage_mixed(MixBatch b) {
  __shared__ __attribute__((aligned(16))) GemmLds<32, kSmallK> L;
  __shared__ float Ybuf[2][512];
  if ((int)blockIdx.x < b.strips) {
    int p = 0;
    while (p + 1 < b.count && (int)blockIdx.x >= b.blk_end[p]) ++p;
    const int blk = blockIdx.x - (p ? b.blk_end[p - 1] : 0);
    trsm_reg_body(b.t[p], b.dinv[p], blk * 16, Ybuf);
    return;
  }
  const int id = blockIdx.x - b.strips;
  int p = 0;
  while (p + 1 < b.count && id >= b.tile_end[p]) ++p;
  const int t = id - (p ? b.tile_end[p - 1] : 0);
  const GemmArgs& g = b.g[p];
  const int tn = (g.N + 31) / 32;
  gemm_body_small(g, (t / tn) * 32, (t % tn) * 32, L);
}

// rho = sqrt(max diag Ql / max diag Qr); QlS = Ql / rho; QrS = rho Qr      (psgd.py:166-170)
struct BalanceBatch {
  int count;
  const float* Ql[kMaxBatch]; const float* Qr[kMaxBatch];
  float* QlS[kMaxBatch]; float* QrS[kMaxBatch];
  float* dinv[kMaxBatch];      // inverted 32 x 32 diagonal blocks: QrS's, then QlS's
  float* scal[kMaxBatch];      // the layer's 64 scratch words (max|grad| accumulators ...): zeroed here, one launch ahead
  int M[kMaxBatch], N[kMaxBatch];
};

__device__ __forceinline__ float balance_rho(const float* __restrict__ Ql, const float* __restrict__ Qr, int M, int N,
                                             float (*red)[4]) {
  float ml = -INFINITY, mr = -INFINITY;
  // (eight strided loads in flight per lane: one after the other they were 16 L2 round trips per factor at 4096 -- every block of
  //  the balance launch starts with this; 92 -> 86 us for the launch)
  for (int i0 = threadIdx.x; i0 < M; i0 += 8 * kThreads) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long i = min(i0 + u * kThreads, M - 1); x[u] = Ql[i * M + i]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) ml = nmaxf(ml, x[u]);            // (a clamped index repeats the last diagonal element: harmless for a maximum)
  }
  for (int i0 = threadIdx.x; i0 < N; i0 += 8 * kThreads) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long i = min(i0 + u * kThreads, N - 1); x[u] = Qr[i * N + i]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) mr = nmaxf(mr, x[u]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ml = nmaxf(ml, __shfl_down(ml, off, 64));
    mr = nmaxf(mr, __shfl_down(mr, off, 64));
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = ml; red[1][w] = mr; }
  __syncthreads();
  ml = nmaxf(nmaxf(red[0][0], red[0][1]), nmaxf(red[0][2], red[0][3]));
  mr = nmaxf(nmaxf(red[1][0], red[1][1]), nmaxf(red[1][2], red[1][3]));
  return sqrtf(ml / mr);
}

// part_l / part_r (optional): this block's max|QlS|, max|QrS| (slot `bid` of `nblocks`) for the f16 x 2 planes of the
// factors; the split kernel reduces the slots (thousands of atomics on one address would serialise in L2: measured, they
// turned this 16 us launch into 98 us)
__device__ __forceinline__ void balance_body(const float* __restrict__ Ql, const float* __restrict__ Qr, int M, int N,
                                             float* QlS, float* QrS, float (*red)[4], int bid, int nblocks,
                                             float* part_l = nullptr, float* part_r = nullptr) {
  const float rho = balance_rho(Ql, Qr, M, N, red);
  const long nl = (long)M * M, nr = (long)N * N;
  const long tid = (long)bid * kThreads + threadIdx.x, nth = (long)nblocks * kThreads;
  float ml = 0.0f, mr = 0.0f;
  // 16-byte accesses where the four pointers allow it (4096^2: 104 -> 6x us for 268 MB), element by element otherwise
  const bool v4 = ((reinterpret_cast<uintptr_t>(Ql) | reinterpret_cast<uintptr_t>(Qr) | reinterpret_cast<uintptr_t>(QlS) |
                    reinterpret_cast<uintptr_t>(QrS)) & 15) == 0;
  long dl = 0, dr = 0;
  if (v4) {
    const long nl4 = nl >> 2, nr4 = nr >> 2;
    for (long i = tid; i < nl4; i += nth) {
      const float4 v = reinterpret_cast<const float4*>(Ql)[i];
      const float4 q = make_float4(v.x / rho, v.y / rho, v.z / rho, v.w / rho);
      reinterpret_cast<float4*>(QlS)[i] = q;
      ml = amaxf(amaxf(ml, fabsf(q.x)), amaxf(amaxf(fabsf(q.y), fabsf(q.z)), fabsf(q.w)));
    }
    for (long i = tid; i < nr4; i += nth) {
      const float4 v = reinterpret_cast<const float4*>(Qr)[i];
      const float4 q = make_float4(rho * v.x, rho * v.y, rho * v.z, rho * v.w);
      reinterpret_cast<float4*>(QrS)[i] = q;
      mr = amaxf(amaxf(mr, fabsf(q.x)), amaxf(amaxf(fabsf(q.y), fabsf(q.z)), fabsf(q.w)));
    }
    dl = nl4 << 2; dr = nr4 << 2;
  }
  for (long i = dl + tid; i < nl; i += nth) { const float q = Ql[i] / rho; QlS[i] = q; ml = amaxf(ml, fabsf(q)); }
  for (long i = dr + tid; i < nr; i += nth) { const float q = rho * Qr[i]; QrS[i] = q; mr = amaxf(mr, fabsf(q)); }
  if (part_l) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ml = amaxf(ml, __shfl_down(ml, off, 64));
      mr = amaxf(mr, __shfl_down(mr, off, 64));
    }
    __syncthreads();                                   // (red: balance_rho is done with it)
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ml; red[1][threadIdx.x >> 6] = mr; }
    __syncthreads();
    if (threadIdx.x == 0) {
      part_l[bid] = amaxf(amaxf(red[0][0], red[0][1]), amaxf(red[0][2], red[0][3]));
      part_r[bid] = amaxf(amaxf(red[1][0], red[1][1]), amaxf(red[1][2], red[1][3]));
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_kron_balance(const float* __restrict__ Ql, const float* __restrict__ Qr,
                                                           int M, int N, float* QlS, float* QrS, float* scal) {
  __shared__ float red[2][4];
  if (scal && blockIdx.x == 0 && threadIdx.x < 64) scal[threadIdx.x] = 0.0f;   // max|grad| accumulators of the later stages
  balance_body(Ql, Qr, M, N, QlS, QrS, red, blockIdx.x, gridDim.x);
}

// Row `lane` of the inverse of an upper-triangular 32 x 32 block held in LDS (lanes 0..31 of one wave).
__device__ __forceinline__ void tri_inv32_rows(const float (*Qd)[33], int lane, float* out) {
  float rr[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) rr[j] = (j == lane) ? 1.0f : 0.0f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float y = rr[j] / Qd[j][j];
    rr[j] = y;
#pragma unroll
    for (int j2 = j + 1; j2 < 32; ++j2) rr[j2] = fmaf(-y, Qd[j][j2], rr[j2]);
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) out[j] = rr[j];
}

// Blocks x < kBalInvBlocks of a layer invert the 32 x 32 diagonal blocks of the *balanced* factors (one block per wave;
// the balanced entries are formed exactly as the other blocks store them), the rest write QlS / QrS: the inversion is
// the longer of the two and needs nothing but rho, so it shares the launch instead of waiting behind two GEMM stages.
// The inverted 32 x 32 diagonal blocks of the balanced factors for workgroup `bid` of the inversion part of a balance
// launch: one block per wave, QrS's first, then QlS's (the layout of dinv the solves read).
__device__ __forceinline__ void balance_inv_rho(const float* __restrict__ Ql, const float* __restrict__ Qr, int M, int N,
                                                float* dinv, float (*Qd)[32][33], int bid, float rho);
__device__ __forceinline__ void balance_inv_body(const float* __restrict__ Ql, const float* __restrict__ Qr, int M, int N,
                                                 float* dinv, float (*red)[4], float (*Qd)[32][33], int bid) {
  balance_inv_rho(Ql, Qr, M, N, dinv, Qd, bid, balance_rho(Ql, Qr, M, N, red));
}
__device__ __forceinline__ void balance_inv_rho(const float* __restrict__ Ql, const float* __restrict__ Qr, int M, int N,
                                                float* dinv, float (*Qd)[32][33], int bid, float rho) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nbr = (N + 31) / 32, nbl = (M + 31) / 32;
  const int bi = bid * 4 + w;
  const bool act = bi < nbr + nbl;
  const bool left = bi >= nbr;
  const float* __restrict__ Q = left ? Ql : Qr;
  const int n = left ? M : N, j0 = (left ? bi - nbr : bi) * 32;
  if (act)
    for (int e = lane; e < 1024; e += 64) {
      const int r = e >> 5, c = e & 31;
      const bool in = j0 + r < n && j0 + c < n;
      const float q = Q[(long)min(j0 + r, n - 1) * n + min(j0 + c, n - 1)];
      Qd[w][r][c] = in ? (left ? q / rho : rho * q) : (r == c ? 1.0f : 0.0f);
    }
  __syncthreads();
  if (act && lane < 32) tri_inv32_rows(Qd[w], lane, dinv + (long)bi * 1024 + lane * 32);
}

// Blocks x < kBalInvBlocks of a layer invert the diagonal blocks, the rest write QlS / QrS: the inversion is the longer of
// the two and needs nothing but rho, so it shares the launch instead of waiting behind two GEMM stages.
constexpr int kBalInvBlocks = 8;    // 4 waves each: up to 32 diagonal blocks per layer (M, N <= 512)
__global__ __launch_bounds__(kThreads) void k_kron_balance_batched(BalanceBatch b) {
  __shared__ float red[2][4];
  __shared__ float Qd[4][32][33];
  const int p = blockIdx.y;
  const int M = b.M[p], N = b.N[p];
  if ((int)blockIdx.x >= kBalInvBlocks) {
    if (blockIdx.x == kBalInvBlocks && threadIdx.x < 64) b.scal[p][threadIdx.x] = 0.0f;
    balance_body(b.Ql[p], b.Qr[p], M, N, b.QlS[p], b.QrS[p], red, blockIdx.x - kBalInvBlocks, gridDim.x - kBalInvBlocks);
    return;
  }
  balance_inv_body(b.Ql[p], b.Qr[p], M, N, b.dinv[p], red, Qd, blockIdx.x);
}

// one problem; the first inv_blocks workgroups invert the diagonal blocks (0 = balance only)
__global__ __launch_bounds__(kThreads) void k_kron_balance_inv(const float* __restrict__ Ql, const float* __restrict__ Qr,
                                                               int M, int N, float* QlS, float* QrS, float* scal,
                                                               float* dinv, int inv_blocks, float* part_l, float* part_r,
                                                               float* zero, int nzero) {
  __shared__ float red[2][4];
  __shared__ float Qd[4][32][33];
  if ((int)blockIdx.x >= inv_blocks) {
    if ((int)blockIdx.x == inv_blocks) {               // accumulators of the later stages (max|grad|; maxima of f16 x 2 planes)
      if (scal && threadIdx.x < 64) scal[threadIdx.x] = 0.0f;
      for (int i = threadIdx.x; i < nzero; i += kThreads) zero[i] = 0.0f;
    }
    balance_body(Ql, Qr, M, N, QlS, QrS, red, blockIdx.x - inv_blocks, gridDim.x - inv_blocks, part_l, part_r);
    return;
  }
  balance_inv_body(Ql, Qr, M, N, dinv, red, Qd, blockIdx.x);
}

// ---- round 6: balance + factor planes in ONE pass (the tile-scale route of the large update) -------------------------------------
// Through round 5 the prologue was two sweeps: k_kron_balance_inv wrote QlS / QrS (every one of its 1024 workgroups first reading the
// 8192 diagonal elements for rho: more L2 sectors than its share of the matrices) and left partial maxima, k_split3_two read QlS /
// QrS again for the planes -- 88 + 62 us at 4096^2, and both moved the zero halves below the diagonals.  A plane set only needs a
// scale per 128 x 128 tile (tile scales), and a tile's maximum is known to the workgroup that holds the tile, so: k_kron_rho (rho's
// partial maxima, and the zeroing of the later stages' accumulators), then one workgroup per UPPER tile -- read, balance, write the
// fp32 tile, the tile's exponent, both plane forms.  Tiles below the diagonal are not read; of QlS / QrS only those inside the diagonal
// 512-blocks are written (zeros: the inversion reads whole diagonal blocks), the factor updates write zeros there (P3Args::lower_zero).
// (k_kron_rho: the 8192 diagonal elements sit 16 KiB apart, one page each -- ONE workgroup reading them all took 25 us, mostly address
//  translation; 64 workgroups leave 2 x 64 partial maxima and every consumer wave reduces those: 25 -> ~6 us)
constexpr int kRhoBlocks = 64;
__global__ __launch_bounds__(kThreads) void k_kron_rho(const float* __restrict__ Ql, const float* __restrict__ Qr, int M, int N,
                                                       float* __restrict__ part, float* scal, float* zero, int nzero,
                                                       unsigned* zero2, int nzero2) {      // (zero2: the gradient grid's arrival counters)
  __shared__ float red[2][4];
  const int b = blockIdx.x;
  if (b == 0) {
    if (scal && threadIdx.x < 64) scal[threadIdx.x] = 0.0f;       // accumulators of the later stages (max|grad|, ...)
    for (int i = threadIdx.x; i < nzero; i += kThreads) zero[i] = 0.0f;
    for (int i = threadIdx.x; i < nzero2; i += kThreads) zero2[i] = 0u;
  }
  const int pl = (M + kRhoBlocks - 1) / kRhoBlocks, pr = (N + kRhoBlocks - 1) / kRhoBlocks;
  float ml = -INFINITY, mr = -INFINITY;
  for (int i = b * pl + threadIdx.x; i < min(M, (b + 1) * pl); i += kThreads) ml = nmaxf(ml, Ql[(long)i * M + i]);
  for (int i = b * pr + threadIdx.x; i < min(N, (b + 1) * pr); i += kThreads) mr = nmaxf(mr, Qr[(long)i * N + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ml = nmaxf(ml, __shfl_down(ml, off, 64));
    mr = nmaxf(mr, __shfl_down(mr, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ml; red[1][threadIdx.x >> 6] = mr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[b] = nmaxf(nmaxf(red[0][0], red[0][1]), nmaxf(red[0][2], red[0][3]));
    part[kRhoBlocks + b] = nmaxf(nmaxf(red[1][0], red[1][1]), nmaxf(red[1][2], red[1][3]));
  }
}
__device__ __forceinline__ float rho_of_parts(const float* __restrict__ part) {       // every wave for itself: no LDS, no barrier
  const int lane = threadIdx.x & 63;
  float a = part[lane], b = part[kRhoBlocks + lane];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a = nmaxf(a, __shfl_xor(a, off, 64));
    b = nmaxf(b, __shfl_xor(b, off, 64));
  }
  return sqrtf(a / b);                                             // (balance_rho's expression: the same bits)
}

struct BalSide { const float* Q; int n; float* S; __bf16* Pr; __bf16* Pc; int* te_r; int* te_c; };      // planes: [pad128(n)]^2, row / column form
// one 128 x 128 upper tile (tr <= tc) of one factor
__device__ __forceinline__ void balance_planes_tile(const BalSide f, bool left, float rho, int tr, int tc, float (*S)[129], float* red) {
  const int tid = threadIdx.x, n = f.n, r0 = tr * 128, c0 = tc * 128;
  const long np = (long)((n + 127) & ~127), ts = np * 32, ps = np * np;
  const int rl = tid >> 5, cl = (tid & 31) * 4;                    // this lane: rows rl + 8 i, columns cl .. cl + 3
  const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(f.Q) | reinterpret_cast<uintptr_t>(f.S)) & 15) == 0;      // (S may be null)
  float4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = r0 + rl + 8 * i, col = c0 + cl;
    if (vec && row < n && col + 3 < n) v[i] = *reinterpret_cast<const float4*>(f.Q + (long)row * n + col);
    else {
      const long rr = min(row, n - 1);
      v[i].x = f.Q[rr * n + min(col, n - 1)]; v[i].y = f.Q[rr * n + min(col + 1, n - 1)];
      v[i].z = f.Q[rr * n + min(col + 2, n - 1)]; v[i].w = f.Q[rr * n + min(col + 3, n - 1)];
    }
  }
  float vmax = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = r0 + rl + 8 * i, col = c0 + cl;
    float4 q = left ? make_float4(v[i].x / rho, v[i].y / rho, v[i].z / rho, v[i].w / rho)
                    : make_float4(rho * v[i].x, rho * v[i].y, rho * v[i].z, rho * v[i].w);
    if (row < n && f.S) {                                          // the fp32 tile: what was read, balanced (a diagonal tile keeps its lower part)
      if (vec && col + 3 < n) *reinterpret_cast<float4*>(f.S + (long)row * n + col) = q;
      else {
        if (col < n) f.S[(long)row * n + col] = q.x;
        if (col + 1 < n) f.S[(long)row * n + col + 1] = q.y;
        if (col + 2 < n) f.S[(long)row * n + col + 2] = q.z;
        if (col + 3 < n) f.S[(long)row * n + col + 3] = q.w;
      }
    }
    // the planes hold the UPPER part only, zeros in the pads
    q.x = (row < n && col < n && col >= row) ? q.x : 0.0f;
    q.y = (row < n && col + 1 < n && col + 1 >= row) ? q.y : 0.0f;
    q.z = (row < n && col + 2 < n && col + 2 >= row) ? q.z : 0.0f;
    q.w = (row < n && col + 3 < n && col + 3 >= row) ? q.w : 0.0f;
    v[i] = q;
    vmax = amaxf(amaxf(vmax, fabsf(q.x)), amaxf(amaxf(fabsf(q.y), fabsf(q.z)), fabsf(q.w)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = vmax;
  __syncthreads();
  vmax = amaxf(amaxf(red[0], red[1]), amaxf(red[2], red[3]));
  float sc = plane_scale_of_bound(vmax);
  {
    const int e0 = p3_exp_of_scale(sc), eq = e0 - (((e0 % kTeQuant) + kTeQuant) % kTeQuant);     // (as the products' epilogue quantises)
    sc = ldexpf(1.0f, max(eq, -126));
  }
  if (tid == 0) {
    const int ex = (__float_as_uint(vmax) == 0u) ? kTeAny : p3_exp_of_scale(sc);
    if (f.te_r) f.te_r[tr * kTeLd + tc] = ex;
    if (f.te_c) f.te_c[tc * kTeLd + tr] = ex;
  }
  // row form straight from the registers (x = row, k = column: a lane's four columns are four consecutive k)
#pragma unroll
  for (int i = 0; i < 16 && f.Pr; ++i) {
    unsigned q0[2], q1[2];
    split2h_pair(v[i].x * sc, v[i].y * sc, q0);
    split2h_pair(v[i].z * sc, v[i].w * sc, q1);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      *reinterpret_cast<uint2*>(f.Pr + pl * ps + p3_index(ts, r0 + rl + 8 * i, c0 + cl)) = make_uint2(q0[pl], q1[pl]);
  }
  // column form (x = column, k = row) through LDS, 64 rows at a time
  if (!f.Pc) return;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float* d = &S[rl + 8 * i][cl];
      d[0] = v[8 * h + i].x; d[1] = v[8 * h + i].y; d[2] = v[8 * h + i].z; d[3] = v[8 * h + i].w;
    }
    __syncthreads();
    // 16 lanes = the 64 rows (k) of one column (x): 128 contiguous bytes per plane
    const int g4 = (tid & 15) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = (tid >> 4) + 16 * j;
      unsigned q0[2], q1[2];
      split2h_pair(S[g4][col] * sc, S[g4 + 1][col] * sc, q0);
      split2h_pair(S[g4 + 2][col] * sc, S[g4 + 3][col] * sc, q1);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        *reinterpret_cast<uint2*>(f.Pc + pl * ps + p3_index(ts, c0 + col, r0 + 64 * h + g4)) = make_uint2(q0[pl], q1[pl]);
    }
  }
}

// Planes of a row-major fp32 matrix X [R x C] at TILE scales, one workgroup per 128 x 128 tile, ONE sweep: row form (x = row, k = column)
// and / or column form (x = column, k = row) -- the inputs dX, dG, G of the large update and apply (they were max|X| over the matrix, then
// the split: two sweeps).  Pads are written as zeros.  Pr: [pad128(R) x pad128(C)], Pc: [pad128(C) x pad128(R)] (either may be null).
__global__ __launch_bounds__(kThreads) void k_split_rows_ts(const float* __restrict__ X, int R, int C, __bf16* __restrict__ P, long ts, long ps,
                                                            int* __restrict__ te, __bf16* __restrict__ Pc, long tsc, long psc,
                                                            int* __restrict__ tec) {
  __shared__ float red[4];
  __shared__ __attribute__((aligned(16))) float S[64][129];
  const int tid = threadIdx.x, r0 = blockIdx.y * 128, c0 = blockIdx.x * 128;
  const int rl = tid >> 5, cl = (tid & 31) * 4;
  const bool vec = (C & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
  float4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = r0 + rl + 8 * i, col = c0 + cl;
    if (vec && row < R && col + 3 < C) v[i] = *reinterpret_cast<const float4*>(X + (long)row * C + col);
    else {
      const long rr = min(row, R - 1);
      v[i].x = X[rr * C + min(col, C - 1)]; v[i].y = X[rr * C + min(col + 1, C - 1)];
      v[i].z = X[rr * C + min(col + 2, C - 1)]; v[i].w = X[rr * C + min(col + 3, C - 1)];
    }
  }
  float vmax = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = r0 + rl + 8 * i, col = c0 + cl;
    v[i].x = (row < R && col < C) ? v[i].x : 0.0f;
    v[i].y = (row < R && col + 1 < C) ? v[i].y : 0.0f;
    v[i].z = (row < R && col + 2 < C) ? v[i].z : 0.0f;
    v[i].w = (row < R && col + 3 < C) ? v[i].w : 0.0f;
    vmax = amaxf(amaxf(vmax, fabsf(v[i].x)), amaxf(amaxf(fabsf(v[i].y), fabsf(v[i].z)), fabsf(v[i].w)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vmax = amaxf(vmax, __shfl_down(vmax, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = vmax;
  __syncthreads();
  vmax = amaxf(amaxf(red[0], red[1]), amaxf(red[2], red[3]));
  float sc = plane_scale_of_bound(vmax);
  {
    const int e0 = p3_exp_of_scale(sc), eq = e0 - (((e0 % kTeQuant) + kTeQuant) % kTeQuant);
    sc = ldexpf(1.0f, max(eq, -126));
  }
  if (tid == 0) {
    const int ex = (__float_as_uint(vmax) == 0u) ? kTeAny : p3_exp_of_scale(sc);
    if (te) te[blockIdx.y * kTeLd + blockIdx.x] = ex;
    if (tec) tec[blockIdx.x * kTeLd + blockIdx.y] = ex;
  }
#pragma unroll
  for (int i = 0; i < 16 && P; ++i) {
    unsigned q0[2], q1[2];
    split2h_pair(v[i].x * sc, v[i].y * sc, q0);
    split2h_pair(v[i].z * sc, v[i].w * sc, q1);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      *reinterpret_cast<uint2*>(P + pl * ps + p3_index(ts, r0 + rl + 8 * i, c0 + cl)) = make_uint2(q0[pl], q1[pl]);
  }
  if (!Pc) return;
#pragma unroll
  for (int h = 0; h < 2; ++h) {                                    // (as balance_planes_tile: 64 rows at a time through LDS)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float* d = &S[rl + 8 * i][cl];
      d[0] = v[8 * h + i].x; d[1] = v[8 * h + i].y; d[2] = v[8 * h + i].z; d[3] = v[8 * h + i].w;
    }
    __syncthreads();
    const int g4 = (tid & 15) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = (tid >> 4) + 16 * j;
      unsigned q0[2], q1[2];
      split2h_pair(S[g4][col] * sc, S[g4 + 1][col] * sc, q0);
      split2h_pair(S[g4 + 2][col] * sc, S[g4 + 3][col] * sc, q1);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        *reinterpret_cast<uint2*>(Pc + pl * psc + p3_index(tsc, c0 + col, r0 + 64 * h + g4)) = make_uint2(q0[pl], q1[pl]);
    }
  }
}

// grid: [inv_blocks: the inverted 32-blocks] [tl: upper tiles of Ql] [tr: of Qr] [zl, zr: the lower tiles inside the diagonal 512-blocks]
__global__ __launch_bounds__(kThreads) void k_kron_balance_planes(BalSide L, BalSide R, const float* __restrict__ part, float* dinv,
                                                                  int inv_blocks, int tl, int tr, int zl) {
  __shared__ float red[2][4];
  __shared__ __attribute__((aligned(16))) float S[64][129];      // (pitch 129: the transposed reads of 16 rows x 4 columns spread over all banks)
  int b = blockIdx.x;
  const float rho = part ? rho_of_parts(part) : 1.0f;             // (no partial maxima: planes of the factors as they are -- the apply)
  if (b < inv_blocks) {
    balance_inv_rho(L.Q, R.Q, L.n, R.n, dinv, reinterpret_cast<float(*)[32][33]>(&S[0][0]), b, rho);      // (4 x 32 x 33 floats fit S)
    return;
  }
  b -= inv_blocks;
  if (b < tl + tr) {
    const bool left = b < tl;
    const BalSide f = left ? L : R;
    int r, c;
    upper_tile(left ? b : b - tl, (f.n + 127) / 128, r, c);
    balance_planes_tile(f, left, rho, r, c, S, red[0]);
    return;
  }
  b -= tl + tr;
  const bool left = b < zl;
  const BalSide f = left ? L : R;
  if (!left) b -= zl;
  // lower 128-tile number (b % 6) of diagonal 512-block (b / 6): (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
  const int blk = b / 6, t = b % 6;
  const int tr_ = t < 1 ? 1 : t < 3 ? 2 : 3, tc_ = t < 1 ? 0 : t < 3 ? t - 1 : t - 3;
  const int r0 = blk * 512 + tr_ * 128, c0 = blk * 512 + tc_ * 128;
  for (int e = threadIdx.x; e < 128 * 128; e += kThreads) {
    const int row = r0 + (e >> 7), col = c0 + (e & 127);
    if (row < f.n && col < f.n) f.S[(long)row * f.n + col] = 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------
// Sparse Kronecker factors (psgd.py:198-391): a *normalization* factor ql [2,M] is the matrix
// diag(ql[0]) with last column ql[1] (ql[1][M-1] unused, kept 0), a *scaling* factor qr [1,N] is
// diag(qr).  Their halves of the update/apply are elementwise / reduction kernels; the dense half
// of a mixed format reuses the MFMA GEMM and the triangular solve above.  Data matrices come as
// strided views (element (m,n) at p[m*rs + n*cs]) so that the mirrored formats of the dispatcher
// (psgd.py:86,102,104: transposed data) need no copies.
struct MatView { const float* p; long rs, cs; };

// Y[m,n] = (q0[m] X[m,n] + q1[m] X[M-1,n]) * c(n),  c = 1 | colv[n] | colv[n]^2     (Ql X, psgd.py:218-219)
__global__ __launch_bounds__(kThreads) void k_norm_left(MatView X, const float* __restrict__ q0,
                                                        const float* __restrict__ q1, int M, int N,
                                                        const float* __restrict__ colv, int colsq, float* Y) {
  const long tot = (long)M * N;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < tot; e += (long)gridDim.x * kThreads) {
    const int m = (int)(e / N), n = (int)(e % N);
    float v = q0[m] * X.p[m * X.rs + n * X.cs] + q1[m] * X.p[(long)(M - 1) * X.rs + n * X.cs];
    if (colv) { const float c = colv[n]; v *= colsq ? c * c : c; }
    Y[e] = v;
  }
}

// out[n] = sum_i w_i Z[i,n];  mode 0: w_i = q1[i] / (q0[i] q0[M-1]) (psgd.py:232);  mode 1: w_i = q1[i] (psgd.py:265);
// mode 2: out[n] = sum_i Z[i,n]^2 - Z2[i,n]^2 (psgd.py:304).
// Grid (column blocks of 64, row blocks): a block sums rows [r0, r1) of its 64 columns, four waves taking every fourth
// row with eight independent loads in flight, and writes part[rb][n]; k_col_reduce_fin adds the row blocks in order.
// (The first version gave a column block ALL rows: an embedding-shaped operand, 30000 x 1000, ran on 16 workgroups, each
// a chain of 7500 dependent loads -- 3 ms for 120 MB.)
constexpr int kColRedRowBlocks = 128;
__global__ __launch_bounds__(kThreads) void k_col_reduce(MatView Z, MatView Z2, const float* __restrict__ q0,
                                                         const float* __restrict__ q1, int M, int N, int mode,
                                                         int rows_per_block, float* part) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float acc = 0.0f;
  if (n < N) {
    const float qlast = (mode == 0) ? q0[M - 1] : 1.0f;
    const float* zp = Z.p + (long)n * Z.cs;
    const float* zp2 = Z2.p + (long)n * Z2.cs;
    for (int i0 = r0 + ty; i0 < r1; i0 += 32) {
      float z[8], z2[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                     // clamped rows: unconditional loads, masked below
        const int i = min(i0 + 4 * u, r1 - 1);
        z[u] = zp[(long)i * Z.rs];
        z2[u] = (mode == 2) ? zp2[(long)i * Z2.rs] : 0.0f;
        w[u] = (mode == 0) ? q1[i] / (q0[i] * qlast) : (mode == 1 ? q1[i] : 0.0f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float t = (mode == 2) ? z[u] * z[u] - z2[u] * z2[u] : w[u] * z[u];
        acc += (i0 + 4 * u < r1) ? t : 0.0f;
      }
    }
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && n < N) part[(long)blockIdx.y * N + n] = ((red[0][tx] + red[1][tx]) + red[2][tx]) + red[3][tx];
}

__global__ __launch_bounds__(kThreads) void k_col_reduce_fin(const float* __restrict__ part, int RB, int N, float* out) {
  const int n = blockIdx.x * kThreads + threadIdx.x;
  if (n >= N) return;
  float s = 0.0f;
  for (int b0 = 0; b0 < RB; b0 += 8) {                  // eight loads in flight, added in row-block order
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = part[(long)min(b0 + u, RB - 1) * N + n];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (b0 + u < RB) ? x[u] : 0.0f;
  }
  out[n] = s;
}

// Y[m,n] = (X[m,n] (1/q0[m]) - [m == M-1] s[n]) * (colv ? 1/colv[n] : 1)            (Ql^-T X, psgd.py:230-232,356)
__global__ __launch_bounds__(kThreads) void k_norm_left_invT(MatView X, const float* __restrict__ q0,
                                                             const float* __restrict__ sv, int M, int N,
                                                             const float* __restrict__ colv, float* Y) {
  const long tot = (long)M * N;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < tot; e += (long)gridDim.x * kThreads) {
    const int m = (int)(e / N), n = (int)(e % N);
    float v = (1.0f / q0[m]) * X.p[m * X.rs + n * X.cs];
    if (m == M - 1) v -= sv[n];
    if (colv) v *= 1.0f / colv[n];
    Y[e] = v;
  }
}

// Y[m,n] = q0[m] Z[m,n] + [m == M-1] t[n]                                           (Ql^T Z, psgd.py:266-268)
__global__ __launch_bounds__(kThreads) void k_norm_leftT(const float* __restrict__ Z, const float* __restrict__ q0,
                                                         const float* __restrict__ tv, int M, int N, float* Y) {
  const long tot = (long)M * N;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < tot; e += (long)gridDim.x * kThreads) {
    const int m = (int)(e / N), n = (int)(e % N);
    float v = q0[m] * Z[e];
    if (m == M - 1) v += tv[n];
    Y[e] = v;
  }
}

// Y[m,n] *= 1/colv[n]                                                               (psgd.py:299)
__global__ __launch_bounds__(kThreads) void k_col_inv_scale(float* Y, const float* __restrict__ colv, int M, int N) {
  const long tot = (long)M * N;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < tot; e += (long)gridDim.x * kThreads)
    Y[e] *= 1.0f / colv[(int)(e % N)];
}

// gd[m] = sum_n A[m,n]^2 - Bt[m,n]^2 ; gb[m] = sum_n A[m,n] A[M-1,n] - Bt[m,n] Bt[M-1,n] (0 for m = M-1)   (psgd.py:235-237)
// one wave per row
__global__ __launch_bounds__(kThreads) void k_row_stats(const float* __restrict__ A, const float* __restrict__ Bt, int M,
                                                        int N, float* gd, float* gb) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float d = 0.0f, b = 0.0f;
  for (int n = lane; n < N; n += 64) {
    const float a = A[(long)m * N + n], t = Bt[(long)m * N + n];
    d += a * a - t * t;
    b += a * A[(long)(M - 1) * N + n] - t * Bt[(long)(M - 1) * N + n];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { d += __shfl_down(d, off, 64); b += __shfl_down(b, off, 64); }
  if (lane == 0) { gd[m] = d; gb[m] = (m == M - 1) ? 0.0f : b; }
}

__device__ __forceinline__ float block_max(float v, float* red) {          // red: one float per wave (<= 16 waves)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = nmaxf(v, __shfl_down(v, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = nmaxf(r, red[w]);
  __syncthreads();
  return r;
}
constexpr int kFinThreads = 1024;      // the one-block kernels below: vectors of up to tens of thousands of entries

// new_ql0 = ql0 - step1 gd ql0 ; new_ql1 = ql1 - step1 (gd ql1 + ql0[M-1] gb),
// step1 = step / (max(max|gd|, max|gb|) + tiny)                                      (psgd.py:239-241); one block
__global__ __launch_bounds__(kFinThreads) void k_norm_finalize(const float* __restrict__ ql, const float* __restrict__ gd,
                                                               const float* __restrict__ gb, int M, float step, float tiny,
                                                               float* qlOut) {
  __shared__ float red[16];
  float v = 0.0f;
  for (int i = threadIdx.x; i < M; i += blockDim.x) v = nmaxf(v, nmaxf(fabsf(gd[i]), fabsf(gb[i])));
  const float step1 = step / (block_max(v, red) + tiny);
  const float qlast = ql[M - 1];
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    qlOut[i] = ql[i] - step1 * gd[i] * ql[i];
    qlOut[M + i] = ql[M + i] - step1 * (gd[i] * ql[M + i] + qlast * gb[i]);
  }
}

// new_qr = qr - step2 g2 qr, step2 = step / (max|g2| + tiny)                          (psgd.py:305-307); one block
__global__ __launch_bounds__(kFinThreads) void k_scale_finalize(const float* __restrict__ qr, const float* __restrict__ g2,
                                                                int N, float step, float tiny, float* qrOut) {
  __shared__ float red[16];
  float v = 0.0f;
  for (int i = threadIdx.x; i < N; i += blockDim.x) v = nmaxf(v, fabsf(g2[i]));
  const float step2 = step / (block_max(v, red) + tiny);
  for (int i = threadIdx.x; i < N; i += blockDim.x) qrOut[i] = qr[i] - step2 * g2[i] * qr[i];
}

// rho = sqrt(max L / max R) over the "diagonals" of the two factors (a dense factor: stride n+1; a
// normalization factor: its first row; a scaling factor: itself); Lout = L / rho, Rout = rho R.
// (psgd.py:211-215, 288-292, 342-346)
__global__ __launch_bounds__(kFinThreads) void k_balance_generic(const float* __restrict__ L, long l_stride, int l_cnt,
                                                                 long l_tot, const float* __restrict__ R, long r_stride,
                                                                 int r_cnt, long r_tot, float* Lout, float* Rout) {
  __shared__ float red[16];
  float ml = -INFINITY, mr = -INFINITY;
  for (int i = threadIdx.x; i < l_cnt; i += blockDim.x) ml = nmaxf(ml, L[i * l_stride]);
  for (int i = threadIdx.x; i < r_cnt; i += blockDim.x) mr = nmaxf(mr, R[i * r_stride]);
  ml = block_max(ml, red);
  mr = block_max(mr, red);
  const float rho = sqrtf(ml / mr);
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  for (long i = tid; i < l_tot; i += nth) Lout[i] = L[i] / rho;
  for (long i = tid; i < r_tot; i += nth) Rout[i] = rho * R[i];
}

// ------------------------------------------------------------- host side ----
static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

// tile-exponent tables (KronWs::te), one per transient plane buffer
enum { kTeU0 = 0, kTeU1, kTeU2, kTeU3, kTeY0, kTeY1, kTeY2, kTeG1, kTeG2, kTeIcL, kTeIcR, kTeTpL, kTeTpR, kTeDXp, kTeX1p, kTeLr, kTeLc, kTeRr, kTeRc,
       kTePP, kTeF1, kTeF2,   // (the PREPARED state of the apply: survive update calls like the planes they describe)
       kTeSlots = 24 };       // (kTeLr .. kTeRc: the factors' planes when one sweep made them, k_kron_balance_planes)
struct KronWs {
  float *scal, *QlS, *QrS, *T, *A, *X1, *Bt, *g1, *g2, *dinv, *Pl, *Pr;
  __bf16 *PP, *F1, *F2, *Y0, *Y1, *Y2;     // operand planes of the large apply (kron_planes): Gram, factor, its transpose; 3 transients
  __bf16 *Lr, *Lc, *Rr, *Rc, *G1, *G2, *U0, *U1, *U2, *U3;   // ... of the large update: balanced factors (both forms), gradients, 4 transients
  float* split_scratch; unsigned* split_cnt;                 // K-split tail of the gradient grid (k_gemm_p3_grad)
  float* sk_scratch; unsigned* sk_cnt;                       // split-K of products with few output tiles (launch_p3_auto)
  __bf16* S0;                                                // planes of one [max(M, N) x 2048] group of the solves
  PlaneMeta* pmeta;                                          // f16 x 2 planes: scales and maxima (kPm* slots)
  int* te;                                                   // ... tile-exponent tables of the transient plane sets (kTe* slots of kTeTable ints)
  float* pm_part;                                            // ... partial maxima: 4 arrays of kPmPartMax (main stream, side stream, QlS, QrS)
  // the solves through explicit inverses (kron_inv_route): column-form planes of the two inverses, planes and fp32 of the
  // levels' intermediate A^-1 B, planes of dX and of X1'
  __bf16 *IcL, *IcR, *TpL, *TpR, *DXp, *X1p;
  float *TfL, *TfR;
  int64_t total;
  bool factor_ts = false;      // (set by the update) the factors' planes carry TILE scales (kTeLr .. kTeRc), made by k_kron_balance_planes
};

constexpr int kSkMaxTiles = 256, kSkItems = 512;         // split-K of few-tile products: at most 512 partial tiles in flight
constexpr int kGradSplitMax = 256, kGradChunks = 8;      // K-split tail of the gradient grid: at most 256 tiles in 8 chunks each
// Large applies run on pre-split operand planes (k_gemm_p3).  A pure function of the shape: the workspace layout follows it.
// Which shapes run on operand planes (pure functions of the shape: the workspace layout follows them).  Measured
// (tools/kron_planes_min_sweep.py, against the in-GEMM split with its x-edge blocks already on the vector fetch modes):
// the planes kernels have no edge or ragged-K path at all (zero-padded tiles) and split an operand once, so they win on
// every shape with >= 64 output tiles (1000^2 apply 0.24 -> 0.16 ms, 520 x 3000 0.45 -> 0.30, 300 x 4000 0.52 -> 0.37,
// 64 x 8192 with half of its tile row padding), below that whenever a dimension is >= 600 and not both are tile multiples
// (500 x 1700 apply 0.35 -> 0.19 ms, 700^2 0.19 -> 0.11), and for every update from ~384 on (500^2 0.47 -> 0.29 ms,
// 512^2 0.31 -> 0.25).  The apply of aligned 512..896 squares stays on the exact 64-tile kernels (0.06-0.13 against
// 0.09-0.14 ms), 300^2 and smaller on the small-tile kernels (apply 0.03 against 0.06 ms).
static inline int kron_planes_old() {      // PSGD_KRON_PLANES_OLD=1: the round's first rule (M, N >= 1024), for A/B runs
  static int v = -1;
  if (v < 0) { const char* e = getenv("PSGD_KRON_PLANES_OLD"); v = (e && e[0] == '1') ? 1 : 0; }
  return v;
}
static inline long kron_t128(int M, int N) { return (long)((M + 127) / 128) * ((N + 127) / 128); }
static inline bool kron_planes_apply(int M, int N) {
  if (kron_planes_old()) return M >= 1024 && N >= 1024;
  const long t = kron_t128(M, N);
  static const int ignore_aligned = getenv("PSGD_KRON_PLANES_ALIGNED") ? 1 : 0;       // (A/B runs)
  const bool aligned = !ignore_aligned && M % 128 == 0 && N % 128 == 0;
  const int mx = M > N ? M : N;
  return t >= 64 || (mx >= 600 && t >= 12 && (!aligned || mx >= 2048));    // (aligned and long: 384 x 2048 0.24 -> 0.21 ms)
}
static inline bool kron_planes(int M, int N) {          // the update, and the workspace
  if (kron_planes_old()) return M >= 1024 && N >= 1024;
  return kron_planes_apply(M, N) || ((M > N ? M : N) >= 384 && kron_t128(M, N) >= 9);
}
static inline int pad128(int x) { return (x + 127) & ~127; }
// The two triangular solves of the large fp32 update through explicit inverses (tri_inverse): (rounds 3-5) both factors at least 2048 -- below,
// a solve is a few strips and the inversion's chain of launches costs more than it saves (tools/trsm_inv_ab.py: 4096^2 3.35 -> 2.94
// ms, 2944^2 1.89 -> 1.70, 2048 x 4096 1.83 -> 1.70, 6144^2 9.35 -> 8.33; at the end of the round, `mid`: 2048^2 0.895 -> 0.866,
// 1792^2 0.789 -> 0.789, 1536^2 0.60 -> 0.71, 1024^2 0.35 -> 0.51, 2048 x 1024 0.62 -> 0.76).  A pure function of the shape (workspace).
// Round 6: the LARGER factor decides -- a 1024 x 4096 layer (an MLP weight of a transformer) solved its 4096-side in eight strips with
// seven trailing products between them, 744 of the update's 1244 us on one launch-bound chain; through the inverses (the small side's is
// one level) 1024 x 4096 1.142 -> 0.908 ms, 4096 x 1024 1.253 -> 0.983, 1536 x 3072 1.086 -> 0.810, 1280 x 5120 1.757 -> 1.301,
// 1024 x 8192 3.11 -> 2.35, 1024 x 2048 0.561 -> 0.526; with a smaller side below 1024 it stops paying (768 x 3072 0.790 -> 0.729 but
// 512 x 2048 0.497 -> 0.519): the larger side from 2048, the smaller from kInvMinSmall (tools/r06_kron_shapes.py).
// The longer the larger side -- the more strips its solve would take --, the smaller the other side may be: from 2560 on a 512-side joins
// (896 x 3584 1.037 -> 0.833, 768 x 4096 1.145 -> 0.898, 512 x 4096 1.035 -> 0.881, 512 x 8192 2.63 -> 2.02, 768 x 3072 0.841 -> 0.738;
// 640 x 2560 and 512 x 3072 -1..-2 %), from 4096 on a 256-side (256 x 4096 0.983 -> 0.850, 384 x 4096 1.003 -> 0.870).
constexpr int kInvMinN = 2048, kInvMinSmall = 1024;
static inline bool kron_inv_route(int M, int N) {
  const int lo = M < N ? M : N, hi = M < N ? N : M;
  return (lo >= kInvMinSmall && hi >= kInvMinN) || (lo >= 512 && hi >= 2560) || (lo >= 256 && hi >= 4096);
}

static KronWs kron_layout(char* base, int M, int N) {
  KronWs k;
  const int64_t mm = (int64_t)M * M * 4, nn = (int64_t)N * N * 4, mn = (int64_t)M * N * 4;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { float* p = reinterpret_cast<float*>(base + off); off = align256(off + bytes); return p; };
  k.scal = take(256);
  k.QlS = take(mm); k.QrS = take(nn);
  k.T = take(mn); k.A = take(mn); k.X1 = take(mn); k.Bt = take(mn);
  k.g1 = take(mm); k.g2 = take(nn);
  k.dinv = take((int64_t)((M + 31) / 32 + (N + 31) / 32) * 1024 * 4);
  k.Pl = take(mm); k.Pr = take(nn);          // Grams of the factors (psgd_kron_dd_prepare_f32): survive update calls
  k.PP = k.F1 = k.F2 = k.Y0 = k.Y1 = k.Y2 = nullptr;
  k.Lr = k.Lc = k.Rr = k.Rc = k.G1 = k.G2 = k.U0 = k.U1 = k.U2 = k.U3 = nullptr;
  k.split_scratch = nullptr; k.split_cnt = nullptr; k.S0 = nullptr;
  k.sk_scratch = nullptr; k.sk_cnt = nullptr; k.pmeta = nullptr; k.pm_part = nullptr; k.te = nullptr;
  k.IcL = k.IcR = k.TpL = k.TpR = k.DXp = k.X1p = nullptr; k.TfL = k.TfR = nullptr;
  if (kron_planes(M, N)) {
    const int64_t Mp = pad128(M), Np = pad128(N), small = Mp < Np ? Mp : Np, big = Mp < Np ? Np : Mp;
    auto planes = [&](int64_t elems) { return reinterpret_cast<__bf16*>(take(elems * 6)); };
    k.PP = planes(small * small); k.F1 = planes(big * big); k.F2 = planes(big * big);      // survive update calls too
    k.Y0 = planes(Mp * Np); k.Y1 = planes(Mp * Np); k.Y2 = planes(Mp * Np);
    k.Lr = planes(Mp * Mp); k.Lc = planes(Mp * Mp); k.G1 = planes(Mp * Mp);
    k.Rr = planes(Np * Np); k.Rc = planes(Np * Np); k.G2 = planes(Np * Np);
    k.U0 = planes(Mp * Np); k.U1 = planes(Mp * Np); k.U2 = planes(Mp * Np); k.U3 = planes(Mp * Np);
    {
      // M = N: the K-split tail of the gradient grid; M != N (round 6): EVERY tile of the smaller gradient is split -- its K is the longer
      // side (launch_p3_grad).  Never more tiles than the smaller triangle has.
      const int64_t T1 = small / 128, nsp = T1 * (T1 + 1) / 2 < kGradSplitMax ? T1 * (T1 + 1) / 2 : kGradSplitMax;
      k.split_scratch = take(nsp * kGradChunks * 64 * kThreads * 4);
      k.split_cnt = reinterpret_cast<unsigned*>(take(2 * kGradSplitMax * 4));      // (second half: the solves' split products, launch_p3_solve)
    }
    k.S0 = planes(big * 2048);
    k.pmeta = reinterpret_cast<PlaneMeta*>(take(1024));      // kPmSlots x 16 B
    k.pm_part = take(4 * 2048 * 4);
    k.te = reinterpret_cast<int*>(take((int64_t)kTeSlots * kTeLd * kTeLd * 4));
    if (kron_inv_route(M, N)) {
      k.IcL = planes(Mp * Mp); k.TpL = planes(Mp * Mp); k.TfL = take(mm);
      k.IcR = planes(Np * Np); k.TpR = planes(Np * Np); k.TfR = take(nn);
      k.DXp = planes(Mp * Np); k.X1p = planes(Mp * Np);
    }
    if (kron_t128(M, N) <= kSkMaxTiles) {                  // few output tiles: room for tiles x chunks <= 512 partial tiles
      k.sk_scratch = take((int64_t)kSkItems * 64 * kThreads * 4);
      k.sk_cnt = reinterpret_cast<unsigned*>(take(kSkMaxTiles * 4));
    }
  }
  k.total = off;
  return k;
}

static int g_gemm_x3 = 1;       // tuning key 1: large products on the bf16 matrix cores with a 3-way operand split
static int g_force_gemm = 0;   // 0 auto, 1 always 64-tile kernel, 2 always 128-tile kernel (experiments)
static int g_small_deep = 1;   // tuning key 3: batched 32-tile products on k_gemm_small (0 = gemm_body<32, 64>)

// fetch mode of the split GEMM for one operand given as an (x, k) view (see g2r_x3 / x3_lane_offset: 32-bit lane offsets)
static int x3_host_mode(const float* P, long rs, long cs) {
  if (rs == 1 && cs > 0 && cs < (1L << 23)) return X3_XROW;
  if (cs == 1 && rs > 0 && rs < (1L << 22) && (rs & 3) == 0 && (reinterpret_cast<uintptr_t>(P) & 15) == 0) return X3_KVEC;
  return X3_EDGE;
}
static void x3_host_modes(const GemmArgs& g, int& ma, int& mb) {
  ma = x3_host_mode(g.A, g.a_rs, g.a_cs);
  mb = x3_host_mode(g.B, g.b_cs, g.b_rs);
  if (g.A2 && (x3_host_mode(g.A2, g.a2_rs, g.a2_cs) != ma || x3_host_mode(g.B2, g.b2_cs, g.b2_rs) != mb)) ma = X3_EDGE;
  if (ma == X3_EDGE || mb == X3_EDGE) ma = mb = X3_EDGE;
}

template <bool LITE>
static void launch_x3(const GemmArgs& g, dim3 grid, hipStream_t st) {
  int ma, mb;
  x3_host_modes(g, ma, mb);
  if (ma == X3_KVEC && mb == X3_XROW) hipLaunchKernelGGL((k_gemm_x3<X3_KVEC, X3_XROW, LITE>), grid, dim3(kThreads), 0, st, g);
  else if (ma == X3_KVEC && mb == X3_KVEC) hipLaunchKernelGGL((k_gemm_x3<X3_KVEC, X3_KVEC, LITE>), grid, dim3(kThreads), 0, st, g);
  else if (ma == X3_XROW && mb == X3_XROW) hipLaunchKernelGGL((k_gemm_x3<X3_XROW, X3_XROW, LITE>), grid, dim3(kThreads), 0, st, g);
  else if (ma == X3_XROW && mb == X3_KVEC) hipLaunchKernelGGL((k_gemm_x3<X3_XROW, X3_KVEC, LITE>), grid, dim3(kThreads), 0, st, g);
  else hipLaunchKernelGGL((k_gemm_x3<X3_EDGE, X3_EDGE, LITE>), grid, dim3(kThreads), 0, st, g);
}

static int launch_gemm(const GemmArgs& g, hipStream_t st) {
  // tiles of one launch read A, B while other tiles write C: an output that IS an operand is a race whatever the timing
  if (g.C == g.A || g.C == g.B || (g.A2 && (g.C == g.A2 || g.C == g.B2))) return PSGD_ERR_BAD_ARG;
  // the 128-tile kernel needs enough tiles to fill the chip; small problems keep 64 x 64 tiles
  const long t128 = (long)((g.N + 127) / 128) * ((g.M + 127) / 128);
  const long t64 = (long)((g.N + 63) / 64) * ((g.M + 63) / 64);
  int T = t128 >= 64 ? 128 : (t64 >= 48 ? 64 : 32);   // tiny problems: more, smaller tiles (latency-bound per block)
  if (g_force_gemm == 1) T = 64;
  if (g_force_gemm == 2) T = 128;
  if (g_force_gemm == 3) T = 32;
  dim3 grid((g.N + T - 1) / T, (g.M + T - 1) / T);
  if (T == 128 && g_gemm_x3 && g.lite) launch_x3<true>(g, grid, st);
  else if (T == 128 && g_gemm_x3) launch_x3<false>(g, grid, st);
  else if (T == 128) hipLaunchKernelGGL((k_gemm_f32<128, kBigK>), grid, dim3(kThreads), 0, st, g);
  else if (T == 64) hipLaunchKernelGGL((k_gemm_f32<64, kSmallK>), grid, dim3(kThreads), 0, st, g);
  else if (g_small_deep && g.K > 2 * kSmallK)      // the ring pays off from the third K tile on; shorter products keep the plain body
    hipLaunchKernelGGL(k_gemm_small_one, grid, dim3(kThreads), 0, st, g);
  else hipLaunchKernelGGL((k_gemm_f32<32, kSmallK>), grid, dim3(kThreads), 0, st, g);
  return (int)hipGetLastError();
}

static bool gemm_uses_x3(const GemmArgs& g) {
  const long t128 = (long)((g.N + 127) / 128) * ((g.M + 127) / 128);
  return g_gemm_x3 && (g_force_gemm == 0 ? t128 >= 64 : g_force_gemm == 2);
}

static int launch_gemm_two(const GemmArgs& a, const GemmArgs& b, hipStream_t st) {
  if (!gemm_uses_x3(a) || !gemm_uses_x3(b)) {
    const int rc = launch_gemm(a, st);
    return rc ? rc : launch_gemm(b, st);
  }
  GemmPair p;
  p.g[0] = a; p.g[1] = b;
  p.tx0 = (a.N + 127) / 128; p.tx1 = (b.N + 127) / 128;
  p.tiles0 = p.tx0 * ((a.M + 127) / 128);
  const int tiles1 = p.tx1 * ((b.M + 127) / 128);
  p.tiles1 = tiles1;
  int ma0, mb0, ma1, mb1;
  x3_host_modes(a, ma0, mb0);
  x3_host_modes(b, ma1, mb1);
  const dim3 grid(p.tiles0 + tiles1);
  if (ma0 == X3_KVEC && mb0 == X3_KVEC && ma1 == X3_XROW && mb1 == X3_XROW)              // A A' - Bt Bt' with A'A - Bt'Bt
    hipLaunchKernelGGL((k_gemm_x3_pair<X3_KVEC, X3_KVEC, X3_XROW, X3_XROW>), grid, dim3(kThreads), 0, st, p);
  else if (ma0 == X3_KVEC && mb0 == X3_XROW && ma1 == X3_KVEC && mb1 == X3_XROW)         // the two factor updates
    hipLaunchKernelGGL((k_gemm_x3_pair<X3_KVEC, X3_XROW, X3_KVEC, X3_XROW>), grid, dim3(kThreads), 0, st, p);
  else {
    const int rc = launch_gemm(a, st);
    return rc ? rc : launch_gemm(b, st);
  }
  return (int)hipGetLastError();
}

// C[M,N] = op(A) op(B); ta/tb: operand stored transposed (row-major [K,M] / [N,K])
static GemmArgs gemm_args(const float* A, int lda, bool ta, const float* B, int ldb, bool tb, float* C, int ldc, int M,
                          int N, int K, int kmode = 0) {
  GemmArgs g = {};
  g.A = A; g.a_rs = ta ? 1 : lda; g.a_cs = ta ? lda : 1;
  g.B = B; g.b_rs = tb ? 1 : ldb; g.b_cs = tb ? ldb : 1;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.epi = EPI_STORE;
  g.kmode = kmode;
  return g;
}

// The products of _precond_grad_dense_dense (psgd.py:189-192), split into what depends on the factors only (`pre`,
// np stages: Grams) and what depends on the gradient (`app`, na stages).
//   Large problems keep the reference's association order, with the K ranges implied by the upper-triangular factors:
//     M < N:   ((Ql'Ql) G) Qr' Qr      pre = {Ql'Ql},  app = {(.) G, (.) Qr', (.) Qr}
//     M >= N:  Ql' (Ql (G (Qr'Qr)))    pre = {Qr'Qr},  app = {G (.), Ql (.), Ql' (.)}
//   Small problems (M, N <= 512: every launch is latency-bound, ~7 us per dependent stage) use BOTH Grams:
//     M < N:   ((Ql'Ql) G) (Qr'Qr)     M >= N:  (Ql'Ql) (G (Qr'Qr))
//   -- the reference's order on one side, re-associated on the other (same product; fp32 rounding differs at the 1e-7
//   level, tests hold it to the same 1e-5) -- so a call is three launches instead of four (the two Grams are one
//   batched launch), and two when the caller keeps the prepared Grams (factors unchanged between applies).
constexpr int kSmallKron = 512;
static inline bool kron_small(int M, int N) { return M <= kSmallKron && N <= kSmallKron; }

static void plan_apply(const float* Ql, const float* Qr, const float* G, float* out, int M, int N, const KronWs& k,
                       GemmArgs (&pre)[2], int& np, GemmArgs (&app)[3], int& na) {
  GemmArgs gl = gemm_args(Ql, M, true, Ql, M, false, k.Pl, M, M, M, M, KHI_M | KHI_N);   // Ql'Ql
  GemmArgs gr = gemm_args(Qr, N, true, Qr, N, false, k.Pr, N, N, N, N, KHI_M | KHI_N);   // Qr'Qr
  gl.sym = gr.sym = 1;
  if (kron_small(M, N)) {
    pre[0] = gl; pre[1] = gr; np = 2; na = 2;
    if (M < N) {
      app[0] = gemm_args(k.Pl, M, false, G, N, false, k.T, N, M, N, M);               // (Ql'Ql) G
      app[1] = gemm_args(k.T, N, false, k.Pr, N, false, out, N, M, N, N);             // (.) (Qr'Qr)
    } else {
      app[0] = gemm_args(G, N, false, k.Pr, N, false, k.T, N, M, N, N);               // G (Qr'Qr)
      app[1] = gemm_args(k.Pl, M, false, k.T, N, false, out, N, M, N, M);             // (Ql'Ql) (.)
    }
    return;
  }
  np = 1; na = 3;
  if (M < N) {                                                                    // psgd.py:189-190
    pre[0] = gl;
    app[0] = gemm_args(k.Pl, M, false, G, N, false, k.T, N, M, N, M);                 // (.) G
    app[1] = gemm_args(k.T, N, false, Qr, N, true, k.A, N, M, N, N, KLO_N);           // (.) Qr'
    app[2] = gemm_args(k.A, N, false, Qr, N, false, out, N, M, N, N, KHI_N);          // (.) Qr
  } else {                                                                        // psgd.py:191-192
    pre[0] = gr;
    app[0] = gemm_args(G, N, false, k.Pr, N, false, k.T, N, M, N, N);                 // G (.)
    app[1] = gemm_args(Ql, M, false, k.T, N, false, k.A, N, M, N, M, KLO_M);          // Ql (.)
    app[2] = gemm_args(Ql, M, true, k.A, N, false, out, N, M, N, M, KHI_M);           // Ql' (.)
  }
}

// GEMM stages of _update_precond_dense_dense (psgd.py:173-179): 0,1 before the solves, 2..5 after.
static void plan_update(const float* dG, float* QlOut, float* QrOut, int M, int N, float step, float tiny,
                        const KronWs& k, GemmArgs (&s)[6]) {
  s[0] = gemm_args(dG, N, false, k.QrS, N, true, k.T, N, M, N, N, KLO_N);           // T = dG QrS'       (:173)
  s[1] = gemm_args(k.QlS, M, false, k.T, N, false, k.A, N, M, N, M, KLO_M);         // A = QlS T
  s[2] = gemm_args(k.A, N, false, k.A, N, true, k.g1, M, M, M, N);                  // grad1 = triu(A A' - Bt Bt') (:175)
  s[2].A2 = k.Bt; s[2].a2_rs = N; s[2].a2_cs = 1; s[2].B2 = k.Bt; s[2].b2_rs = 1; s[2].b2_cs = N; s[2].K2 = N;
  s[2].epi = EPI_TRIU_MAX; s[2].maxout = k.scal + 0;
  s[3] = gemm_args(k.A, N, true, k.A, N, false, k.g2, N, N, N, M);                  // grad2 = triu(A'A - Bt'Bt) (:176)
  s[3].A2 = k.Bt; s[3].a2_rs = 1; s[3].a2_cs = N; s[3].B2 = k.Bt; s[3].b2_rs = N; s[3].b2_cs = 1; s[3].K2 = M;
  s[3].epi = EPI_TRIU_MAX; s[3].maxout = k.scal + 1;
  s[4] = gemm_args(k.g1, M, false, k.QlS, M, false, QlOut, M, M, M, M, KLO_M | KHI_N);   // QlS - (step1 grad1) QlS (:179)
  s[4].epi = EPI_D_MINUS; s[4].D = k.QlS; s[4].ldd = M; s[4].scale_max = k.scal + 0; s[4].step = step; s[4].tiny = tiny;
  s[5] = gemm_args(k.g2, N, false, k.QrS, N, false, QrOut, N, N, N, N, KLO_M | KHI_N);   // QrS - (step2 grad2) QrS
  s[5].epi = EPI_D_MINUS; s[5].D = k.QrS; s[5].ldd = N; s[5].scale_max = k.scal + 1; s[5].step = step; s[5].tiny = tiny;
}

__global__ __launch_bounds__(kThreads) void k_copy_strided(const float* X, long xi, long xj, float* Y, long si, long sj,
                                                           int nvec, int n) {
  const long tot = (long)nvec * n;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < tot; e += (long)gridDim.x * kThreads) {
    const long i = (si < sj) ? e % nvec : e / n, j = (si < sj) ? e / nvec : e % n;
    Y[i * si + j * sj] = X[i * xi + j * xj];
  }
}

constexpr int kTrsmPlanesK = 2048;      // group width (4 strips) whose update runs on planes when the factor's planes exist
// Which update products of the solves run on planes (the finished strips are split once instead of once per column tile).
// Measured on the f16 x 2 planes (tools/trsm_planes_k_ab.py, profiles/r03_trsm_planes_k_ab.txt): the in-group K = 512 updates
// too, from 64 output tiles on: 4096^2 update 3.24 -> 3.20 ms, 2560^2 1.32 -> 1.25, 3072^2 2.34 -> 2.25, 6144^2 9.05 -> 8.91
// (on the bf16 x 3 planes of round 2 they did not pay: 557 + 7 x 9 us of strip splits against 647 us per solve).
constexpr int g_trsm_planes_min_k = 512;         // (frozen in round 4, was tuning key 13) least K of a solve's update product that runs on planes
constexpr int g_trsm_planes_min_tiles = 64;      // (frozen in round 4, was tuning key 14) ... and its least number of output tiles
constexpr int g_trsm_planes_min_n = 1100;        // (frozen in round 4, was tuning key 15) the solves use planes when M or N exceeds this (tools/trsm_planes_n_ab.py:
                                              // 1300^2 0.629 -> 0.618 ms, 2048^2 0.915 -> 0.898, but 1024^2 0.342 -> 0.351)
// (Round 4 built fused strip kernels for single calls on small layers -- psgd_kron_small.hip, tuning key 21: one launch per call --
//  parity-green and SLOWER than the stage kernels (LeNet5 set: apply 106 vs 67 us, update 551 vs 299; profiles/r04_lenet_fused_ab.txt).
//  Round 5 gave the per-layer call pattern the batched launches instead (kron.layer_batch: 53 / 84 us) and deleted them.)
static int g_stage_mix = 3;     // tuning key 7: bit 0 = the batched small-layer update runs a product stage and a solve stage per launch;
                                // bit 1 = a single update with M, N <= 512 takes the batched route (5 launches instead of 10-13)
static int g_planes = 1;        // tuning key 4: 0 = large applies on k_gemm_x3 (operands split inside the GEMM)

static int g_planes_f16 = 2;    // tuning key 12: planes in the f16 x 2 format: 0 = none (bf16 x 3), 1 = of the large apply, 2 = and of the
                                // large update.  Like key 4 it changes what psgd_kron_dd_prepare_f32 leaves in the workspace: prepare
                                // again after changing it

// planes of a matrix with padded extents x = rows, k = ld (multiples of 128); meta != null <=> f16 x 2 format
static int balance_grid(int M, int N) {                // workgroups of the balance launch that write QlS / QrS
  const long tot = (long)M * M + (long)N * N;
  const long grid = (tot + kThreads - 1) / kThreads;
  return grid > 1024 ? 1024 : (int)grid;
}
// (f16 x 2, planes split from fp32 data: part[0 .. npart) = the partial maxima of |X| that the launch ahead left)
// te: the table of tile exponents of a plane set written with TILE scales (kTeLd x kTeLd ints, indexed [x / 128][k / 128]); null = one
// scale for the matrix (meta)
struct P3Buf { __bf16* p; long rows, ld; PlaneMeta* meta = nullptr; const float* part = nullptr; int npart = 0; int* te = nullptr; };
constexpr int kPmPartMax = 2048;      // partial maxima per array (KronWs::pm_part holds four arrays)
static P3 p3_of(const P3Buf& b) { return P3{b.p, b.rows * 32, b.rows * b.ld, b.meta, b.te}; }
constexpr int kTeTable = kTeLd * kTeLd;        // ints per table
// slots of KronWs::pmeta
enum { kPmPP = 0, kPmF = 1, kPmQs = 2, kPmG = 4, kPmT = 5, kPmA = 6,                       // the apply (0, 1: prepared state)
       kPmL = 8, kPmR = 9, kPmdG = 10, kPmUT = 11, kPmUA = 12, kPmBt = 13, kPmG1 = 14, kPmG2 = 15,   // the update ...
       kPmStrip = 16, kPmStripEnd = 48,                                                     // ... its solves' groups (strips)
       kPmInvR = 48, kPmInvL = 49, kPmdX = 50, kPmX1 = 51, kPmTR = 52, kPmTL = 58,          // ... or the inverse route (6 levels each)
       kPmSlots = 64 };

// f16 x 2: the partial maxima of |X| (n consecutive floats) for the split of `buf` behind, into `part`; zero: see k_absmax
static int launch_absmax(const float* X, long n, P3Buf& buf, float* part, hipStream_t st, float* zero = nullptr, int nzero = 0) {
  long blocks = (n / 4 + kThreads * 4 - 1) / (kThreads * 4);
  if (blocks > kPmPartMax) blocks = kPmPartMax;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_absmax, dim3((unsigned)blocks), dim3(kThreads), 0, st, X, 0L, 1L, n, part, zero, nzero);
  buf.part = part; buf.npart = (int)blocks;
  return (int)hipGetLastError();
}
// ... of the R x C view X(r, c) = X[r * rs + c * cs] with rs == 1 or cs == 1
static int launch_absmax_view(const float* X, long rs, long cs, long R, long C, P3Buf& buf, float* part, hipStream_t st) {
  const long os = cs == 1 ? rs : cs, O = cs == 1 ? R : C, L = cs == 1 ? C : R;
  if (os == L) return launch_absmax(X, O * L, buf, part, st);
  const long blocks = O < kPmPartMax ? O : kPmPartMax;
  hipLaunchKernelGGL(k_absmax, dim3((unsigned)blocks), dim3(kThreads), 0, st, X, os, O, L, part, (float*)nullptr, 0);
  buf.part = part; buf.npart = (int)blocks;
  return (int)hipGetLastError();
}
// grid of a split launch: one workgroup per 64 x 64 tile of the padded view, or -- under a block filter -- per tile of the blocks kept
static dim3 split_grid(long rows, long ld, const SplitOpt& opt) {
  if (!opt.blk) return dim3((unsigned)(ld / 64), (unsigned)(rows / 64));
  const long nblk = ((rows > ld ? rows : ld) + opt.blk - 1) / opt.blk;
  const unsigned t = (unsigned)(opt.blk / 64);
  return opt.off ? dim3(t / 2, t / 2, (unsigned)nblk) : dim3(t, t, (unsigned)nblk);
}
static int launch_split3(const float* X, long rs, long cs, int R, int C, const P3Buf& out, hipStream_t st,
                         SplitOpt opt = SplitOpt{0, 0, 0, 0}) {
  const dim3 grid = split_grid(out.rows, out.ld, opt);
  if (out.meta)
    hipLaunchKernelGGL(k_split3<1>, grid, dim3(kThreads), 0, st, X, rs, cs, R, C, out.p, out.rows * 32, out.rows * out.ld,
                       (__bf16*)nullptr, 0L, 0L, out.meta, out.part, out.npart, opt);
  else
    hipLaunchKernelGGL(k_split3<0>, grid, dim3(kThreads), 0, st, X, rs, cs, R, C, out.p, out.rows * 32, out.rows * out.ld,
                       (__bf16*)nullptr, 0L, 0L, (PlaneMeta*)nullptr, (const float*)nullptr, 0, opt);
  return (int)hipGetLastError();
}
// planes of the view (out) and of its transpose (outT: rows/ld swapped) from one read
// two row-major matrices [Ra x Ca] (row stride Ca) and [Rb x Cb], f16 x 2 row-form planes each (+ the column-form planes when oaT / obT
// are given), in ONE launch
// row-form planes of a contiguous row-major X [R x C] at tile scales (out.te), one launch
static int launch_split_rows_ts(const float* X, int R, int C, const P3Buf& out, hipStream_t st) {
  const P3 o = p3_of(out);
  hipLaunchKernelGGL(k_split_rows_ts, dim3((unsigned)(out.ld / 128), (unsigned)(out.rows / 128)), dim3(kThreads), 0, st, X, R, C, out.p, o.ts, o.ps,
                     out.te, static_cast<__bf16*>(nullptr), 0L, 0L, static_cast<int*>(nullptr));
  return (int)hipGetLastError();
}
// planes of two upper-triangular matrices as they are (no balance), upper tiles only, tile scales: one launch (k_kron_balance_planes)
static int g_fused_prologue = 1;    // tuning key 31: 0 = the round-5 prologues (max|X| launches ahead of the splits, one scale per matrix)
static int launch_factor_planes_ts(const BalSide& L, const BalSide& R, hipStream_t st) {
  const int TL = (L.n + 127) / 128, TR = (R.n + 127) / 128;
  const int tl = TL * (TL + 1) / 2, tr = TR * (TR + 1) / 2;
  hipLaunchKernelGGL(k_kron_balance_planes, dim3(tl + tr), dim3(kThreads), 0, st, L, R, static_cast<const float*>(nullptr),
                     static_cast<float*>(nullptr), 0, tl, tr, 0);
  return (int)hipGetLastError();
}
// ... the column form (x = column, k = row): outT [pad128(C) x pad128(R)]
static int launch_split_cols_ts(const float* X, int R, int C, const P3Buf& outT, hipStream_t st) {
  const P3 o = p3_of(outT);
  hipLaunchKernelGGL(k_split_rows_ts, dim3((unsigned)(outT.rows / 128), (unsigned)(outT.ld / 128)), dim3(kThreads), 0, st, X, R, C,
                     static_cast<__bf16*>(nullptr), 0L, 0L, static_cast<int*>(nullptr), outT.p, o.ts, o.ps, outT.te);
  return (int)hipGetLastError();
}
static int launch_split3_two(const float* Xa, int Ra, int Ca, const P3Buf& oa, const float* Xb, int Rb, int Cb, const P3Buf& ob,
                             hipStream_t st, SplitOpt opt, const P3Buf* oaT = nullptr, const P3Buf* obT = nullptr) {
  if (!oa.meta || !ob.meta || opt.blk) return 1;
  const dim3 ga = split_grid(oa.rows, oa.ld, opt), gb = split_grid(ob.rows, ob.ld, opt);
  SplitJob a = {Xa, (long)Ca, 1L, Ra, Ca, oa.p, oa.rows * 32, oa.rows * oa.ld, oaT ? oaT->p : nullptr, oaT ? oaT->rows * 32 : 0L,
                oaT ? oaT->rows * oaT->ld : 0L, oa.meta, oa.part, oa.npart, opt, (int)ga.x};
  SplitJob b = {Xb, (long)Cb, 1L, Rb, Cb, ob.p, ob.rows * 32, ob.rows * ob.ld, obT ? obT->p : nullptr, obT ? obT->rows * 32 : 0L,
                obT ? obT->rows * obT->ld : 0L, ob.meta, ob.part, ob.npart, opt, (int)gb.x};
  hipLaunchKernelGGL(k_split3_two<1>, dim3(ga.x > gb.x ? ga.x : gb.x, ga.y + gb.y), dim3(kThreads), 0, st, a, b, (int)ga.y);
  return (int)hipGetLastError();
}
static int launch_split3_both(const float* X, long rs, long cs, int R, int C, const P3Buf& out, const P3Buf& outT, hipStream_t st,
                              SplitOpt opt = SplitOpt{0, 0, 0, 0}) {
  const dim3 grid = split_grid(out.rows, out.ld, opt);
  if (out.meta)
    hipLaunchKernelGGL(k_split3<1>, grid, dim3(kThreads), 0, st, X, rs, cs, R, C, out.p, out.rows * 32, out.rows * out.ld,
                       outT.p, outT.rows * 32, outT.rows * outT.ld, out.meta, out.part, out.npart, opt, out.te, outT.te);
  else
    hipLaunchKernelGGL(k_split3<0>, grid, dim3(kThreads), 0, st, X, rs, cs, R, C, out.p, out.rows * 32, out.rows * out.ld,
                       outT.p, outT.rows * 32, outT.rows * outT.ld, (PlaneMeta*)nullptr, (const float*)nullptr, 0, opt);
  return (int)hipGetLastError();
}

static P3Args p3_args(const P3Buf& A, const P3Buf& B, int M, int N, int K, int kmode) {
  P3Args g = {};
  g.A = p3_of(A); g.B = p3_of(B);
  g.fmt = A.meta ? 1 : 0;
  g.e.M = M; g.e.N = N; g.e.K = K; g.e.kmode = kmode; g.e.epi = EPI_STORE;
  return g;
}
// (f16 x 2: the scale of the output from K max|A| max|B|, see PlaneMeta; one meta for both forms of the output)
static void p3_out_meta(P3Args& g, const P3Buf& C) {
  if (!g.fmt) return;
  g.ometa = C.meta; g.oa = g.A.meta; g.ob = g.B.meta; g.okmul = (float)g.e.K;
  g.oa2 = g.ob2 = nullptr; g.okmul2 = 0.0f;
  if (g.e.A2) { g.oa2 = g.A2.meta; g.ob2 = g.B2.meta; g.okmul2 = (float)g.e.K2; }      // (the second pair is named before the outputs)
}
static void p3_out_row(P3Args& g, const P3Buf& C) {
  g.Crow = C.p; g.crow_ts = C.rows * 32; g.crow_ps = C.rows * C.ld;
  if (C.te) g.te_row = C.te; else p3_out_meta(g, C);
}
static void p3_out_col(P3Args& g, const P3Buf& Ct) {
  g.Ccol = Ct.p; g.ccol_ts = Ct.rows * 32; g.ccol_ps = Ct.rows * Ct.ld;
  if (Ct.te) g.te_col = Ct.te; else p3_out_meta(g, Ct);
}
// plane outputs into a corner of a larger plane set (tile scales only): C(x = x0 + row, k = k0 + col) / C'(x = x0 + col, k = k0 + row);
// x0, k0 multiples of 128
static void p3_out_row_at(P3Args& g, const P3Buf& C, long x0, long k0) {
  g.Crow = C.p + (k0 / 32) * (C.rows * 32) + x0 * 32; g.crow_ts = C.rows * 32; g.crow_ps = C.rows * C.ld;
  g.te_row = C.te + (x0 / 128) * kTeLd + k0 / 128;
}
static void p3_out_col_at(P3Args& g, const P3Buf& Ct, long x0, long k0) {
  g.Ccol = Ct.p + (k0 / 32) * (Ct.rows * 32) + x0 * 32; g.ccol_ts = Ct.rows * 32; g.ccol_ps = Ct.rows * Ct.ld;
  g.te_col = Ct.te + (x0 / 128) * kTeLd + k0 / 128;
}
// does this product read or write planes with tile scales (the TS kernels)?
static inline bool p3_uses_te(const P3Args& g) {
  return g.fmt && (g.A.te || g.B.te || (g.e.A2 && (g.A2.te || g.B2.te)) || g.te_row || g.te_col);
}

constexpr int g_sparse_planes = 1;     // (frozen in round 4, was tuning key 20) 0 = every product of the sparse formats on launch_gemm / bf16 x 3 planes (see sparse_gemm)
constexpr int g_force_er = -1;     // (frozen in round 4, was tuning key 19) -1 = the launchers choose the form of the f16 x 2 kernels (P3_EARLY), 0 / 2 = always that one
static inline bool p3_no_early(bool auto_choice) { return g_force_er < 0 ? auto_choice : g_force_er == 0; }
static int p3_block_slots() {           // two resident blocks per CU
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 512;
    slots = 2 * cus;
  }
  return slots;
}
static int launch_p3(const P3Args& g, hipStream_t st) {
  const dim3 grid((g.e.N + 127) / 128, (g.e.M + 127) / 128);
  const bool none = p3_no_early((long)grid.x * grid.y >= p3_block_slots() * 3 / 4);                         // (see P3_EARLY)
  if (!g.fmt) hipLaunchKernelGGL((k_gemm_p3<0, 2>), grid, dim3(kThreads), 0, st, g);
  else if (p3_uses_te(g)) {
    if (none) hipLaunchKernelGGL((k_gemm_p3<1, 0, true>), grid, dim3(kThreads), 0, st, g);
    else hipLaunchKernelGGL((k_gemm_p3<1, 2, true>), grid, dim3(kThreads), 0, st, g);
  } else if (none) hipLaunchKernelGGL((k_gemm_p3<1, 0>), grid, dim3(kThreads), 0, st, g);
  else hipLaunchKernelGGL((k_gemm_p3<1, 2>), grid, dim3(kThreads), 0, st, g);
  return (int)hipGetLastError();
}

// launch_p3, or the K range of every tile dealt to several blocks when the product has few output tiles and a K worth
// splitting (scratch, cnt: KronWs::sk_*; null = never split)
constexpr int g_splitk = 1;        // (frozen in round 4, was tuning key 8) 0 = no split-K of few-tile products
static thread_local bool t_p3_alone = false;      // set by the callers whose products run alone on the device (planes_apply*)
struct P3AloneScope { bool old; P3AloneScope() : old(t_p3_alone) { t_p3_alone = true; } ~P3AloneScope() { t_p3_alone = old; } };
static int launch_p3_auto(const P3Args& g, float* scratch, unsigned* cnt, hipStream_t st) {
  const int tx = (g.e.N + 127) / 128, ty = (g.e.M + 127) / 128, tiles = tx * ty, steps = (g.e.K + 31) / 32;
  // Measured (tools/kron_splitk_ab.py): the partial tiles cost 64 KiB of traffic each way per item, so the split only pays for
  // a tile or two of rows with a long K (64 x 8192 apply 0.78 -> 0.51 ms, 128 x 4096 0.44 -> 0.18, 200 x 3072 0.29 -> 0.18,
  // 2048 x 256 0.20 -> 0.15); at 1000^2, in chunks of 4 K steps, it doubles the time.  Hence: K >= 2048, at most 80 tiles,
  // at least 32 K steps per block.
  // Up to 44 tiles and short K (200 x 1700, 384 x 1300) chunks of 16 K steps still pay (0.175 -> 0.129 ms); from there to 80
  // tiles they do not (1000^2: 0.156 -> 0.195 with chunks of 16), chunks of 32 from K = 2048 on do.
  static const int env_steps = getenv("PSGD_SPLITK_MIN_STEPS") ? atoi(getenv("PSGD_SPLITK_MIN_STEPS")) : 0;     // (env: A/B runs)
  static const int env_chunk = getenv("PSGD_SPLITK_CHUNK") ? atoi(getenv("PSGD_SPLITK_CHUNK")) : 0;
  // (round 3, after the split's hand-off stopped releasing the whole L2: the stricter rule for 45 .. 80 tiles -- 64 steps, chunks of
  //  32 -- is gone: 500 x 1700 apply 0.147 -> 0.129 ms, 900 x 1400 update 0.60 -> 0.53; chunks of 8 steps still lose 20 %)
  const int min_steps = env_steps ? env_steps : 32;
  const int min_chunk = env_chunk ? env_chunk : (steps < 72 ? 16 : 32);   // (128 x 4096: 4 chunks of 32 beat 8 of 16)
  int nchunk = (!g_splitk || !scratch || g.e.A2 || g.e.sym || tiles > 80 || steps < min_steps) ? 1 : kSkItems / tiles;
  if (nchunk > 8) nchunk = 8;
  while (nchunk > 1 && steps / nchunk < min_chunk) --nchunk;
  // Round 6: 81 .. 256 tiles with a long K -- the products of a 1024 x 4096 layer with its 4096-factor are 256 tiles of up to 128 K
  // steps: half the block slots stay empty and the launch takes its longest tile's time (117 us).  Two chunks per tile fill the chip.
  // Only where the product has the chip to itself (the apply: t_p3_alone): in the update the products of :173 run beside the solves' chain,
  // and twice the workgroups there are twice the slots the other chain waits for (1024 x 4096 update 0.904 -> 0.936 ms with it, the apply
  // 0.339 -> 0.301; 512 x 4096 apply 0.306 -> 0.235, 768 x 3072 0.263 -> 0.222, 1024 x 2048 0.226 -> 0.202).
  static const int env_mid = getenv("PSGD_SPLITK_MID") ? atoi(getenv("PSGD_SPLITK_MID")) : 1;      // (env: A/B runs)
  if (env_mid && t_p3_alone && scratch && !g.e.A2 && !g.e.sym && tiles > 80 && tiles <= kSkMaxTiles && steps >= 64) nchunk = 2;
  if (nchunk <= 1) return launch_p3(g, st);
  if (hipMemsetAsync(cnt, 0, (size_t)tiles * 4, st) != hipSuccess) return 1;
  if (g.fmt && p3_uses_te(g)) hipLaunchKernelGGL((k_gemm_p3_splitk_rect<1, true>), dim3(tiles * nchunk), dim3(kThreads), 0, st, g, ty, tx, nchunk, scratch, cnt);
  else if (g.fmt) hipLaunchKernelGGL(k_gemm_p3_splitk_rect<1>, dim3(tiles * nchunk), dim3(kThreads), 0, st, g, ty, tx, nchunk, scratch, cnt);
  else hipLaunchKernelGGL(k_gemm_p3_splitk_rect<0>, dim3(tiles * nchunk), dim3(kThreads), 0, st, g, ty, tx, nchunk, scratch, cnt);
  return (int)hipGetLastError();
}

// The products of the blocked solves of a layer with one long side (1024 x 4096: X1_j = W_j Ri_jj is 128 tiles of up to 64 K steps --
// a quarter of the block slots): two workgroups per tile.  Counters: zeroed by the prologue's first launch, left at zero by every use.
static int g_solve_split = 1;      // tuning key 34
static int launch_p3_solve(const P3Args& g, float* scratch, int slots, unsigned* cnt, hipStream_t st) {
  const int tx = (g.e.N + 127) / 128, ty = (g.e.M + 127) / 128, tiles = tx * ty, steps = (g.e.K + 31) / 32;
  if (!g_solve_split || !scratch || !cnt || !g.fmt || !p3_uses_te(g) || g.e.A2 || g.e.sym || tiles > kGradSplitMax || 2 * tiles > slots || steps < 64)
    return launch_p3(g, st);
  hipLaunchKernelGGL((k_gemm_p3_splitk_rect<1, true>), dim3(tiles * 2), dim3(kThreads), 0, st, g, ty, tx, 2, scratch, cnt);
  return (int)hipGetLastError();
}

// A product whose result the next product reads as planes (`row`: x = row, k = column; `col`: the transposed form; either
// or both).  bf16 x 3: the epilogue writes the planes, the result never exists in fp32.  f16 x 2: the scale of a matrix
// has to come from its ACTUAL maximum -- a bound K max|A| max|B| from the operands' maxima is loose by the conditioning of
// the factors (products like QlS (dG QrS') cancel by orders of magnitude), and every factor of two of looseness is a bit
// of fp16 range lost at the bottom: on factors with cond 1e4 the update's increments came out at 4e-3 .. 1.5e-2 instead of
// 4e-5 (tools/illcond_increment_probe.py).  So the epilogue writes fp32 into `tmp` [M x N, row-major] and accumulates
// max|C| (into the planes' meta, or wherever `amax` points for an epilogue that has its own: EPI_TRIU_MAX), and a split
// launch makes the planes.  Tuning key 16 = 0 keeps the epilogue planes with bound scales (A/B runs).
static int g_planes_exact = 1;
static int g_bg_front = -1;      // tuning key 30: 1 = (both inversions first) the products of :173 on a THIRD stream from the fork point on, beside both inversions,
                                 // instead of behind Ql's on the side stream; 0 = the round-5 order; -1 (default) = 1 when both factors reach 4096.
                                 // Round 6 (profiles/r06_kron_update_order.txt): the full-chip products delay every launch of the two inversion
                                 // chains (they end at 580 us instead of 405), but X1 and Bt then run alone at their isolated times: 4096^2
                                 // 2.250 -> 2.215 ms (four alternations), 6144^2 equal, 2048 x 4096 / 3072^2 +1-2 % (hence the rule).
static inline bool kron_bg_front(int M, int N) { return g_bg_front < 0 ? (M >= 4096 && N >= 4096) : g_bg_front == 1; }
static int g_bg_planes = 1;      // tuning key 32: (shapes below key 30's rule) dX's and dG's planes on the third stream, Ql's inversion from the fork point on
static int g_x0_side = 1;        // tuning key 29: 1 = (both inversions first) dX's planes on the side stream ahead of Ql's inversion
static int g_tile_scales = 1;   // tuning key 28: chained f16 x 2 products write their planes with TILE scales from the epilogue (default); 0 = fp32
                                // out + max|C| + a split launch per chained product (the round-3/4 form, one scale per matrix)
static inline bool kron_tile_scales(int M, int N) { return g_tile_scales && g_planes_exact && g_planes_f16 > 0 && M <= 8192 && N <= 8192; }
static int p3_chain(P3Args& g, float* tmp, const P3Buf* row, const P3Buf* col, const float* amax, float* sk_scratch,
                    unsigned* sk_cnt, hipStream_t st) {
  const P3Buf& any = row ? *row : *col;
  int e;
  if (!g.fmt || !g_planes_exact || any.te) {          // (te: tile scales -- the epilogue writes the planes at each tile's own maximum)
    if (row) p3_out_row(g, *row);
    if (col) p3_out_col(g, *col);
    return launch_p3_auto(g, sk_scratch, sk_cnt, st);
  }
  g.e.C = tmp; g.e.ldc = g.e.N;
  if (!amax) { g.ometa = any.meta; amax = &any.meta->amax; }
  if ((e = launch_p3_auto(g, sk_scratch, sk_cnt, st))) return e;
  P3Buf r = any, c = any;
  if (row) r = *row;
  if (col) c = *col;
  r.part = c.part = amax; r.npart = c.npart = 1;
  if (row && col) return launch_split3_both(tmp, g.e.N, 1, g.e.M, g.e.N, r, c, st);
  if (row) return launch_split3(tmp, g.e.N, 1, g.e.M, g.e.N, r, st);
  return launch_split3(tmp, 1, g.e.N, g.e.N, g.e.M, c, st);                         // (x, k) = C[k][x]
}

// Solve  y[i,:] Q = x[i,:]  (see k_trsm_ut).  `dinv` is scratch for the inverted 32 x 32 diagonal sub-blocks
// (ceil(n/32) * 1024 floats).  n <= 512: one strip kernel.  Larger n: right-looking over 512-wide column blocks,
//   Y <- X;  for each block jb:  Y[:, jb] <- Y[:, jb] Q[jb, jb]^-1  (strip kernel, in place)
//                                Y[:, >jb] -= Y[:, jb] Q[jb, >jb]   (one wide MFMA GEMM, K = 512)
constexpr int kTrsmBlock = kStripN;

static int g_trsm_lds = 0;     // tuning key 2: 1 = the LDS-resident strip kernels (A/B measurements)
constexpr int g_trsm_group = 0;   // (frozen in round 4, was tuning key 5) strips per group of the blocked solve (0 = automatic, 1 = every strip updates all columns to its right)

static int launch_strip(const TrsmArgs& t, const float* dinv, hipStream_t st) {
  if (!g_trsm_lds) {
    if (t.n == kStripN && t.nvec % 16 == 0)
      hipLaunchKernelGGL(k_trsm_ut_reg_full, dim3(t.nvec / 16), dim3(kThreads), 0, st, t, dinv);
    else
      hipLaunchKernelGGL(k_trsm_ut_reg, dim3((t.nvec + 15) / 16), dim3(kThreads), 0, st, t, dinv);
    return (int)hipGetLastError();
  }
  static DeviceOnce attr_set;
  const int pitch = ((t.n + 31) & ~31) + 2;
  if (attr_set.needed()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trsm_ut_inv<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((64 * (kStripN + 2) + 32 * 33) * sizeof(float))) != hipSuccess) return 1;
    attr_set.done();
  }
  // 64-vector workgroups keep a 4096-vector solve on 64 CUs; 16-vector ones cover the chip and shorten the per-sub-step
  // dependency chains (one MFMA row tile per wave)
  if ((t.nvec + 63) / 64 < 192) {
    const size_t lds = (size_t)(16 * pitch + 32 * 33) * sizeof(float);
    hipLaunchKernelGGL(k_trsm_ut_inv<16>, dim3((t.nvec + 15) / 16), dim3(kThreads), lds, st, t, dinv);
  } else {
    const size_t lds = (size_t)(64 * pitch + 32 * 33) * sizeof(float);
    hipLaunchKernelGGL(k_trsm_ut_inv<64>, dim3((t.nvec + 63) / 64), dim3(kThreads), lds, st, t, dinv);
  }
  return (int)hipGetLastError();
}

// Qc / blk (optional): column-form planes of Q (x = column, k = row) and a plane buffer for [nvec x 2048] of Y: the wide
// group updates (K = 2048) then run on planes -- the finished group is split once instead of once per column tile.
static int trsm_ut(const float* Q, int n, const float* X, float* Y, int nvec, long si, long sj, float* dinv,
                   hipStream_t st, long xi = 0, long xj = 0, int lite = 0, const P3Buf* Qc = nullptr, __bf16* blk = nullptr,
                   bool inv_ready = false, PlaneMeta* strip_meta = nullptr, int strip_slots = 0, float* part = nullptr) {
  if (!inv_ready) {                                    // (the update's balance launch has made them already)
    hipLaunchKernelGGL(k_tri_inv32, dim3((n + 31) / 32), dim3(64), 0, st, Q, n, n, dinv);
    if (hipGetLastError() != hipSuccess) return 1;
  }
  if (n <= kStripN) {
    TrsmArgs t = {Q, n, n, X, Y, nvec, si, sj, xi, xj};
    return launch_strip(t, dinv, st);
  }
  // X with the strides of Y is never copied: the first strip and the two products with k0 = 0 (which between them touch
  // every later column first) read it in place of Y
  const bool lazy = xi == 0 && xj == 0;
  if (lazy) {
  } else {
    hipLaunchKernelGGL(k_copy_strided, dim3(1024), dim3(kThreads), 0, st, X, xi, xj, Y, si, sj, nvec, n);
    if (hipGetLastError() != hipSuccess) return 1;
  }
  // Strips are solved in GROUPS of g_trsm_group: inside a group a solved strip updates the group's remaining columns
  // (K = 512), and the finished group updates everything to its right in one product with K = 512 * group -- the
  // 16 K steps of a K = 512 product are dominated by the fixed parts of a block, so the wide products get a longer K.
  auto update = [&](int k0, int kw, int c0, int cw) {                    // Y[:, c0:c0+cw] -= Y[:, k0:k0+kw] Q[k0:k0+kw, c0:c0+cw]
    // (f16 x 2 planes of Q: every finished group needs a zeroed meta slot for its maximum; out of slots -> the fp32 kernel)
    if (Qc && blk && kw >= g_trsm_planes_min_k && kw <= kTrsmPlanesK && kw % 128 == 0 &&
        (long)((cw + 127) / 128) * ((nvec + 127) / 128) >= g_trsm_planes_min_tiles && (!Qc->meta || strip_slots > 0)) {
      P3Buf Yg = {blk, pad128(nvec), kw};
      int e;
      if (Qc->meta) {
        Yg.meta = strip_meta++; --strip_slots;
        if ((e = launch_absmax_view(Y + (long)k0 * sj, si, sj, nvec, kw, Yg, part, st))) return e;
      }
      if ((e = launch_split3(Y + (long)k0 * sj, si, sj, nvec, kw, Yg, st))) return e; // (i, k) = Y[i, k0 + k]
      P3Args g = p3_args(Yg, *Qc, nvec, cw, kw, 0);                                  // (x, k) = Q[k0 + k, c0 + x]
      g.B.p = Qc->p + (long)(k0 / 32) * (Qc->rows * 32) + (long)c0 * 32;
      float* Yr = Y + (long)c0 * sj;
      g.e.C = Yr; g.e.ldc = si; g.e.c_cs = sj;
      g.e.D = (lazy && k0 == 0 ? X : Y) + (long)c0 * sj; g.e.ldd = si;
      g.e.epi = EPI_D_MINUS;
      return launch_p3(g, st);
    }
    GemmArgs g = {};
    g.A = Y + (long)k0 * sj; g.a_rs = si; g.a_cs = sj;
    g.B = Q + (long)k0 * n + c0; g.b_rs = n; g.b_cs = 1;
    float* Yr = Y + (long)c0 * sj;
    g.C = Yr; g.ldc = si; g.c_cs = sj;
    g.D = (lazy && k0 == 0 ? X : Y) + (long)c0 * sj; g.ldd = si;        // in place (or first touch: from X): C = D - A B
    g.M = nvec; g.N = cw; g.K = kw;
    g.epi = EPI_D_MINUS;
    g.lite = lite;
    return launch_gemm(g, st);
  };
  // measured (tools/trsm_group_ab.py): groups of 4 are 4 % of the 4096^2 updates, nothing at 2048 (one group = no wide product)
  const int group = kTrsmBlock * (g_trsm_group > 0 ? g_trsm_group : (n >= 8 * kTrsmBlock ? 4 : 1));
  for (int g0 = 0; g0 < n; g0 += group) {
    const int gend = (n - g0 < group) ? n : g0 + group;
    for (int j0 = g0; j0 < gend; j0 += kTrsmBlock) {
      const int jw = (gend - j0 < kTrsmBlock) ? (gend - j0) : kTrsmBlock;
      float* Yb = Y + (long)j0 * sj;
      TrsmArgs t = {Q + (long)j0 * n + j0, jw, n, (lazy && j0 == 0) ? X : Yb, Yb, nvec, si, sj, 0L, 0L};
      int e = launch_strip(t, dinv + (long)(j0 / 32) * 1024, st);
      if (e) return e;
      if (j0 + jw < gend && (e = update(j0, jw, j0 + jw, gend - j0 - jw))) return e;
    }
    if (gend < n) {
      const int e = update(g0, gend - g0, gend, n - gend);
      if (e) return e;
    }
  }
  return 0;
}

static int launch_gemm_batch(const GemmArgs* g, int count, hipStream_t st) {
  GemmBatch b;
  b.count = count;
  long t64 = 0;
  for (int p = 0; p < count; ++p) t64 += (long)((g[p].N + 63) / 64) * ((g[p].M + 63) / 64);
  const int T = t64 >= 96 ? 64 : 32;
  int tiles = 0;
  for (int p = 0; p < count; ++p) {
    b.g[p] = g[p];
    tiles += ((g[p].N + T - 1) / T) * ((g[p].M + T - 1) / T);
    b.tile_end[p] = tiles;
  }
  if (T == 64) hipLaunchKernelGGL((k_gemm_f32_batched<64>), dim3(tiles), dim3(kThreads), 0, st, b);
  else if (g_small_deep) hipLaunchKernelGGL(k_gemm_small, dim3(tiles), dim3(kThreads), 0, st, b);
  else hipLaunchKernelGGL((k_gemm_f32_batched<32>), dim3(tiles), dim3(kThreads), 0, st, b);
  return (int)hipGetLastError();
}

static int launch_gram_batch(const GemmArgs* g, int count, hipStream_t st) {      // g: Gram problems made by plan_apply
  for (int p0 = 0; p0 < count; p0 += kMaxGrams) {
    GramBatch b;
    b.count = (count - p0 < kMaxGrams) ? count - p0 : kMaxGrams;
    long t64 = 0;
    for (int p = 0; p < b.count; ++p) t64 += (long)((g[p0 + p].N + 63) / 64) * ((g[p0 + p].M + 63) / 64);
    const int T = t64 >= 96 ? 64 : 32;
    int tiles = 0;
    for (int p = 0; p < b.count; ++p) {
      const GemmArgs& a = g[p0 + p];
      b.Q[p] = a.A; b.P[p] = a.C; b.n[p] = a.M;
      tiles += ((a.N + T - 1) / T) * ((a.M + T - 1) / T);
      b.tile_end[p] = tiles;
    }
    if (T == 64) hipLaunchKernelGGL((k_gram_batched<64>), dim3(tiles), dim3(kThreads), 0, st, b);
    else hipLaunchKernelGGL((k_gram_batched<32>), dim3(tiles), dim3(kThreads), 0, st, b);
    if (hipGetLastError() != hipSuccess) return 1;
  }
  return 0;
}

// (Round 4, measured and not kept: the K ranges of the pair's LONG tiles -- the corner tile of a 4096^2 factor is a chain of 128 K
//  steps, ~190 us, in a launch whose whole work is ~135 us of the chip -- dealt to four blocks each (P3Split): 4096^2 update 2.60 ->
//  2.69 ms; the partial tiles' traffic and the arrivals cost more than the chain.  profiles/r04_kron_update_notes.txt)
static int launch_p3_two(const P3Args& a, const P3Args& b, hipStream_t st) {
  P3Pair p;
  p.g[0] = a; p.g[1] = b;
  p.tx0 = (a.e.N + 127) / 128; p.tx1 = (b.e.N + 127) / 128;
  p.tiles0 = p.tx0 * ((a.e.M + 127) / 128);
  const int tiles1 = p.tx1 * ((b.e.M + 127) / 128);
  p.tiles1 = tiles1;
  const dim3 grid(p.tiles0 + tiles1);
  const bool none = p3_no_early((p.tiles0 + tiles1) / 2 >= p3_block_slots() * 3 / 4);
  if (!a.fmt) hipLaunchKernelGGL((k_gemm_p3_pair<0, 2>), grid, dim3(kThreads), 0, st, p);
  else if (p3_uses_te(a) || p3_uses_te(b)) {
    if (none) hipLaunchKernelGGL((k_gemm_p3_pair<1, 0, true>), grid, dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_p3_pair<1, 2, true>), grid, dim3(kThreads), 0, st, p);
  } else if (none) hipLaunchKernelGGL((k_gemm_p3_pair<1, 0>), grid, dim3(kThreads), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_p3_pair<1, 2>), grid, dim3(kThreads), 0, st, p);        // (half of the factor-update tiles are copies)
  return (int)hipGetLastError();
}

static int g_grad_rect = 1;     // tuning key 33: M != N, every tile of the smaller gradient split along its (long) K; 0 = whole tiles (rounds 1-5)
static int g_grad_split = 1;    // tuning key 6: 0 = no K split of the gradient grid's tail.  (While the split's hand-off was a __threadfence() per
                                // block -- a release of the whole L2 -- it cost more than it saved on the f16 x 2 planes and was off for a
                                // while: profiles/r03_grad_grid_isolated.txt.  With the write-through hand-off: 4096^2 0.762 -> 0.665 ms,
                                // 2944^2 0.365 -> 0.328; a long last round is still better left whole: 3072^2 0.376 -> 0.440.)
constexpr int g_grad_order = 1;    // (frozen in round 4, was tuning key 17) tile order of the gradient grid for M = N (see k_gemm_p3_grad; tools/grad_order_ab.py:
                                // 4096^2 update 2.95-2.98 -> 2.87-2.88 ms, 6144^2 8.5 -> 8.3; patches of 4 x 4 tiles (2) are no better:
                                // L2 locality is not what bounds this grid; 2048 x 4096 loses 9 % with either)
static int launch_p3_grad(const P3Args& a, const P3Args& b, float* scratch, unsigned* cnt, hipStream_t st, bool cnt_zeroed = false) {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 1;
    slots = 2 * cus;                                   // two resident blocks per CU
  }
  P3Grad p;
  // M != N (round 6): the gradient of the SMALLER factor has few tiles with the longer side as its K -- a 1024 x 4096 layer: 36 tiles of 2 x 128
  // K steps beside 528 tiles of 2 x 32, and the launch took the long tiles' 246 us where the work is ~95 us of the chip.  That product goes
  // second and ALL its tiles are split into K chunks as long as the other product's tiles (the chunk count = the ratio of the sides, even, <= 8).
  const bool rect_split = g_grad_split && g_grad_rect && scratch && a.e.K != b.e.K;
  const P3Args& lng = a.e.K > b.e.K ? a : b;          // (the longer K: the smaller output)
  const P3Args& sht = a.e.K > b.e.K ? b : a;
  const int Tl = (lng.e.M + 127) / 128, nl = Tl * (Tl + 1) / 2;
  const int ratio = (lng.e.K + sht.e.K / 2) / sht.e.K;
  const bool rect = rect_split && ratio >= 2 && nl <= kGradSplitMax;
  if (rect) { p.g[0] = sht; p.g[1] = lng; } else { p.g[0] = a; p.g[1] = b; }
  p.T0 = (p.g[0].e.M + 127) / 128; p.T1 = (p.g[1].e.M + 127) / 128;
  p.n0 = p.T0 * (p.T0 + 1) / 2; p.n1 = p.T1 * (p.T1 + 1) / 2;
  p.nchunk = kGradChunks; p.scratch = scratch; p.cnt = cnt;
  p.order = a.e.K == b.e.K ? g_grad_order : 0;       // (M != N: the two products' tiles cost differently, contiguous runs unbalance the XCDs)
  const int rem = (p.n0 + p.n1) % slots;
  p.nsplit = 0;
  if (rect) {
    p.nsplit = p.n1;
    p.nchunk = ratio >= 7 ? 8 : ratio >= 5 ? 6 : ratio >= 3 ? 4 : 2;
    if (p.nchunk > kGradChunks) p.nchunk = kGradChunks;
  }
  // a short last round (at most an eighth of the slots): twice that many tiles become eighth-size items, which the idle
  // slots of the last full round and one short extra round absorb
  // (only when the tiles of both products cost the same, M = N: measured -0.11 ms of 1.55 at 4096^2 -- blocks of a thin last
  // round run faster than the model's, each has its SIMDs to itself -- and +0.3 ms at 3000 x 5000, where they do not)
  if (!rect && g_grad_split && scratch && a.e.K == b.e.K && rem > 0 && rem <= slots / 8 && p.n0 + p.n1 > slots) {
    p.nsplit = 2 * rem;
    if (p.nsplit > p.n1) p.nsplit = p.n1;
    if (p.nsplit > kGradSplitMax) p.nsplit = kGradSplitMax;
    const int steps = (b.e.K + 31) / 32;               // per pair; every chunk needs at least one K step
    if (steps < kGradChunks / 2) p.nsplit = 0;
  }
  if (p.nsplit && !cnt_zeroed && hipMemsetAsync(cnt, 0, (size_t)p.nsplit * 4, st) != hipSuccess) return 1;
  const dim3 grid(p.n0 + p.n1 - p.nsplit + p.nsplit * p.nchunk);
  const bool none = p3_no_early(p.n0 + p.n1 >= slots / 2);
  if (!a.fmt) hipLaunchKernelGGL((k_gemm_p3_grad<0, 2>), grid, dim3(kThreads), 0, st, p);
  else if (p3_uses_te(a) || p3_uses_te(b)) {
    if (none) hipLaunchKernelGGL((k_gemm_p3_grad<1, 0, true>), grid, dim3(kThreads), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_p3_grad<1, 2, true>), grid, dim3(kThreads), 0, st, p);
  } else if (none) hipLaunchKernelGGL((k_gemm_p3_grad<1, 0>), grid, dim3(kThreads), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_p3_grad<1, 2>), grid, dim3(kThreads), 0, st, p);
  return (int)hipGetLastError();
}

// The apply of plan_apply (large branch, same association order and K ranges) on planes:
//   M < N:   prepare  Ql' -> Y0,  PP = planes(Ql'Ql),  F1 = planes(Qr),  F2 = planes(Qr')
//            apply    Y0 = planes(G'),  Y1 = planes(PP G),  Y2 = planes(Y1 Qr'),  out = Y2 Qr
//   M >= N:  prepare  Qr' -> Y0,  PP = planes(Qr'Qr),  F1 = planes(Ql),  F2 = planes(Ql')
//            apply    Y0 = planes(G),  Y1 = planes((G PP)'),  Y2 = planes((Ql Y1')'),  out = Ql' Y2'
// Every B operand is an (n, k) view, so a product that feeds the B side of the next one stores its result transposed.
static int planes_prepare(const float* Ql, const float* Qr, int M, int N, const KronWs& k, hipStream_t st) {
  const bool left = M < N;                            // Gram of the left factor
  const int ns = left ? M : N, nb = left ? N : M;
  const float* Qs = left ? Ql : Qr;
  const float* Qb = left ? Qr : Ql;
  const long nsp = pad128(ns), nbp = pad128(nb);
  PlaneMeta* pm = g_planes_f16 ? k.pmeta : nullptr;
  P3Buf QsT = {k.Y0, nsp, nsp, pm ? pm + kPmQs : pm}, F1 = {k.F1, nbp, nbp, pm ? pm + kPmF : pm};
  P3Buf PP = {k.PP, nsp, nsp, pm ? pm + kPmPP : pm};
  int e;
  if (pm && g_fused_prologue && kron_tile_scales(M, N)) {
    // (round 6) tile scales: both factors' planes from ONE sweep over their upper tiles (no max|.| launch, nothing below the diagonals
    // read or written), the Gram's planes straight from its epilogue (no fp32 Gram, no split launch)
    QsT.te = k.te + kTeY0 * kTeTable; PP.te = k.te + kTePP * kTeTable;
    const BalSide a = {Qs, ns, nullptr, nullptr, k.Y0, nullptr, QsT.te};
    const BalSide b = {Qb, nb, nullptr, k.F1, k.F2, k.te + kTeF1 * kTeTable, k.te + kTeF2 * kTeTable};
    if ((e = launch_factor_planes_ts(a, b, st))) return e;
    P3Args g = p3_args(QsT, QsT, ns, ns, ns, KHI_M | KHI_N);                            // Qs'Qs, symmetric
    g.e.sym = 1;
    p3_out_row(g, PP); p3_out_col(g, PP);
    return launch_p3(g, st);
  }
  // the maxima of both factors from one launch over their upper triangles (second array of the partial maxima for the bigger factor)
  const bool both = pm && ((reinterpret_cast<uintptr_t>(Qs) | reinterpret_cast<uintptr_t>(Qb)) & 15) == 0 && ns % 4 == 0 && nb % 4 == 0;
  if (both) {
    const int ba = ns < 1024 ? ns : 1024, bb = nb < 1024 ? nb : 1024;
    hipLaunchKernelGGL(k_absmax_tri2, dim3(ba + bb), dim3(kThreads), 0, st, Qs, ns, k.pm_part, ba, Qb, nb, k.pm_part + kPmPartMax,
                       &pm[kPmPP].amax, 1);
    if (hipGetLastError() != hipSuccess) return 1;
    QsT.part = k.pm_part; QsT.npart = ba;
    F1.part = k.pm_part + kPmPartMax; F1.npart = bb;
  } else if (pm && (e = launch_absmax(Qs, (long)ns * ns, QsT, k.pm_part, st, &pm[kPmPP].amax, 1))) return e;
  if ((e = launch_split3(Qs, 1, ns, ns, ns, QsT, st, SplitOpt{2, 0, 0, 0}))) return e;   // (x, k) = Qs[k][x]: zero where k > x, unread
  P3Args g = p3_args(QsT, QsT, ns, ns, ns, KHI_M | KHI_N);                              // Qs'Qs, symmetric
  g.e.sym = 1;
  if (g.fmt && g_planes_exact) {                      // (fp32 Gram in the workspace's Gram buffer, mirrored by the epilogue)
    if ((e = p3_chain(g, left ? k.Pl : k.Pr, &PP, nullptr, nullptr, nullptr, nullptr, st))) return e;
  } else {
    p3_out_row(g, PP); p3_out_col(g, PP);
    if ((e = launch_p3(g, st))) return e;
  }
  if (pm && !both && (e = launch_absmax(Qb, (long)nb * nb, F1, k.pm_part, st))) return e;   // (the split above is done with the array)
  P3Buf F2 = F1;
  F2.p = k.F2;
  return launch_split3_both(Qb, nb, 1, nb, nb, F1, F2, st, SplitOpt{1, 0, 0, 0});        // both forms from one read (upper tiles only)
}

static int planes_apply(const float* G, float* out, int M, int N, const KronWs& k, hipStream_t st) {
  const long Mp = pad128(M), Np = pad128(N);
  int e;
  P3AloneScope alone;
  PlaneMeta* pm = g_planes_f16 ? k.pmeta : nullptr;
  PlaneMeta *mPP = pm ? pm + kPmPP : pm, *mF = pm ? pm + kPmF : pm, *mG = pm ? pm + kPmG : pm, *mT = pm ? pm + kPmT : pm,
            *mA = pm ? pm + kPmA : pm;
  P3Buf Gin = {k.Y0, M < N ? Np : Mp, M < N ? Mp : Np, mG};                              // planes of G' (M < N) or G
  const bool ts = pm && g_fused_prologue && kron_tile_scales(M, N);                      // (as planes_prepare: the prepared state has tile scales)
  int* const te = k.te;
  if (ts) Gin.te = te + kTeY0 * kTeTable;
  // (the maxima of the two intermediates, slots kPmT and kPmA, start from zero: consecutive PlaneMeta, 8 floats)
  if (pm && !ts && (e = launch_absmax(G, (long)M * N, Gin, k.pm_part, st, &mT->scale, 8))) return e;
  if (M < N) {
    P3Buf PP = {k.PP, Mp, Mp, mPP}, F1 = {k.F1, Np, Np, mF}, F2 = {k.F2, Np, Np, mF};
    P3Buf Gt = Gin, T = {k.Y1, Mp, Np, mT}, A = {k.Y2, Mp, Np, mA};
    if (ts) {
      PP.te = te + kTePP * kTeTable; F1.te = te + kTeF1 * kTeTable; F2.te = te + kTeF2 * kTeTable;
      T.te = te + kTeY1 * kTeTable; A.te = te + kTeY2 * kTeTable;
      if ((e = launch_split_cols_ts(G, M, N, Gt, st))) return e;                         // one sweep, tile scales
    } else if ((e = launch_split3(G, 1, N, N, M, Gt, st))) return e;                     // (n, k = m) = G[m][n]
    P3Args g0 = p3_args(PP, Gt, M, N, M, 0);                                             // (Ql'Ql) G
    if ((e = p3_chain(g0, k.T, &T, nullptr, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
    P3Args g1 = p3_args(T, F1, M, N, N, KLO_N);                                          // (.) Qr'
    if ((e = p3_chain(g1, k.A, &A, nullptr, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
    P3Args g2 = p3_args(A, F2, M, N, N, KHI_N);                                          // (.) Qr
    g2.e.C = out; g2.e.ldc = N;
    return launch_p3_auto(g2, k.sk_scratch, k.sk_cnt, st);
  }
  P3Buf PP = {k.PP, Np, Np, mPP}, F1 = {k.F1, Mp, Mp, mF}, F2 = {k.F2, Mp, Mp, mF};
  P3Buf Gp = Gin, Tt = {k.Y1, Np, Mp, mT}, At = {k.Y2, Np, Mp, mA};
  if (ts) {
    PP.te = te + kTePP * kTeTable; F1.te = te + kTeF1 * kTeTable; F2.te = te + kTeF2 * kTeTable;
    Tt.te = te + kTeY1 * kTeTable; At.te = te + kTeY2 * kTeTable;
    if ((e = launch_split_rows_ts(G, M, N, Gp, st))) return e;
  } else if ((e = launch_split3(G, N, 1, M, N, Gp, st))) return e;
  P3Args g0 = p3_args(Gp, PP, M, N, N, 0);                                               // G (Qr'Qr)
  if ((e = p3_chain(g0, k.T, nullptr, &Tt, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
  P3Args g1 = p3_args(F1, Tt, M, N, M, KLO_M);                                           // Ql (.)
  if ((e = p3_chain(g1, k.A, nullptr, &At, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
  P3Args g2 = p3_args(F2, At, M, N, M, KHI_M);                                           // Ql' (.)
  g2.e.C = out; g2.e.ldc = N;
  return launch_p3_auto(g2, k.sk_scratch, k.sk_cnt, st);
}

// The apply WITHOUT a Gram (f16 x 2 planes only): out = Ql' (Ql ((G Qr') Qr)), four triangular plane products chained like the
// update's.  With factors that are new on every call -- the reference's pattern: an apply right after an update -- the Gram of the
// prepared form is made for one use only: its product, its planes and the second factor's maxima cost more than the fourth
// triangular product (4096^2: 1.01 -> 0.93 ms).  Uses the update's plane buffers; the prepared state (PP, F1, F2) is left alone.
static int planes_apply_direct(const float* Ql, const float* Qr, const float* G, float* out, int M, int N, const KronWs& k, hipStream_t st) {
  const long Mp = pad128(M), Np = pad128(N);
  P3AloneScope alone;
  PlaneMeta* pm = k.pmeta;
  P3Buf Lr = {k.Lr, Mp, Mp, pm + kPmL}, Rr = {k.Rr, Np, Np, pm + kPmR};
  const P3Buf Lc = {k.Lc, Mp, Mp, pm + kPmL}, Rc = {k.Rc, Np, Np, pm + kPmR};
  float* zero = &pm[kPmT].scale;                       // the maxima of the three intermediates (slots 5, 6, 7) start from zero
  int e;
  if (g_fused_prologue && kron_tile_scales(M, N)) {
    // (round 6) every plane set of the chain at tile scales: the factors (both forms of both) from one sweep over their upper tiles, G
    // from one sweep -- no max|.| launches
    int* const te = k.te;
    P3Buf Lr2 = Lr, Lc2 = Lc, Rr2 = Rr, Rc2 = Rc;
    Lr2.te = te + kTeLr * kTeTable; Lc2.te = te + kTeLc * kTeTable; Rr2.te = te + kTeRr * kTeTable; Rc2.te = te + kTeRc * kTeTable;
    const BalSide a = {Ql, M, nullptr, k.Lr, k.Lc, Lr2.te, Lc2.te}, b = {Qr, N, nullptr, k.Rr, k.Rc, Rr2.te, Rc2.te};
    if ((e = launch_factor_planes_ts(a, b, st))) return e;
    P3Buf Gp = {k.U0, Mp, Np, pm + kPmG};
    Gp.te = te + kTeU0 * kTeTable;
    if ((e = launch_split_rows_ts(G, M, N, Gp, st))) return e;
    P3Buf T1r = {k.U2, Mp, Np, pm + kPmT}, T2c = {k.U1, Np, Mp, pm + kPmA}, T3c = {k.U3, Np, Mp, pm + kPmA + 1};
    T1r.te = te + kTeU2 * kTeTable; T2c.te = te + kTeU1 * kTeTable; T3c.te = te + kTeU3 * kTeTable;
    P3Args s0 = p3_args(Gp, Rr2, M, N, N, KLO_N);                 // T1 = G Qr'
    if ((e = p3_chain(s0, k.T, &T1r, nullptr, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
    P3Args s1 = p3_args(T1r, Rc2, M, N, N, KHI_N);                // T2 = T1 Qr
    if ((e = p3_chain(s1, k.A, nullptr, &T2c, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
    P3Args s2 = p3_args(Lr2, T2c, M, N, M, KLO_M);                // T3 = Ql T2
    if ((e = p3_chain(s2, k.T, nullptr, &T3c, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
    P3Args s3 = p3_args(Lc2, T3c, M, N, M, KHI_M);                // out = Ql' T3
    s3.e.C = out; s3.e.ldc = N;
    return launch_p3_auto(s3, k.sk_scratch, k.sk_cnt, st);
  }
  if (((reinterpret_cast<uintptr_t>(Ql) | reinterpret_cast<uintptr_t>(Qr)) & 15) == 0 && M % 4 == 0 && N % 4 == 0) {
    const int ba = M < 1024 ? M : 1024, bb = N < 1024 ? N : 1024;
    hipLaunchKernelGGL(k_absmax_tri2, dim3(ba + bb), dim3(kThreads), 0, st, Ql, M, k.pm_part, ba, Qr, N, k.pm_part + kPmPartMax, zero, 12);
    if (hipGetLastError() != hipSuccess) return 1;
    Lr.part = k.pm_part; Lr.npart = ba; Rr.part = k.pm_part + kPmPartMax; Rr.npart = bb;
  } else {
    if ((e = launch_absmax(Ql, (long)M * M, Lr, k.pm_part, st, zero, 12))) return e;
    if ((e = launch_absmax(Qr, (long)N * N, Rr, k.pm_part + kPmPartMax, st))) return e;
  }
  if ((e = launch_split3_two(Ql, M, M, Lr, Qr, N, N, Rr, st, SplitOpt{1, 0, 0, 0}, &Lc, &Rc))) return e;
  P3Buf Gp = {k.U0, Mp, Np, pm + kPmG};
  if ((e = launch_absmax(G, (long)M * N, Gp, k.pm_part, st))) return e;                    // (the factors' split is done with the array)
  if ((e = launch_split3(G, N, 1, M, N, Gp, st))) return e;
  P3Buf T1r = {k.U2, Mp, Np, pm + kPmT}, T2c = {k.U1, Np, Mp, pm + kPmA}, T3c = {k.U3, Np, Mp, pm + kPmA + 1};
  if (kron_tile_scales(M, N)) {                                 // the chained results at tile scales: no fp32 round trip, no split launches
    T1r.te = k.te + kTeU2 * kTeTable; T2c.te = k.te + kTeU1 * kTeTable; T3c.te = k.te + kTeU3 * kTeTable;
  }
  P3Args s0 = p3_args(Gp, Rr, M, N, N, KLO_N);                  // T1 = G Qr'     (n, k) view of Qr' = Qr, k >= n
  if ((e = p3_chain(s0, k.T, &T1r, nullptr, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
  P3Args s1 = p3_args(T1r, Rc, M, N, N, KHI_N);                 // T2 = T1 Qr     (n, k) = Qr[k][n], k <= n
  if ((e = p3_chain(s1, k.A, nullptr, &T2c, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
  P3Args s2 = p3_args(Lr, T2c, M, N, M, KLO_M);                 // T3 = Ql T2     (m, k) = Ql[m][k], k >= m
  if ((e = p3_chain(s2, k.T, nullptr, &T3c, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
  P3Args s3 = p3_args(Lc, T3c, M, N, M, KHI_M);                 // out = Ql' T3   (m, k) = Ql[k][m], k <= m
  s3.e.C = out; s3.e.ldc = N;
  return launch_p3_auto(s3, k.sk_scratch, k.sk_cnt, st);
}

// The GEMM stages of plan_update on planes (same products and K ranges; the solves stay on the fp32 kernels):
//   after the balance:  Lr/Lc = planes(QlS / QlS'),  Rr/Rc = planes(QrS / QrS'),  U0 = planes(dG)
//   s0  U1 = planes((dG QrS')')            s1  U2, U3 = planes(A), planes(A'),  A = QlS (dG QrS')
//   after the solves:   U0, U1 = planes(Bt), planes(Bt')
//   s2  G1 = planes(triu(A A' - Bt Bt')), max -> scal[0]     s3  G2 = planes(triu(A'A - Bt'Bt)), max -> scal[1]
//   s4  QlOut = QlS - (step / max) G1 QlS                    s5  QrOut = QrS - (step / max) G2 QrS
// planes of the balanced factors (row and column forms): both chains of the update read them
static int planes_update_factors(int M, int N, const KronWs& k, hipStream_t st, PlaneMeta* pm) {
  const long Mp = pad128(M), Np = pad128(N);
  PlaneMeta *mL = pm ? pm + kPmL : pm, *mR = pm ? pm + kPmR : pm;        // (f16 x 2: the balance launch left the partial maxima)
  const int nb = balance_grid(M, N);
  const P3Buf Lr = {k.Lr, Mp, Mp, mL, k.pm_part + 2 * kPmPartMax, nb}, Lc = {k.Lc, Mp, Mp, mL};
  const P3Buf Rr = {k.Rr, Np, Np, mR, k.pm_part + 3 * kPmPartMax, nb}, Rc = {k.Rc, Np, Np, mR};
  int e;
  // (tri: the factors are upper triangular by contract -- the tiles below the diagonal become zeros without being read;
  //  f16 x 2: both factors, both forms, one launch)
  if (pm) return launch_split3_two(k.QlS, M, M, Lr, k.QrS, N, N, Rr, st, SplitOpt{1, 0, 0, 0}, &Lc, &Rc);
  if ((e = launch_split3_both(k.QlS, M, 1, M, M, Lr, Lc, st, SplitOpt{1, 0, 0, 0}))) return e;
  return launch_split3_both(k.QrS, N, 1, N, N, Rr, Rc, st, SplitOpt{1, 0, 0, 0});
}

static int planes_update_front(const float* dG, int M, int N, const KronWs& k, hipStream_t st, PlaneMeta* pm, int phase = 3) {     // phase: 1 = dG's planes, 2 = the products
  const long Mp = pad128(M), Np = pad128(N);
  auto slot = [&](int i) { return pm ? pm + i : pm; };
  P3Buf Lr = {k.Lr, Mp, Mp, slot(kPmL)}, Rr = {k.Rr, Np, Np, slot(kPmR)};
  if (k.factor_ts) { Lr.te = k.te + kTeLr * kTeTable; Rr.te = k.te + kTeRr * kTeTable; }
  P3Buf dGp = {k.U0, Mp, Np, slot(kPmdG)};
  P3Buf Tt = {k.U1, Np, Mp, slot(kPmUT)}, Ar = {k.U2, Mp, Np, slot(kPmUA)}, Ac = {k.U3, Np, Mp, slot(kPmUA)};
  if (pm && kron_tile_scales(M, N)) { Tt.te = k.te + kTeU1 * kTeTable; Ar.te = k.te + kTeU2 * kTeTable; Ac.te = k.te + kTeU3 * kTeTable; }
  int e;
  if (k.factor_ts) {                                            // (round 6, the tile-scale inverse route: dG's planes in one sweep)
    dGp.te = k.te + kTeU0 * kTeTable;
    if ((phase & 1) && (e = launch_split_rows_ts(dG, M, N, dGp, st))) return e;
  } else if (phase & 1) {
    if (pm && (e = launch_absmax(dG, (long)M * N, dGp, k.pm_part + kPmPartMax, st))) return e;     // (the side stream's array)
    if ((e = launch_split3(dG, N, 1, M, N, dGp, st))) return e;
  }
  if (!(phase & 2)) return 0;
  P3Args s0 = p3_args(dGp, Rr, M, N, N, KLO_N);                 // T = dG QrS'  (:173); (n, k) view of QrS' = QrS
  if ((e = p3_chain(s0, k.T, nullptr, &Tt, nullptr, k.sk_scratch, k.sk_cnt, st))) return e;
  P3Args s1 = p3_args(Lr, Tt, M, N, M, KLO_M);                  // A = QlS T
  return p3_chain(s1, k.A, &Ar, &Ac, nullptr, k.sk_scratch, k.sk_cnt, st);
}

static int g_pair_order = 1;    // tuning key 27: 1 (default) = the factor updates' tiles in 4 x 4 patches (gemm_tile_from_id case 3': 4096^2 update
                                // 2.54 -> 2.48 ms, the launch 251 -> 183 us; 6144^2 7.10 -> 6.91; bit-identical results), 0 = whole tile rows per XCD
static int planes_update_back(float* QlOut, float* QrOut, int M, int N, float step, float tiny, const KronWs& k, hipStream_t st,
                              PlaneMeta* pm, bool bt_planes_ready = false, const P3Buf* bt_br = nullptr, const P3Buf* bt_bc = nullptr) {
  const long Mp = pad128(M), Np = pad128(N);
  auto slot = [&](int i) { return pm ? pm + i : pm; };
  const bool ts = pm && kron_tile_scales(M, N);
  P3Buf Lc = {k.Lc, Mp, Mp, slot(kPmL)}, Rc = {k.Rc, Np, Np, slot(kPmR)};
  if (k.factor_ts) { Lc.te = k.te + kTeLc * kTeTable; Rc.te = k.te + kTeRc * kTeTable; }
  P3Buf G1 = {k.G1, Mp, Mp, slot(kPmG1)}, G2 = {k.G2, Np, Np, slot(kPmG2)};
  P3Buf Br = {k.U0, Mp, Np, slot(kPmBt)};
  P3Buf Bc = {k.U1, Np, Mp, slot(kPmBt)}, Ar = {k.U2, Mp, Np, slot(kPmUA)}, Ac = {k.U3, Np, Mp, slot(kPmUA)};
  if (ts) {                                             // (planes_update_front left A's planes at tile scales)
    Ar.te = k.te + kTeU2 * kTeTable; Ac.te = k.te + kTeU3 * kTeTable;
    G1.te = k.te + kTeG1 * kTeTable; G2.te = k.te + kTeG2 * kTeTable;
  }
  if (bt_br) { Br = *bt_br; Bc = *bt_bc; }              // (the solves left Bt's planes elsewhere)
  int e;
  if (!bt_planes_ready) {                               // (the inverse route's last product has made them)
    if (pm && (e = launch_absmax(k.Bt, (long)M * N, Br, k.pm_part, st))) return e;
    if ((e = launch_split3_both(k.Bt, N, 1, M, N, Br, Bc, st))) return e;
  }
  P3Args s2 = p3_args(Ar, Ar, M, M, N, 0);                      // grad1 = triu(A A' - Bt Bt')  (:175)
  s2.A2 = p3_of(Br); s2.B2 = p3_of(Br); s2.e.A2 = k.Bt; s2.e.K2 = N;
  s2.e.epi = EPI_TRIU_MAX; s2.e.maxout = k.scal + 0;
  P3Args s3 = p3_args(Ac, Ac, N, N, M, 0);                      // grad2 = triu(A'A - Bt'Bt)  (:176)
  s3.A2 = p3_of(Bc); s3.B2 = p3_of(Bc); s3.e.A2 = k.Bt; s3.e.K2 = M;
  s3.e.epi = EPI_TRIU_MAX; s3.e.maxout = k.scal + 1;
  if (ts) {
    // tile scales: the gradient grid's epilogue does triu and max|.| on its registers (the step size of :177-178) and writes the planes
    // of every upper tile at that tile's own maximum -- no fp32 gradients, no split launch
    p3_out_row(s2, G1);
    p3_out_row(s3, G2);
    if ((e = launch_p3_grad(s2, s3, k.split_scratch, k.split_cnt, st, k.factor_ts))) return e;      // (factor_ts: the prologue zeroed the counters)
  } else if (pm && g_planes_exact) {
    // f16 x 2 (see p3_chain): the gradients in fp32 (the epilogue's triu and max|.| as on the fp32 route), then their planes
    // with the scale of that very maximum.  Tiles below the diagonal are not written and not read (K ranges of s4 / s5).
    s2.e.C = k.g1; s2.e.ldc = M;
    s3.e.C = k.g2; s3.e.ldc = N;
    if ((e = launch_p3_grad(s2, s3, k.split_scratch, k.split_cnt, st))) return e;
    P3Buf g1p = G1, g2p = G2;
    g1p.part = k.scal + 0; g2p.part = k.scal + 1; g1p.npart = g2p.npart = 1;
    // (tri: the gradient grid wrote the upper tiles only; what lies below the diagonal in k.g1 / k.g2 is whatever the workspace
    //  held -- on the inverse route the fp32 inverses -- and becomes defined zeros in the planes)
    if ((e = launch_split3_two(k.g1, M, M, g1p, k.g2, N, N, g2p, st, SplitOpt{1, 0, 0, 0}))) return e;      // (one launch for both)
  } else {
    p3_out_row(s2, G1);
    p3_out_row(s3, G2);
    if ((e = launch_p3_grad(s2, s3, k.split_scratch, k.split_cnt, st))) return e;
  }
  const int pord = g_pair_order ? KORD_PATCH : 0;
  P3Args s4 = p3_args(G1, Lc, M, M, M, KLO_M | KHI_N | pord);   // QlS - (step1 grad1) QlS  (:179); (n, k) view of QlS = QlS'
  s4.e.epi = EPI_D_MINUS; s4.e.C = QlOut; s4.e.ldc = M; s4.e.D = k.QlS; s4.e.ldd = M;
  s4.e.scale_max = k.scal + 0; s4.e.step = step; s4.e.tiny = tiny;
  P3Args s5 = p3_args(G2, Rc, N, N, N, KLO_M | KHI_N | pord);
  s5.e.epi = EPI_D_MINUS; s5.e.C = QrOut; s5.e.ldc = N; s5.e.D = k.QrS; s5.e.ldd = N;
  s5.e.scale_max = k.scal + 1; s5.e.step = step; s5.e.tiny = tiny;
  s4.lower_zero = s5.lower_zero = k.factor_ts ? 1 : 0;         // (the fused prologue leaves QlS / QrS unwritten below the diagonal)
  return launch_p3_two(s4, s5, st);
}

// ---- the triangular solves of psgd.py:174 through explicit inverses (large fp32 update on f16 x 2 planes) ---------------
// Round 3 costed this route on the bf16 x 3 planes at break-even (profiles/r03_trsm_inverse_route.txt): a solve as one
// product was 0.33 ms and the inversion another n^3 / 3 per factor.  On the f16 x 2 planes a triangular 4096^3 product is
// 0.17 ms, and the substitution strips (16 launches of 41 us, 14 update products between them) are what is left of the
// update's critical path.  Inv = Q^-1 by recursive doubling, [A B; 0 C]^-1 = [A^-1, -A^-1 B C^-1; 0, C^-1]: the 128-blocks
// in one launch (k_tri_inv128, from the 32-blocks the balance launch inverted), then per level b = 128, 256, ... four
// launches for ALL pairs of b-blocks: planes of the b-blocks inverted so far (both forms, one scale from the running
// max|Inv|), T = A^-1 B (fp32 + max), planes of -T, W = (-T) C^-1 straight into Inv (fp32 + running max).  K ranges:
// A^-1 upper (k >= m0) and inside A's half (KBLK_HI_M); C^-1 upper (k <= n0 + 127) and inside C's half (KBLK_LO_N).  At
// the end the whole inverse is split once into column-form planes, the form both solves read:
//   X1 = dX Ri   (B operand (n, k) = Ri[k][n], k <= n)         Bt = Li' X1   (A operand (m, k) = Li[k][m], k <= m)
// Accuracy: tools/group_inverse_error_study.py -- the solve through fp32 inverses stays within 1.3-1.7x of fp32
// substitution up to cond 1e9; the parity tests on ill-conditioned factors hold their bars on this route.
struct InvSide {
  const float* Q; int n; const float* dinv;      // balanced factor, its inverted 32-blocks
  float* Inv; float* Tf;                         // fp32 [n x n]: the inverse; the levels' A^-1 B
  P3Buf Qc, Ir, Ic, Tp;                          // column-form planes of Q; planes of the inverse (row / column form); of -T
  PlaneMeta* mT;                                 // 6 slots: one per level
  int b0 = 128;                                  // size of the diagonal blocks tri_inverse_blocks inverted (the first level's b)
};
// (Round 4, measured and not kept: the K ranges of the top levels' tiles dealt to 2-4 blocks -- 4096^2 fp32 update 2.74 -> 2.82 ms,
//  bf16 operands 2.06 -> 2.19: the levels run beside full-chip products of the other chain, and twice the workgroups are twice the
//  slots to wait for.  profiles/r04_inv_ab.txt)
static int g_trsm_inv = 1;      // tuning key 11: 0 = the solves of every size stay on the substitution strips

static int g_inv_strip512 = 1;  // tuning key 23: 0 = the inversion starts from k_tri_inv128 + levels 128, 256 for every n
static int tri_inverse_blocks(InvSide& f, hipStream_t st) {           // the inverted 128-blocks (f.b0 = 128) or 512-blocks (512)
  if (g_inv_strip512 && f.n % 512 == 0 && (f.n & 1) == 0) {
    hipLaunchKernelGGL(k_tri_inv512, dim3((f.n / 512) * 32), dim3(kThreads), 0, st, f.Q, f.n, f.dinv, f.Inv, &f.Ir.meta->amax);
    f.Ir.part = f.Ic.part = &f.Ir.meta->amax;
    f.Ir.npart = f.Ic.npart = 1;
    f.b0 = 512;
    return (int)hipGetLastError();
  }
  f.b0 = 128;
  static DeviceOnce attr_set;
  const size_t lds = (size_t)2 * 128 * 129 * sizeof(float);
  if (attr_set.needed()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tri_inv128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) return 1;
    attr_set.done();
  }
  hipLaunchKernelGGL(k_tri_inv128, dim3((f.n + 127) / 128), dim3(kThreads), lds, st, f.Q, f.n, f.dinv, f.Inv, &f.Ir.meta->amax);
  f.Ir.part = f.Ic.part = &f.Ir.meta->amax;
  f.Ir.npart = f.Ic.npart = 1;
  return (int)hipGetLastError();
}
static int tri_inverse_level(const InvSide& f, int b, int level, hipStream_t st) {     // b-blocks -> 2b-blocks
  const int n = f.n;
  int e;
  if ((e = launch_split3_both(f.Inv, n, 1, n, n, f.Ir, f.Ic, st, SplitOpt{1, b, 0, 0}))) return e;     // the inverted b-blocks
  const int tb = b / 128, pairs = (n + 2 * b - 1) / (2 * b);
  P3Args g1 = p3_args(f.Ir, f.Qc, n, n, n, KLO_M | KBLK_HI_M);                     // T = A^-1 B
  g1.e.kblk = 2 * b; g1.e.C = f.Tf; g1.e.ldc = n; g1.ometa = f.mT + level;
  const bool full = p3_no_early(pairs * tb * tb >= p3_block_slots() / 2);           // (two inversions share the chip)
  if (full) hipLaunchKernelGGL((k_gemm_p3_blk<1, 0>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g1, b);
  else hipLaunchKernelGGL((k_gemm_p3_blk<1, 2>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g1, b);
  if (hipGetLastError() != hipSuccess) return 1;
  P3Buf tp = f.Tp;
  tp.meta = f.mT + level; tp.part = &tp.meta->amax; tp.npart = 1;
  if ((e = launch_split3(f.Tf, n, 1, n, n, tp, st, SplitOpt{0, 2 * b, 1, 1}))) return e;              // planes of -T
  P3Args g2 = p3_args(tp, f.Ic, n, n, n, KHI_N | KBLK_LO_N);                       // W = (-T) C^-1 into Inv
  g2.e.kblk = 2 * b; g2.e.C = f.Inv; g2.e.ldc = n; g2.ometa = f.Ir.meta;
  if (full) hipLaunchKernelGGL((k_gemm_p3_blk<1, 0>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g2, b);
  else hipLaunchKernelGGL((k_gemm_p3_blk<1, 2>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g2, b);
  return (int)hipGetLastError();
}
// The same level with tile scales (f.Ir.te, f.Ic.te, f.Tp.te): T's epilogue writes the planes of -T, W's epilogue the planes of the new
// off-diagonal blocks of the inverse in both forms -- two launches per level instead of four, nothing in fp32, and no final split: the
// inverted b0-blocks are split once (tri_inverse_pair), every later tile of Ir / Ic comes out of the product that makes it.
static inline bool inv_tile_scales(const InvSide& f) { return f.Ir.te && f.Ic.te && f.Tp.te; }
static int tri_inverse_level_ts(const InvSide& f, int b, hipStream_t st) {
  const int n = f.n;
  const int tb = b / 128, pairs = (n + 2 * b - 1) / (2 * b);
  const bool full = p3_no_early(pairs * tb * tb >= p3_block_slots() / 2);
  P3Args g1 = p3_args(f.Ir, f.Qc, n, n, n, KLO_M | KBLK_HI_M);                     // -T = -(A^-1 B)
  g1.e.kblk = 2 * b; g1.neg = 1;
  p3_out_row(g1, f.Tp);
  if (full) hipLaunchKernelGGL((k_gemm_p3_blk<1, 0, true>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g1, b);
  else hipLaunchKernelGGL((k_gemm_p3_blk<1, 2, true>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g1, b);
  if (hipGetLastError() != hipSuccess) return 1;
  P3Args g2 = p3_args(f.Tp, f.Ic, n, n, n, KHI_N | KBLK_LO_N);                     // W = (-T) C^-1: the block of the inverse
  g2.e.kblk = 2 * b;
  p3_out_row(g2, f.Ir);
  p3_out_col(g2, f.Ic);
  if (full) hipLaunchKernelGGL((k_gemm_p3_blk<1, 0, true>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g2, b);
  else hipLaunchKernelGGL((k_gemm_p3_blk<1, 2, true>), dim3(pairs * tb * tb), dim3(kThreads), 0, st, g2, b);
  return (int)hipGetLastError();
}
// column-form planes of the inverted h-blocks on the diagonal: (x, k) = Inv[k][x], k <= x, both inside one block (h >= n: the whole inverse)
static int tri_inverse_planes(const InvSide& f, int h, hipStream_t st) {
  return launch_split3(f.Inv, 1, f.n, f.n, f.n, f.Ic, st, SplitOpt{2, h < f.n ? h : 0, 0, 0});
}
// Two inversions on two streams, their launches queued level by level in turn (either stream has work early: a caller whose
// host thread is not far ahead of the device would otherwise leave the second stream idle for the ~25 launches of the first).
// h: the doubling stops at diagonal blocks of h (a power of two times 128; >= n: the whole inverse).
static int tri_inverse_pair(InvSide a, hipStream_t sa, InvSide b, hipStream_t sb, int h) {
  int e;
  if ((e = tri_inverse_blocks(a, sa)) || (e = tri_inverse_blocks(b, sb))) return e;
  const bool tsa = inv_tile_scales(a), tsb = inv_tile_scales(b);
  // tile scales: the planes (both forms) of the inverted b0-blocks, once
  if (tsa && (e = launch_split3_both(a.Inv, a.n, 1, a.n, a.n, a.Ir, a.Ic, sa, SplitOpt{1, a.b0, 0, 0}))) return e;
  if (tsb && (e = launch_split3_both(b.Inv, b.n, 1, b.n, b.n, b.Ir, b.Ic, sb, SplitOpt{1, b.b0, 0, 0}))) return e;
  int level = 0;
  for (int w = 128; w < h && (w < a.n || w < b.n); w *= 2, ++level) {
    if (w >= a.b0 && w < a.n && (e = tsa ? tri_inverse_level_ts(a, w, sa) : tri_inverse_level(a, w, level, sa))) return e;
    if (w >= b.b0 && w < b.n && (e = tsb ? tri_inverse_level_ts(b, w, sb) : tri_inverse_level(b, w, level, sb))) return e;
  }
  if (!tsa && (e = tri_inverse_planes(a, h, sa))) return e;
  return tsb ? 0 : tri_inverse_planes(b, h, sb);
}

// ---- the two solves of psgd.py:174 through the inverses of the DIAGONAL h-BLOCKS of the balanced factors (round 4) --------------
// The doubling's top level is its most expensive and least efficient one (n = 4096: two products of 256 tiles with K chains of up
// to 64 steps, one workgroup per CU: 190 us alone, 330 beside a full-chip product), and it only exists to turn the solve into ONE
// product.  Stopping at h-blocks (h = 2048) and solving blocked, right-looking over the n / h block columns,
//     X1_j = W_j Ri_jj,        W_{>j} -= X1_j R[j, >j]        (W = dX at the start)
//     Bt_i = Li_ii' V_i,       V_{>i} -= L[i, >i]' Bt_i       (V = X1 at the start)
// costs the same product flops (M n^2 / 2 per solve: the trailing updates are what the top level's T = A^-1 B, W = -T C^-1 would have
// folded into the inverse) in products of 512 full-K tiles each, and drops that level from both inversion chains.  Every piece that
// feeds a product is split into f16 x 2 planes at its actual maximum (the running max|.| its producers accumulate), like p3_chain.
static int g_inv_order = -1;    // tuning key 25: 1 = both inversions ahead of the products of :173 (those then run beside X1 and Bt), 0 = the products
                                // first on the side stream, beside Qr's inversion; -1 (default) = by the shape (kron_inv_first)
// Both inversions first -- undisturbed by full-chip products, every launch of a level at its isolated time, the chip mostly idle for
// ~0.4 ms -- pays once both factors reach 4096 (4096^2 fp32 update 2.66 -> 2.56 ms; 2048 x 4096, 3072^2, 6144^2 within 1 % either way:
// profiles/r04_kron_update_notes.txt)
// Round 5, with tile scales (no split launches left inside the chains): both inversions first wins or ties on every shape up to 4096^2
// elements (2304 x 2048 0.887 -> 0.812 ms, 4096 x 2048 1.425 -> 1.310, 8192 x 2048 3.69 -> 3.62, bf16 operands 2.69 -> 2.47; 3072^2 equal)
// and loses 2 % at 6144^2, where the products dwarf the chains (profiles/r05_order_ab.txt).
bool kron_inv_first(int M, int N) { return g_inv_order < 0 ? (M * (long)N <= 4096L * 4096) : g_inv_order == 1; }
static int g_inv_blk = 2048;    // tuning key 24: h (0 = whole inverses and one product per solve, the round-3 form)
struct BlkSolve {
  int M, N, h;
  PlaneMeta* pm;                 // kPmSlots slots, zero on entry
  float* part;                   // kPmPartMax partial maxima (main stream)
  InvSide L, R;
  const float* X0;               // dX, fp32 [M x N]
  P3Buf X0p;                     // its row-form planes (made here)
  float *X1, *Bt;                // fp32 [M x N]; Bt doubles as the W of the right solve
  P3Buf pa, pb;                  // two transient plane buffers, pad128(M) x pad128(N) elements each
  P3Buf Br, Bc;                  // planes of Bt, row / column form (p = nullptr: not wanted)
  // tile scales (pa.te, pb.te, x1c.te, Br.te, Bc.te all set): every piece's planes come out of the epilogue of the product that makes it
  P3Buf x1c;                     // column-form planes of X1 / V: (x = n, k = m), [pad128(N) x pad128(M)]
  bool bt_fp32 = true;           // the caller reads Bt in fp32 as well
  float* sk_scratch = nullptr; int sk_slots = 0; unsigned* sk_cnt = nullptr;      // (launch_p3_solve; null: whole tiles)
};
static inline bool blk_tile_scales(const BlkSolve& s) { return s.pa.te && s.pb.te && s.x1c.te && s.h % 128 == 0; }
static P3 p3_sub(const P3Buf& b, long x0, long k0) {              // the (x >= x0, k >= k0) corner of a plane set (k0 a multiple of 32;
  P3 v = p3_of(b);                                                  // of 128, like x0, when the set has tile scales)
  v.p += (k0 / 32) * v.ts + x0 * 32;
  if (v.te) v.te += (x0 / 128) * kTeLd + k0 / 128;
  return v;
}
static P3Args blk_product(const P3& A, const P3& B, int M, int N, int K, int kmode, float* C, int ldc, const float* D, int ldd,
                          PlaneMeta* ometa) {
  P3Args g = {};
  g.A = A; g.B = B; g.fmt = 1;
  g.e.M = M; g.e.N = N; g.e.K = K; g.e.kmode = kmode;
  g.e.epi = D ? EPI_D_MINUS : EPI_STORE;
  g.e.C = C; g.e.ldc = ldc; g.e.D = D; g.e.ldd = ldd;
  g.ometa = ometa;
  return g;
}
enum { kPmPieceX = kPmStrip, kPmPieceW = kPmStrip + 8, kPmPieceV = kPmStrip + 16 };      // (the strips' slots: free on this route)

// dX planes, both inversions (R on `main`, L on `side`), then X1 = dX R^-1 on `main`
// (l_ready: recorded on `side` behind L's inversion, for callers that put more work on `side` before the join)
// (x0_ready: an event of the caller's.  Given one, dX's planes are made on `side` AHEAD of L's inversion -- which is not needed before
//  the left solve -- so that R's inversion, the head of the critical path, starts ~75 us earlier; `main` waits for the event before X1.)
// (x0_stream: a third stream for dX's planes instead of `side` -- L's inversion then starts at the fork point too)
static int blk_solves_front(BlkSolve& s, hipStream_t main, hipStream_t side, hipEvent_t l_ready = nullptr, hipEvent_t x0_ready = nullptr,
                            hipStream_t x0_stream = nullptr) {
  const int M = s.M, N = s.N, h = s.h;
  int e;
  const bool x0_side = x0_ready && side != main;
  hipStream_t sx = x0_side ? (x0_stream ? x0_stream : side) : main;
  if (s.X0p.te) {                                               // (tile scales: one sweep)
    if ((e = launch_split_rows_ts(s.X0, M, N, s.X0p, sx))) return e;
  } else {
    if (!s.X0p.part && (e = launch_absmax(s.X0, (long)M * N, s.X0p, s.part, sx))) return e;      // (unless the caller has the maxima)
    if ((e = launch_split3(s.X0, N, 1, M, N, s.X0p, sx))) return e;
  }
  if (x0_side && hipEventRecord(x0_ready, sx) != hipSuccess) return 1;
  if ((e = tri_inverse_pair(s.R, main, s.L, side, h))) return e;
  if (l_ready && hipEventRecord(l_ready, side) != hipSuccess) return 1;
  if (x0_side && hipStreamWaitEvent(main, x0_ready, 0) != hipSuccess) return 1;
  PlaneMeta* mX1 = s.pm + kPmX1;
  const int nb = (N + h - 1) / h;
  const bool ts = blk_tile_scales(s);
  for (int j = 0; j < nb && ts; ++j) {
    // Tile scales: the product that makes a piece writes that piece's planes (row form for the trailing update, column form of X1 for
    // the left solve) -- the chain is  d_0, t_0, d_1, ...  with nothing in between (it was d, split, t, split: the splits 140 us each
    // beside the full-chip products of the other stream).
    const int c0 = j * h, hj = N - c0 < h ? N - c0 : h;
    P3 Wp = j == 0 ? p3_sub(s.X0p, 0, 0) : p3_of(s.pb);
    P3Args d = blk_product(Wp, p3_sub(s.R.Ic, c0, c0), M, hj, hj, KHI_N, s.X1 + c0, N, nullptr, 0, nullptr);     // X1_j = W_j Ri_jj
    P3Buf Xj = s.pa;
    Xj.ld = pad128(hj);
    if (j < nb - 1) p3_out_row(d, Xj);
    p3_out_col_at(d, s.x1c, c0, 0);
    if ((e = launch_p3_solve(d, s.sk_scratch, s.sk_slots, s.sk_cnt, main))) return e;
    if (j == nb - 1) break;
    const int c1 = c0 + hj;
    P3Args t = blk_product(p3_of(Xj), p3_sub(s.R.Qc, c1, c0), M, N - c1, hj, 0, s.Bt + c1, N, (j == 0 ? s.X0 : s.Bt) + c1, N, nullptr);
    s.pb.ld = pad128(N - c1);                                         // W_{>j}: the next block column is its first h columns
    p3_out_row(t, s.pb);
    if ((e = launch_p3_solve(t, s.sk_scratch, s.sk_slots, s.sk_cnt, main))) return e;
  }
  for (int j = 0; j < nb && !ts; ++j) {
    const int c0 = j * h, hj = N - c0 < h ? N - c0 : h;
    P3 Wp = j == 0 ? p3_sub(s.X0p, 0, 0) : p3_of(s.pb);
    if (j == 0) Wp.meta = s.X0p.meta;
    // X1_j = W_j Ri_jj   (B operand (n, k) = Ri[k][n], k <= n)
    P3Args d = blk_product(Wp, p3_sub(s.R.Ic, c0, c0), M, hj, hj, KHI_N, s.X1 + c0, N, nullptr, 0, mX1);
    if ((e = launch_p3(d, main))) return e;
    if (j == nb - 1) break;
    P3Buf Xj = s.pa;                                                  // row-form planes of X1_j (the running max|X1| as their scale)
    Xj.ld = pad128(hj); Xj.meta = s.pm + kPmPieceX + j; Xj.part = &mX1->amax; Xj.npart = 1;
    if ((e = launch_split3(s.X1 + c0, N, 1, M, hj, Xj, main))) return e;
    // W_{>j} = W_{>j} - X1_j R[j, >j]   (B operand (n, k) = R[k][n]: the factor's column-form planes)
    const int c1 = c0 + hj;
    PlaneMeta* mW = s.pm + kPmPieceW + j;
    P3Args t = blk_product(p3_of(Xj), p3_sub(s.R.Qc, c1, c0), M, N - c1, hj, 0, s.Bt + c1, N, (j == 0 ? s.X0 : s.Bt) + c1, N, mW);
    if ((e = launch_p3(t, main))) return e;
    const int hn = N - c1 < h ? N - c1 : h;
    P3Buf Wn = s.pb;                                                  // row-form planes of W_{j+1}
    Wn.ld = pad128(hn); Wn.meta = s.pm + kPmPieceW + j; Wn.part = &mW->amax; Wn.npart = 1;
    if ((e = launch_split3(s.Bt + c1, N, 1, M, hn, Wn, main))) return e;
    s.pb.ld = Wn.ld; s.pb.meta = Wn.meta;
  }
  return 0;
}

// Bt = L^-T X1 (after the caller's join: needs L's inverse), and the planes of Bt when wanted
static int blk_solves_back(BlkSolve& s, hipStream_t main) {
  const int M = s.M, N = s.N, h = s.h;
  int e;
  PlaneMeta *mX1 = s.pm + kPmX1, *mBt = s.pm + kPmBt;
  const int mb = (M + h - 1) / h;
  if (blk_tile_scales(s) && s.Bc.te) {
    // tile scales: V's column-form planes are in x1c (the products of the right solve and the trailing updates below write them),
    // every Bt_i goes straight into the planes of Bt the gradient grid reads, and its column form is the trailing update's operand
    for (int i = 0; i < mb; ++i) {
      const int r0 = i * h, hi = M - r0 < h ? M - r0 : h;
      P3Args d = blk_product(p3_sub(s.L.Ic, r0, r0), p3_sub(s.x1c, 0, r0), hi, N, hi, KHI_M,
                             s.bt_fp32 ? s.Bt + (long)r0 * N : nullptr, N, nullptr, 0, nullptr);             // Bt_i = Li_ii' V_i
      if (s.Br.p) p3_out_row_at(d, s.Br, r0, 0);
      p3_out_col_at(d, s.Bc, 0, r0);
      if ((e = launch_p3_solve(d, s.sk_scratch, s.sk_slots, s.sk_cnt, main))) return e;
      if (i == mb - 1) break;
      const int r1 = r0 + hi;
      P3Args t = blk_product(p3_sub(s.L.Qc, r1, r0), p3_sub(s.Bc, 0, r0), M - r1, N, hi, 0, s.X1 + (long)r1 * N, N,
                             s.X1 + (long)r1 * N, N, nullptr);                                                // V_{>i} -= L[i, >i]' Bt_i
      p3_out_col_at(t, s.x1c, 0, r1);
      if ((e = launch_p3_solve(t, s.sk_scratch, s.sk_slots, s.sk_cnt, main))) return e;
    }
    return 0;
  }
  for (int i = 0; i < mb; ++i) {
    const int r0 = i * h, hi = M - r0 < h ? M - r0 : h;
    P3Buf Vi = s.pb;                                                  // column-form planes of V_i: (x = n, k = m) = V[m][n]
    Vi.rows = pad128(N); Vi.ld = pad128(hi); Vi.meta = s.pm + kPmPieceV + i;
    Vi.part = i == 0 ? &mX1->amax : &(s.pm + kPmPieceV + 8 + i - 1)->amax; Vi.npart = 1;
    if ((e = launch_split3(s.X1 + (long)r0 * N, 1, N, N, hi, Vi, main))) return e;
    // Bt_i = Li_ii' V_i   (A operand (m, k) = Li[k][m], k <= m)
    P3Args d = blk_product(p3_sub(s.L.Ic, r0, r0), p3_of(Vi), hi, N, hi, KHI_M, s.Bt + (long)r0 * N, N, nullptr, 0, mBt);
    if ((e = launch_p3(d, main))) return e;
    if (i == mb - 1) break;
    P3Buf Bi = s.pa;                                                  // column-form planes of Bt_i
    Bi.rows = pad128(N); Bi.ld = pad128(hi); Bi.meta = s.pm + kPmPieceX + 4 + i; Bi.part = &mBt->amax; Bi.npart = 1;
    if ((e = launch_split3(s.Bt + (long)r0 * N, 1, N, N, hi, Bi, main))) return e;
    // V_{>i} = V_{>i} - L[i, >i]' Bt_i   (A operand (m, k) = L[k][m]: the factor's column-form planes), in place in X1
    const int r1 = r0 + hi;
    P3Args t = blk_product(p3_sub(s.L.Qc, r1, r0), p3_of(Bi), M - r1, N, hi, 0, s.X1 + (long)r1 * N, N, s.X1 + (long)r1 * N, N,
                           s.pm + kPmPieceV + 8 + i);
    if ((e = launch_p3(t, main))) return e;
  }
  if (!s.Br.p) return 0;
  P3Buf br = s.Br, bc = s.Bc;
  br.part = bc.part = &mBt->amax; br.npart = bc.npart = 1;
  return launch_split3_both(s.Bt, N, 1, M, N, br, bc, main);
}

// The route for callers outside this file (kron_shared.h: the bf16-operand update): own workspace, same launches.
struct InvSolveWs {
  PlaneMeta* pm; float* part; int* te;
  __bf16 *Lc, *Rc, *IrL, *IcL, *TpL, *IrR, *IcR, *TpR, *DXp, *X1p;
  float *InvL, *InvR, *TfL, *TfR;
  __bf16 *Pa, *Pb;                 // transient planes of the blocked solves' pieces
  int64_t total;
};
static InvSolveWs inv_solve_layout(char* base, int M, int N) {
  InvSolveWs k;
  const int64_t Mp = pad128(M), Np = pad128(N);
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base + off; off = align256(off + bytes); return p; };
  auto planes = [&](int64_t elems) { return reinterpret_cast<__bf16*>(take(elems * 4)); };        // (two fp16 planes)
  k.pm = reinterpret_cast<PlaneMeta*>(take(kPmSlots * sizeof(PlaneMeta)));
  k.part = reinterpret_cast<float*>(take(4 * kPmPartMax * 4));
  k.te = reinterpret_cast<int*>(take((int64_t)kTeSlots * kTeTable * 4));
  k.Lc = planes(Mp * Mp); k.IrL = planes(Mp * Mp); k.IcL = planes(Mp * Mp); k.TpL = planes(Mp * Mp);
  k.Rc = planes(Np * Np); k.IrR = planes(Np * Np); k.IcR = planes(Np * Np); k.TpR = planes(Np * Np);
  k.DXp = planes(Mp * Np); k.X1p = planes(Mp * Np);
  k.InvL = reinterpret_cast<float*>(take((int64_t)M * M * 4)); k.TfL = reinterpret_cast<float*>(take((int64_t)M * M * 4));
  k.InvR = reinterpret_cast<float*>(take((int64_t)N * N * 4)); k.TfR = reinterpret_cast<float*>(take((int64_t)N * N * 4));
  k.Pa = planes(Mp * Np); k.Pb = planes(Mp * Np);
  k.total = off;
  return k;
}
int64_t kron_inv_solves_bytes(int M, int N) {           // (the shape alone: a workspace sized once stays valid whatever the keys)
  return (kron_inv_route(M, N) && M <= 8192 && N <= 8192) ? inv_solve_layout(nullptr, M, N).total : 0;
}
bool kron_inv_solves_on(int M, int N) {
  return g_trsm_inv && g_planes && g_gemm_x3 && g_planes_f16 > 1 && kron_inv_solves_bytes(M, N) > 0;
}

int kron_inv_prepare(void* ws, int M, int N, hipStream_t main) {
  const InvSolveWs k = inv_solve_layout(static_cast<char*>(ws), M, N);
  return hipMemsetAsync(k.pm, 0, kPmSlots * sizeof(PlaneMeta), main) != hipSuccess;
}

// h of the blocked solves.  The pieces' PlaneMeta slots (kPmPieceX/W/V: the strips' range) hold at most FOUR block columns per side:
// a key-24 value that would make more (h = 512 at 4096: the slots overlap; h <= 256: they run into kPmInvR .. kPmX1) falls back to
// 2048, which is <= 4 blocks for every shape this route takes (M, N <= 8192).
static inline int inv_blk(int M, int N) {
  if (g_inv_blk <= 0) return 1 << 30;
  const int n = M > N ? M : N;
  return ((n + g_inv_blk - 1) / g_inv_blk > 4) ? 2048 : g_inv_blk;
}
static BlkSolve inv_solve_problem(const InvSolveWs& k, const float* QlS, const float* QrS, const float* dinv_r, const float* dinv_l,
                                  const float* X0, float* X1, float* Bt, int M, int N) {
  const long Mp = pad128(M), Np = pad128(N);
  PlaneMeta* pm = k.pm;
  BlkSolve s = {};
  s.M = M; s.N = N; s.h = inv_blk(M, N); s.pm = pm; s.part = k.part;
  s.L = InvSide{QlS, M, dinv_l, k.InvL, k.TfL, P3Buf{k.Lc, Mp, Mp, pm + kPmL}, P3Buf{k.IrL, Mp, Mp, pm + kPmInvL},
                P3Buf{k.IcL, Mp, Mp, pm + kPmInvL}, P3Buf{k.TpL, Mp, Mp, nullptr}, pm + kPmTL};
  s.R = InvSide{QrS, N, dinv_r, k.InvR, k.TfR, P3Buf{k.Rc, Np, Np, pm + kPmR}, P3Buf{k.IrR, Np, Np, pm + kPmInvR},
                P3Buf{k.IcR, Np, Np, pm + kPmInvR}, P3Buf{k.TpR, Np, Np, nullptr}, pm + kPmTR};
  s.X0 = X0; s.X0p = P3Buf{k.DXp, Mp, Np, pm + kPmdX};
  s.X1 = X1; s.Bt = Bt;
  s.pa = P3Buf{k.Pa, Mp, Np, nullptr}; s.pb = P3Buf{k.Pb, Mp, Np, nullptr};
  s.Br = P3Buf{nullptr, 0, 0, nullptr}; s.Bc = s.Br;
  if (kron_tile_scales(M, N) && s.h % 128 == 0) {
    // tile scales (see blk_solves_*): every piece's planes from the epilogue of the product that makes it.  Bt stays fp32 for this
    // caller (the bf16-operand update converts it); its column form -- the trailing updates' operand -- goes to Pa, free by then.
    s.L.Ir.te = k.te + kTeG1 * kTeTable; s.L.Ic.te = k.te + kTeIcL * kTeTable; s.L.Tp.te = k.te + kTeTpL * kTeTable;
    s.R.Ir.te = k.te + kTeG2 * kTeTable; s.R.Ic.te = k.te + kTeIcR * kTeTable; s.R.Tp.te = k.te + kTeTpR * kTeTable;
    s.L.Tp.meta = pm + kPmTL; s.R.Tp.meta = pm + kPmTR;
    s.pa.te = k.te + kTeY0 * kTeTable; s.pb.te = k.te + kTeY1 * kTeTable;
    s.x1c = P3Buf{k.X1p, Np, Mp, nullptr, nullptr, 0, k.te + kTeX1p * kTeTable};
    s.Bc = P3Buf{k.Pa, Np, Mp, nullptr, nullptr, 0, k.te + kTeY0 * kTeTable};
    s.bt_fp32 = true;
  }
  return s;
}

int kron_inv_solves_front(const float* QlS, const float* QrS, const float* dinv_r, const float* dinv_l, const float* X0, float* X1,
                          float* Bt, int M, int N, void* ws, hipStream_t main, hipStream_t side, hipEvent_t l_ready, bool maxima_ready,
                          int x0_parts, bool planes_ready, hipEvent_t x0_ready, hipStream_t x0_stream) {
  const InvSolveWs k = inv_solve_layout(static_cast<char*>(ws), M, N);
  int e;
  // column-form planes of the balanced factors (the B operand of T = A^-1 B, and of the blocked solves' trailing updates)
  BlkSolve s = inv_solve_problem(k, QlS, QrS, dinv_r, dinv_l, X0, X1, Bt, M, N);
  if (planes_ready) {                                  // (kron_balance_planes: the planes are there, at tile scales)
    s.R.Qc.te = k.te + kTeRc * kTeTable; s.L.Qc.te = k.te + kTeLc * kTeTable;
    s.X0p.te = k.te + kTeDXp * kTeTable;
    return blk_solves_front(s, main, side, l_ready, x0_stream ? x0_ready : nullptr, x0_stream);
  }
  if (maxima_ready) {                                  // (the balance launch left one partial maximum per workgroup of its grid)
    s.R.Qc.part = k.part + 2 * kPmPartMax; s.L.Qc.part = k.part + kPmPartMax;
    s.R.Qc.npart = s.L.Qc.npart = balance_grid(M, N);
  } else {
    if ((e = launch_absmax(QrS, (long)N * N, s.R.Qc, k.part + 2 * kPmPartMax, main))) return e;
    if ((e = launch_absmax(QlS, (long)M * M, s.L.Qc, k.part + kPmPartMax, side))) return e;
  }
  // (tri 2: the view is the transposed factor -- nonzero where k <= x; the other tiles become zeros without being read)
  if ((e = launch_split3(QrS, 1, N, N, N, s.R.Qc, main, SplitOpt{2, 0, 0, 0}))) return e;   // (x, k) = QrS[k][x]
  if ((e = launch_split3(QlS, 1, M, M, M, s.L.Qc, side, SplitOpt{2, 0, 0, 0}))) return e;
  if (x0_parts > 0) { s.X0p.part = k.part; s.X0p.npart = x0_parts; }
  return blk_solves_front(s, main, side, l_ready);
}

float* kron_inv_part(void* ws, int M, int N) { return inv_solve_layout(static_cast<char*>(ws), M, N).part; }
int kron_inv_part_max() { return kPmPartMax; }

int kron_inv_solves_back(const float* QlS, float* X1, float* Bt, int M, int N, void* ws, hipStream_t main, bool planes_ready) {
  const InvSolveWs k = inv_solve_layout(static_cast<char*>(ws), M, N);
  BlkSolve s = inv_solve_problem(k, QlS, nullptr, nullptr, nullptr, nullptr, X1, Bt, M, N);
  if (planes_ready) { s.R.Qc.te = k.te + kTeRc * kTeTable; s.L.Qc.te = k.te + kTeLc * kTeTable; }
  return blk_solves_back(s, main);
}

// entry points shared with psgd_kron_bf16.hip (kron_shared.h)
// part_l, part_r: balance_grid(M, N) partial maxima of |QlS|, |QrS| each; zero[0 .. nzero) is cleared (all optional)
static int kron_balance_amax(const float* Ql, const float* Qr, int M, int N, float* QlS, float* QrS, hipStream_t st, float* scal,
                             float* dinv, float* part_l, float* part_r, float* zero, int nzero) {
  const int grid = balance_grid(M, N);
  const int inv_blocks = ((M + 31) / 32 + (N + 31) / 32 + 3) / 4;
  hipLaunchKernelGGL(k_kron_balance_inv, dim3(inv_blocks + grid), dim3(kThreads), 0, st, Ql, Qr, M, N, QlS, QrS, scal, dinv,
                     inv_blocks, part_l, part_r, zero, nzero);
  return (int)hipGetLastError();
}
// rho + (balance, fp32 copies of the upper tiles, both plane forms with tile scales, the inverted 32-blocks) -- two launches
static int launch_balance_planes(const BalSide& L, const BalSide& R, float* part, float* scal, float* dinv, float* zero, int nzero,
                                 hipStream_t st, unsigned* zero2 = nullptr, int nzero2 = 0) {
  const int M = L.n, N = R.n;
  hipLaunchKernelGGL(k_kron_rho, dim3(kRhoBlocks), dim3(kThreads), 0, st, L.Q, R.Q, M, N, part, scal, zero, nzero, zero2, nzero2);
  const int TL = (M + 127) / 128, TR = (N + 127) / 128;
  const int tl = TL * (TL + 1) / 2, tr = TR * (TR + 1) / 2, zl = ((M + 511) / 512) * 6, zr = ((N + 511) / 512) * 6;
  const int inv_blocks = ((M + 31) / 32 + (N + 31) / 32 + 3) / 4;
  hipLaunchKernelGGL(k_kron_balance_planes, dim3(inv_blocks + tl + tr + zl + zr), dim3(kThreads), 0, st, L, R, part, dinv, inv_blocks,
                     tl, tr, zl);
  return (int)hipGetLastError();
}
static int kron_balance_planes(const float* Ql, const float* Qr, int M, int N, const KronWs& k, hipStream_t st, float* zero, int nzero) {
  float* part = k.pm_part + 2 * kPmPartMax;            // (the balance launch's partial maxima on the other route: 2 x kRhoBlocks words here)
  const BalSide L = {Ql, M, k.QlS, k.Lr, k.Lc, k.te + kTeLr * kTeTable, k.te + kTeLc * kTeTable};
  const BalSide R = {Qr, N, k.QrS, k.Rr, k.Rc, k.te + kTeRr * kTeTable, k.te + kTeRc * kTeTable};
  // (the arrival counters of the gradient grid's K-split tail: zeroed here instead of by a memset launch in front of that grid --
  //  9 us of fill kernel and a launch gap on the critical path)
  return launch_balance_planes(L, R, part, k.scal, k.dinv, zero, nzero, st, k.split_cnt, k.split_cnt ? 2 * kGradSplitMax : 0);
}

int kron_balance(const float* Ql, const float* Qr, int M, int N, float* QlS, float* QrS, hipStream_t st, float* scal,
                 float* dinv, void* inv_ws) {
  const long tot = (long)M * M + (long)N * N;
  int grid = (int)((tot + kThreads - 1) / kThreads);
  if (grid > 1024) grid = 1024;
  if (dinv && inv_ws) {
    const InvSolveWs k = inv_solve_layout(static_cast<char*>(inv_ws), M, N);
    return kron_balance_amax(Ql, Qr, M, N, QlS, QrS, st, scal, dinv, k.part + kPmPartMax, k.part + 2 * kPmPartMax,
                             reinterpret_cast<float*>(k.pm), kPmSlots * (int)(sizeof(PlaneMeta) / sizeof(float)));
  }
  if (dinv) {
    return kron_balance_amax(Ql, Qr, M, N, QlS, QrS, st, scal, dinv, nullptr, nullptr, nullptr, 0);
  } else {
    hipLaunchKernelGGL(k_kron_balance, dim3(grid), dim3(kThreads), 0, st, Ql, Qr, M, N, QlS, QrS, scal);
  }
  return (int)hipGetLastError();
}

bool kron_fused_prologue_on(int M, int N) {
  return g_fused_prologue && kron_inv_solves_on(M, N) && kron_tile_scales(M, N) && inv_blk(M, N) % 128 == 0;
}
int kron_balance_planes(const float* Ql, const float* Qr, int M, int N, float* QlS, float* QrS, hipStream_t st, float* dinv, void* inv_ws,
                        float* scal) {
  const InvSolveWs k = inv_solve_layout(static_cast<char*>(inv_ws), M, N);
  const BalSide L = {Ql, M, QlS, nullptr, k.Lc, nullptr, k.te + kTeLc * kTeTable};
  const BalSide R = {Qr, N, QrS, nullptr, k.Rc, nullptr, k.te + kTeRc * kTeTable};
  return launch_balance_planes(L, R, k.part + 2 * kPmPartMax, scal, dinv, reinterpret_cast<float*>(k.pm),
                               kPmSlots * (int)(sizeof(PlaneMeta) / sizeof(float)), st);
}

int kron_trsm_ut(const float* Q, int n, const float* X, float* Y, int nvec, long si, long sj, float* dinv, hipStream_t st,
                 int lite, bool inv_ready) {
  return trsm_ut(Q, n, X, Y, nvec, si, sj, dinv, st, 0, 0, lite, nullptr, nullptr, inv_ready);
}

static int g_overlap = 1;       // tuning key 9: 0 = the two chains of a large update run one after the other on the caller's stream
bool kron_overlap_chains(int M, int N) { return g_overlap != 0 && (M > 512 || N > 512); }
constexpr int g_side_prio = 1;     // (frozen in round 4, was tuning key 10) (before the first forked call on a stream): 0 = side streams at the lowest priority,
                                // 1 = at the default priority (default), 2 = at the highest.  A lowest- (or highest-) priority
                                // stream CREATED after an RCCL communicator has existed in the process runs the forked update up
                                // to 2x slower than the serial order (1024^2: 0.87 vs 0.47 ms); a default-priority one is as fast
                                // as the lowest-priority one otherwise and does not care (profiles/r03_rccl_fork_probe.txt)

KronFork* kron_fork(hipStream_t main) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, KronFork> tab;
  int cur = 0, dev = 0;
  if (hipGetDevice(&cur) != hipSuccess) return nullptr;
  dev = cur;
  if (main) {                                           // the side stream belongs to the device of the caller's stream
    hipDevice_t d;
    if (hipStreamGetDevice(main, &d) != hipSuccess) return nullptr;
    dev = (int)d;
  }
  KronFork* f = nullptr;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = tab.find({dev, main});
    if (it == tab.end()) {
      // no stream or event is created while the caller's stream is being captured: that first call stays serial
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (main && (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) return nullptr;
      KronFork n = {};
      int least = 0, greatest = 0;
      if (dev != cur && hipSetDevice(dev) != hipSuccess) return nullptr;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      const int prio = g_side_prio == 2 ? greatest : g_side_prio == 1 ? 0 : least;
      const bool ok = hipStreamCreateWithPriority(&n.side, hipStreamNonBlocking, prio) == hipSuccess &&
                      hipEventCreateWithFlags(&n.fork, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&n.join, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&n.mid, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&n.aux, hipEventDisableTiming) == hipSuccess &&
                      hipStreamCreateWithPriority(&n.bg, hipStreamNonBlocking, prio) == hipSuccess &&
                      hipEventCreateWithFlags(&n.bg_done, hipEventDisableTiming) == hipSuccess;
      if (dev != cur) (void)hipSetDevice(cur);
      if (!ok) return nullptr;
      it = tab.emplace(std::make_pair(dev, main), n).first;
    }
    f = &it->second;
  }
  if (hipEventRecord(f->fork, main) != hipSuccess || hipStreamWaitEvent(f->side, f->fork, 0) != hipSuccess) return nullptr;
  f->bg_live = 0;
  return f;
}

int kron_fork_bg(KronFork* f) {
  if (!f || !f->bg || hipStreamWaitEvent(f->bg, f->fork, 0) != hipSuccess) return 1;
  f->bg_live = 1;
  return 0;
}

// (Round 4, measured and not kept: CU-masked streams, hipExtStreamCreateWithCUMask -- the two inversion chains on c CUs each, the
//  products of :173 beside them on the other 256 - 2c.  The chains are launch-bound on the whole chip but WORK-bound on a slice of it
//  (k_tri_inv512 alone is 256 workgroups): 4096^2 fp32 update 2.56 -> 3.99 / 3.52 / 3.66 ms at c = 16 / 32 / 64; and masked streams
//  are blocking streams, so with the legacy default stream as the caller's they serialise against it: 5.1-5.4 ms.
//  profiles/r04_cumask_ab.txt)
int kron_join(KronFork* f, hipStream_t main) {
  if (f->bg_live) {
    f->bg_live = 0;
    if (hipEventRecord(f->bg_done, f->bg) != hipSuccess || hipStreamWaitEvent(main, f->bg_done, 0) != hipSuccess) return 1;
  }
  if (hipEventRecord(f->join, f->side) != hipSuccess) return 1;
  return hipStreamWaitEvent(main, f->join, 0) != hipSuccess;
}

}  // namespace psgdk

using namespace psgdk;

#define KRON_LAUNCH(expr)                     \
  do {                                        \
    if ((expr) != 0) return PSGD_ERR_LAUNCH;  \
  } while (0)

static int kron_ws_check(void* ws, int64_t ws_bytes, int64_t need) {
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < need) return PSGD_ERR_WORKSPACE;
  return PSGD_OK;
}

extern "C" {

int psgd_kron_set_tuning(int key, int value) {
  if (key == 0) { g_force_gemm = value; return PSGD_OK; }
  if (key == 1) { g_gemm_x3 = value; return PSGD_OK; }
  if (key == 2) { g_trsm_lds = value; return PSGD_OK; }
  if (key == 3) { g_small_deep = value; return PSGD_OK; }
  if (key == 4) { g_planes = value; return PSGD_OK; }
  if (key == 6) { g_grad_split = value; return PSGD_OK; }
  if (key == 7) { g_stage_mix = value; return PSGD_OK; }
  if (key == 9) { g_overlap = value; return PSGD_OK; }
  if (key == 11) { g_trsm_inv = value; return PSGD_OK; }
  if (key == 12) { g_planes_f16 = value; return PSGD_OK; }
  if (key == 16) { g_planes_exact = value; return PSGD_OK; }
  if (key == 23) { g_inv_strip512 = value; return PSGD_OK; }
  if (key == 24) { g_inv_blk = value; return PSGD_OK; }
  if (key == 25) { g_inv_order = value; return PSGD_OK; }
  if (key == 27) { g_pair_order = value; return PSGD_OK; }
  if (key == 28) { g_tile_scales = value; return PSGD_OK; }
  if (key == 29) { g_x0_side = value; return PSGD_OK; }
  if (key == 30) { g_bg_front = value; return PSGD_OK; }
  if (key == 31) { g_fused_prologue = value; return PSGD_OK; }
  if (key == 32) { g_bg_planes = value; return PSGD_OK; }
  if (key == 33) { g_grad_rect = value; return PSGD_OK; }
  if (key == 34) { g_solve_split = value; return PSGD_OK; }
  return PSGD_ERR_BAD_ARG;
}

int64_t psgd_kron_dd_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  return kron_layout(nullptr, M, N).total;
}

/* Factor-only half of the apply: the Gram(s) of the factors into the workspace (see plan_apply). */
int psgd_kron_dd_prepare_f32(const float* Ql, const float* Qr, int M, int N, void* ws, int64_t ws_bytes, void* stream) {
  if (!Ql || !Qr) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0) return PSGD_ERR_SHAPE;
  if (kron_ws_check(ws, ws_bytes, kron_layout(nullptr, M, N).total)) return PSGD_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KronWs k = kron_layout(static_cast<char*>(ws), M, N);
  if (kron_planes_apply(M, N) && g_planes && g_gemm_x3) {
    KRON_LAUNCH(planes_prepare(Ql, Qr, M, N, k, st));
    return PSGD_OK;
  }
  GemmArgs pre[2], app[3];
  int np = 0, na = 0;
  plan_apply(Ql, Qr, Ql /*unused*/, k.T /*unused*/, M, N, k, pre, np, app, na);
  if (np == 2) KRON_LAUNCH(launch_gram_batch(pre, 2, st));
  else KRON_LAUNCH(launch_gemm(pre[0], st));
  return PSGD_OK;
}

/* Gradient-dependent half: needs the Grams psgd_kron_dd_prepare_f32 left in `ws` for these very factors. */
int psgd_kron_dd_apply_prepared_f32(const float* Ql, const float* Qr, const float* G, float* out, int M, int N, void* ws,
                                    int64_t ws_bytes, void* stream) {
  if (!Ql || !Qr || !G || !out) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0) return PSGD_ERR_SHAPE;
  if (kron_ws_check(ws, ws_bytes, kron_layout(nullptr, M, N).total)) return PSGD_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KronWs k = kron_layout(static_cast<char*>(ws), M, N);
  if (kron_planes_apply(M, N) && g_planes && g_gemm_x3) {
    KRON_LAUNCH(planes_apply(G, out, M, N, k, st));
    return PSGD_OK;
  }
  GemmArgs pre[2], app[3];
  int np = 0, na = 0;
  plan_apply(Ql, Qr, G, out, M, N, k, pre, np, app, na);
  for (int i = 0; i < na; ++i) KRON_LAUNCH(launch_gemm(app[i], st));
  return PSGD_OK;
}

int psgd_kron_dd_apply_f32(const float* Ql, const float* Qr, const float* G, float* out, int M, int N, void* ws,
                           int64_t ws_bytes, void* stream) {
  const int rc = psgd_kron_dd_prepare_f32(Ql, Qr, M, N, ws, ws_bytes, stream);
  if (rc) return rc;
  return psgd_kron_dd_apply_prepared_f32(Ql, Qr, G, out, M, N, ws, ws_bytes, stream);
}

/* The apply for factors that are new on every call: no Gram, nothing prepared (planes_apply_direct) where the f16 x 2 plane products
 * apply; the same as psgd_kron_dd_apply_f32 elsewhere.  Leaves no prepared state: psgd_kron_dd_apply_prepared_f32 needs a
 * psgd_kron_dd_prepare_f32 (or psgd_kron_dd_apply_f32) for these factors first. */
static bool apply_direct_is_distinct(int M, int N) {
  return kron_planes_apply(M, N) && g_planes && g_gemm_x3 && g_planes_f16 && g_planes_exact;
}
/* 1 when psgd_kron_dd_apply_direct_f32 is a path of its own for this shape under the current tuning (else it is psgd_kron_dd_apply_f32,
 * which leaves prepared state) */
int psgd_kron_dd_apply_direct_distinct(int M, int N) { return (M > 0 && N > 0 && apply_direct_is_distinct(M, N)) ? 1 : 0; }
int psgd_kron_dd_apply_direct_f32(const float* Ql, const float* Qr, const float* G, float* out, int M, int N, void* ws,
                                  int64_t ws_bytes, void* stream) {
  if (!Ql || !Qr || !G || !out) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0) return PSGD_ERR_SHAPE;
  if (kron_ws_check(ws, ws_bytes, kron_layout(nullptr, M, N).total)) return PSGD_ERR_WORKSPACE;
  KronWs k = kron_layout(static_cast<char*>(ws), M, N);
  if (apply_direct_is_distinct(M, N) && k.pmeta) {
    KRON_LAUNCH(planes_apply_direct(Ql, Qr, G, out, M, N, k, static_cast<hipStream_t>(stream)));
    return PSGD_OK;
  }
  return psgd_kron_dd_apply_f32(Ql, Qr, G, out, M, N, ws, ws_bytes, stream);
}

int psgd_kron_dd_update_f32(const float* Ql, const float* Qr, const float* dX, const float* dG, float* QlOut,
                            float* QrOut, int M, int N, float step, float tiny, void* ws, int64_t ws_bytes,
                            void* stream) {
  if (!Ql || !Qr || !dX || !dG || !QlOut || !QrOut) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0) return PSGD_ERR_SHAPE;
  if (kron_ws_check(ws, ws_bytes, kron_layout(nullptr, M, N).total)) return PSGD_ERR_WORKSPACE;
  // small layers are launch-bound: the batch-of-one route with half the launches of the large-layer path (stages of independent chains share them)
  if (M <= 512 && N <= 512 && (g_stage_mix & 2))
    return psgd_kron_dd_update_batched_f32(&Ql, &Qr, &dX, &dG, &QlOut, &QrOut, &M, &N, 1, step, tiny, ws, ws_bytes, stream);
  hipStream_t st = static_cast<hipStream_t>(stream);
  KronWs k = kron_layout(static_cast<char*>(ws), M, N);
  const bool planes = kron_planes(M, N) && g_planes && g_gemm_x3;
  PlaneMeta* pm = (planes && g_planes_f16 > 1) ? k.pmeta : nullptr;        // f16 x 2 planes of the update
  // K0: balance (:166-170); zeroes k.scal (and the update's plane maxima); the same launch inverts the diagonal blocks the
  // solves of K2 start from
  const bool inv_route = pm && g_trsm_inv && kron_inv_route(M, N) && M <= 8192 && N <= 8192;    // (6 levels of meta slots)
  // (round 6) on the tile-scale route of the inverse solves the whole prologue is rho + ONE sweep (k_kron_balance_planes)
  k.factor_ts = inv_route && g_fused_prologue && kron_tile_scales(M, N) && inv_blk(M, N) % 128 == 0;
  if (k.factor_ts)
    KRON_LAUNCH(kron_balance_planes(Ql, Qr, M, N, k, st, &pm[kPmL].scale, (kPmSlots - kPmL) * 4));
  else
    KRON_LAUNCH(kron_balance_amax(Ql, Qr, M, N, k.QlS, k.QrS, st, k.scal, k.dinv, pm ? k.pm_part + 2 * kPmPartMax : nullptr,
                                  pm ? k.pm_part + 3 * kPmPartMax : nullptr, pm ? &pm[kPmL].scale : nullptr,
                                  pm ? (kPmSlots - kPmL) * 4 : 0));
  float* dinv_l = k.dinv + (long)((N + 31) / 32) * 1024;
  GemmArgs s[6];
  plan_update(dG, QlOut, QrOut, M, N, step, tiny, k, s);
  // the products (:173) go to the side stream, the solves (:174) stay here; they meet at the gradient products.  The
  // factors' planes belong to the product chain unless the solves read them too (their K = 2048 group products, which
  // exist from 4096 on -- or from 2048 on with tuning key 5): then they are made before the fork.
  const bool solves_on_planes = planes && (inv_route || M > g_trsm_planes_min_n || N > g_trsm_planes_min_n);
  if (solves_on_planes && !k.factor_ts) KRON_LAUNCH(planes_update_factors(M, N, k, st, pm));
  KronFork* fk = kron_overlap_chains(M, N) ? kron_fork(st) : nullptr;
  KronForkScope fork_scope(fk, st);          // joins on every exit path, early error returns included
  hipStream_t sf = fk ? fk->side : st;
  if (inv_route) {
    // K2 through explicit inverses (tri_inverse): Qr's on this stream, then X1 = dX Ri; Ql's on the side stream ahead of the
    // products of :173; after the join Bt = Li' X1, whose epilogue leaves max|Bt| for the planes of the gradient products.
    const long Mp = pad128(M), Np = pad128(N);
    P3Buf Lc = {k.Lc, Mp, Mp, pm + kPmL}, Rc = {k.Rc, Np, Np, pm + kPmR};
    if (k.factor_ts) { Lc.te = k.te + kTeLc * kTeTable; Rc.te = k.te + kTeRc * kTeTable; }
    InvSide L = {k.QlS, M, dinv_l, k.g1, k.TfL, Lc, P3Buf{k.G1, Mp, Mp, pm + kPmInvL},
                 P3Buf{k.IcL, Mp, Mp, pm + kPmInvL}, P3Buf{k.TpL, Mp, Mp, nullptr}, pm + kPmTL};
    InvSide R = {k.QrS, N, k.dinv, k.g2, k.TfR, Rc, P3Buf{k.G2, Np, Np, pm + kPmInvR},
                 P3Buf{k.IcR, Np, Np, pm + kPmInvR}, P3Buf{k.TpR, Np, Np, nullptr}, pm + kPmTR};
    // Order: the full-chip products of :173 run beside the launch-bound lower levels of Qr's inversion, Ql's inversion beside the
    // products of X1 = dX R^-1 (with the two inversions first and the products colliding afterwards the join came 0.3 ms later).
    const bool inv_first = kron_inv_first(M, N) && fk != nullptr;
    if (!inv_first) KRON_LAUNCH(planes_update_front(dG, M, N, k, sf, pm));
    BlkSolve bs = {};
    bs.M = M; bs.N = N; bs.h = inv_blk(M, N); bs.pm = pm; bs.part = k.pm_part;
    bs.L = L; bs.R = R;
    bs.X0 = dX; bs.X0p = P3Buf{k.DXp, Mp, Np, pm + kPmdX};
    bs.X1 = k.X1; bs.Bt = k.Bt;
    bs.pa = P3Buf{k.Y0, Mp, Np, nullptr}; bs.pb = P3Buf{k.Y1, Mp, Np, nullptr};      // (transients of the apply: free during an update)
    bs.Br = P3Buf{k.U0, Mp, Np, pm + kPmBt}; bs.Bc = P3Buf{k.U1, Np, Mp, pm + kPmBt};
    const bool ts = kron_tile_scales(M, N) && bs.h % 128 == 0;
    if (ts) {
      bs.L.Ir.te = k.te + kTeG1 * kTeTable; bs.L.Ic.te = k.te + kTeIcL * kTeTable; bs.L.Tp.te = k.te + kTeTpL * kTeTable;
      bs.R.Ir.te = k.te + kTeG2 * kTeTable; bs.R.Ic.te = k.te + kTeIcR * kTeTable; bs.R.Tp.te = k.te + kTeTpR * kTeTable;
      bs.L.Tp.meta = pm + kPmTL; bs.R.Tp.meta = pm + kPmTR;        // (fmt of p3_args: any meta)
      // tile scales: X1's column form lives in X1p; Bt's planes go to Y0 / Y1 (the pieces' buffers of the right solve: free by then),
      // whichever stream order -- U0 / U1 stay with the products of :173
      bs.pa.te = k.te + kTeY0 * kTeTable; bs.pb.te = k.te + kTeY1 * kTeTable;
      bs.x1c = P3Buf{k.X1p, Np, Mp, nullptr, nullptr, 0, k.te + kTeX1p * kTeTable};
      if (k.factor_ts) bs.X0p.te = k.te + kTeDXp * kTeTable;
      if (k.factor_ts && k.split_cnt) {          // (nothing writes k.T on this route: partial tiles of the solves' split products)
        bs.sk_scratch = k.T; bs.sk_slots = (int)std::min<int64_t>((int64_t)M * N / (64 * kThreads), 1 << 20);
        bs.sk_cnt = k.split_cnt + kGradSplitMax;
      }
      bs.bt_fp32 = false;
      if (!inv_first) KRON_LAUNCH(blk_solves_front(bs, st, sf));
      else if (kron_bg_front(M, N) && fk->bg) {
        // third stream: dX's planes, dG's planes, the two products; the side stream has Ql's inversion alone, from the fork point on (behind
        // dX's planes its first launch -- 256 workgroups -- met the first product and took 449 us instead of 63: Bt then waited for it)
        KRON_LAUNCH(kron_fork_bg(fk));
        KRON_LAUNCH(blk_solves_front(bs, st, sf, fk->mid, g_x0_side ? fk->aux : nullptr, fk->bg));
        KRON_LAUNCH(planes_update_front(dG, M, N, k, fk->bg, pm));
        if (hipStreamWaitEvent(st, fk->mid, 0) != hipSuccess) return PSGD_ERR_LAUNCH;
      } else if (g_bg_planes && k.factor_ts && g_x0_side && fk->bg) {
        // the input planes (one sweep each) on the third stream; the side stream: Ql's inversion from the fork point on, then the products
        KRON_LAUNCH(kron_fork_bg(fk));
        KRON_LAUNCH(blk_solves_front(bs, st, sf, fk->mid, fk->aux, fk->bg));
        KRON_LAUNCH(planes_update_front(dG, M, N, k, fk->bg, pm, 1));
        if (hipEventRecord(fk->bg_done, fk->bg) != hipSuccess || hipStreamWaitEvent(sf, fk->bg_done, 0) != hipSuccess) return PSGD_ERR_LAUNCH;
        KRON_LAUNCH(planes_update_front(dG, M, N, k, sf, pm, 2));
        if (hipStreamWaitEvent(st, fk->mid, 0) != hipSuccess) return PSGD_ERR_LAUNCH;      // (Ql's inverse, for Bt)
      } else {
        KRON_LAUNCH(blk_solves_front(bs, st, sf, fk->mid, g_x0_side ? fk->aux : nullptr));
        KRON_LAUNCH(planes_update_front(dG, M, N, k, sf, pm));
        if (hipStreamWaitEvent(st, fk->mid, 0) != hipSuccess) return PSGD_ERR_LAUNCH;
      }
      bs.Br = P3Buf{k.Y0, Mp, Np, nullptr, nullptr, 0, k.te + kTeY0 * kTeTable};
      bs.Bc = P3Buf{k.Y1, Np, Mp, nullptr, nullptr, 0, k.te + kTeY1 * kTeTable};
      if (!inv_first) KRON_LAUNCH(fork_scope.join());                 // (Ql's inverse is made on the side stream)
      KRON_LAUNCH(blk_solves_back(bs, st));
      if (inv_first) KRON_LAUNCH(fork_scope.join());
      KRON_LAUNCH(planes_update_back(QlOut, QrOut, M, N, step, tiny, k, st, pm, true, &bs.Br, &bs.Bc));
      return PSGD_OK;
    }
    if (inv_first) {
      // both inversions first, undisturbed by full-chip products; the products of :173 then run on the side stream beside X1 and Bt
      KRON_LAUNCH(blk_solves_front(bs, st, sf, fk->mid));
      KRON_LAUNCH(planes_update_front(dG, M, N, k, sf, pm));
      if (hipStreamWaitEvent(st, fk->mid, 0) != hipSuccess) return PSGD_ERR_LAUNCH;
      P3Buf br = bs.Br, bc = bs.Bc;                               // (U0 / U1: dG's chain on the side stream is still using them)
      bs.Br.p = nullptr;
      KRON_LAUNCH(blk_solves_back(bs, st));
      KRON_LAUNCH(fork_scope.join());
      br.part = bc.part = &pm[kPmBt].amax; br.npart = bc.npart = 1;
      KRON_LAUNCH(launch_split3_both(k.Bt, N, 1, M, N, br, bc, st));
    } else {
      KRON_LAUNCH(blk_solves_front(bs, st, sf));
      KRON_LAUNCH(fork_scope.join());
      KRON_LAUNCH(blk_solves_back(bs, st));                       // (leaves max|Bt| and the planes of Bt for the gradient products)
    }
    KRON_LAUNCH(planes_update_back(QlOut, QrOut, M, N, step, tiny, k, st, pm, true));
    return PSGD_OK;
  }
  if (planes) {
    if (!solves_on_planes) KRON_LAUNCH(planes_update_factors(M, N, k, sf, pm));
    KRON_LAUNCH(planes_update_front(dG, M, N, k, sf, pm));
  } else {
    KRON_LAUNCH(launch_gemm(s[0], sf));
    KRON_LAUNCH(launch_gemm(s[1], sf));
  }
  // K2 (:174): X1 = dX QrS^-1 (rows independent), Bt = QlS^-T X1 (columns independent)
  // (the K = 512 trailing products of the solves were tried on planes too: 557 + 7 x 9 us of strip splits against 647 us
  // per solve -- their 16 K steps per block are dominated by the fixed parts of a block either way -- so only the wide
  // K = 2048 group updates use the factors' column-form planes)
  if (solves_on_planes) {
    const P3Buf Rc = {k.Rc, pad128(N), pad128(N), pm ? pm + kPmR : pm}, Lc = {k.Lc, pad128(M), pad128(M), pm ? pm + kPmL : pm};
    constexpr int half = (kPmStripEnd - kPmStrip) / 2;                 // meta slots of the groups of either solve
    KRON_LAUNCH(trsm_ut(k.QrS, N, dX, k.X1, M, (long)N, 1L, k.dinv, st, 0, 0, 0, &Rc, k.S0, true, pm ? pm + kPmStrip : pm, half,
                        k.pm_part));
    KRON_LAUNCH(trsm_ut(k.QlS, M, k.X1, k.Bt, N, 1L, (long)N, dinv_l, st, 0, 0, 0, &Lc, k.S0, true,
                        pm ? pm + kPmStrip + half : pm, half, k.pm_part));
  } else {
    KRON_LAUNCH(trsm_ut(k.QrS, N, dX, k.X1, M, (long)N, 1L, k.dinv, st, 0, 0, 0, nullptr, nullptr, true));
    KRON_LAUNCH(trsm_ut(k.QlS, M, k.X1, k.Bt, N, 1L, (long)N, dinv_l, st, 0, 0, 0, nullptr, nullptr, true));
  }
  KRON_LAUNCH(fork_scope.join());
  if (planes) {
    KRON_LAUNCH(planes_update_back(QlOut, QrOut, M, N, step, tiny, k, st, pm));
    return PSGD_OK;
  }
  KRON_LAUNCH(launch_gemm_two(s[2], s[3], st));      // the two gradient products
  KRON_LAUNCH(launch_gemm_two(s[4], s[5], st));      // the two factor updates
  return PSGD_OK;
}

/* ---- batched forms: the same stage of every layer in one launch (LeNet5-size layers are
 * launch-bound: 4 launches per apply of the whole set instead of 4 per layer). ------------- */

int64_t psgd_kron_dd_workspace_bytes_batched(const int* M, const int* N, int count) {
  if (!M || !N || count <= 0) return PSGD_ERR_BAD_ARG;
  int64_t tot = 0;
  for (int p = 0; p < count; ++p) {
    if (M[p] <= 0 || N[p] <= 0) return PSGD_ERR_BAD_ARG;
    tot += kron_layout(nullptr, M[p], N[p]).total;
  }
  return tot;
}

/* mode: 1 = factor-only half (Grams), 2 = gradient half (needs the Grams of mode 1 in ws), 3 = both */
static int apply_batched_impl(const float* const* Ql, const float* const* Qr, const float* const* G, float* const* out,
                              const int* M, const int* N, int count, void* ws, int64_t ws_bytes, void* stream, int mode) {
  if (!Ql || !Qr || !M || !N || count <= 0) return PSGD_ERR_BAD_ARG;
  if ((mode & 2) && (!G || !out)) return PSGD_ERR_BAD_ARG;
  const int64_t need = psgd_kron_dd_workspace_bytes_batched(M, N, count);
  if (need < 0) return (int)need;
  if (kron_ws_check(ws, ws_bytes, need)) return PSGD_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(ws);
  for (int p0 = 0; p0 < count; p0 += kMaxLayers) {
    const int nb = (count - p0 < kMaxLayers) ? count - p0 : kMaxLayers;
    GemmArgs pre[2 * kMaxLayers], app[3][kMaxLayers];
    int npre = 0, na_max = 0, na[kMaxLayers];
    for (int q = 0; q < nb; ++q) {
      const int p = p0 + q;
      if (!Ql[p] || !Qr[p] || ((mode & 2) && (!G[p] || !out[p]))) return PSGD_ERR_BAD_ARG;
      KronWs k = kron_layout(base, M[p], N[p]);
      base += k.total;
      GemmArgs pr[2], ap[3];
      int np = 0;
      plan_apply(Ql[p], Qr[p], (mode & 2) ? G[p] : Ql[p], (mode & 2) ? out[p] : k.T, M[p], N[p], k, pr, np, ap, na[q]);
      for (int i = 0; i < np; ++i) pre[npre++] = pr[i];
      for (int i = 0; i < na[q]; ++i) app[i][q] = ap[i];
      if (na[q] > na_max) na_max = na[q];
    }
    if (mode & 1) KRON_LAUNCH(launch_gram_batch(pre, npre, st));
    if (mode & 2) {
      for (int stage = 0; stage < na_max; ++stage) {
        GemmArgs g[kMaxBatch];
        int ng = 0;
        for (int q = 0; q < nb; ++q)
          if (stage < na[q]) g[ng++] = app[stage][q];
        KRON_LAUNCH(launch_gemm_batch(g, ng, st));
      }
    }
  }
  return PSGD_OK;
}

int psgd_kron_dd_apply_batched_f32(const float* const* Ql, const float* const* Qr, const float* const* G,
                                   float* const* out, const int* M, const int* N, int count, void* ws,
                                   int64_t ws_bytes, void* stream) {
  return apply_batched_impl(Ql, Qr, G, out, M, N, count, ws, ws_bytes, stream, 3);
}

int psgd_kron_dd_prepare_batched_f32(const float* const* Ql, const float* const* Qr, const int* M, const int* N, int count,
                                     void* ws, int64_t ws_bytes, void* stream) {
  return apply_batched_impl(Ql, Qr, nullptr, nullptr, M, N, count, ws, ws_bytes, stream, 1);
}

int psgd_kron_dd_apply_prepared_batched_f32(const float* const* Ql, const float* const* Qr, const float* const* G,
                                            float* const* out, const int* M, const int* N, int count, void* ws,
                                            int64_t ws_bytes, void* stream) {
  return apply_batched_impl(Ql, Qr, G, out, M, N, count, ws, ws_bytes, stream, 2);
}

int psgd_kron_dd_update_batched_f32(const float* const* Ql, const float* const* Qr, const float* const* dX,
                                    const float* const* dG, float* const* QlOut, float* const* QrOut, const int* M,
                                    const int* N, int count, float step, float tiny, void* ws, int64_t ws_bytes,
                                    void* stream) {
  if (!Ql || !Qr || !dX || !dG || !QlOut || !QrOut || !M || !N || count <= 0) return PSGD_ERR_BAD_ARG;
  const int64_t need = psgd_kron_dd_workspace_bytes_batched(M, N, count);
  if (need < 0) return (int)need;
  if (kron_ws_check(ws, ws_bytes, need)) return PSGD_ERR_WORKSPACE;
  for (int p = 0; p < count; ++p)
    if (M[p] > 512 || N[p] > 512) return PSGD_ERR_SHAPE;   // batched form is for small layers
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(ws);
  for (int p0 = 0; p0 < count; p0 += kMaxLayers) {
    const int nb = (count - p0 < kMaxLayers) ? count - p0 : kMaxLayers;
    GemmArgs s[kMaxLayers][6];
    KronWs k[kMaxLayers];
    BalanceBatch bb;
    TrsmBatch t1, t2;
    bb.count = t1.count = t2.count = nb;
    int blk1 = 0, blk2 = 0, nmax = 0;
    for (int q = 0; q < nb; ++q) {
      const int p = p0 + q;
      if (!Ql[p] || !Qr[p] || !dX[p] || !dG[p] || !QlOut[p] || !QrOut[p]) return PSGD_ERR_BAD_ARG;
      k[q] = kron_layout(base, M[p], N[p]);
      base += k[q].total;
      plan_update(dG[p], QlOut[p], QrOut[p], M[p], N[p], step, tiny, k[q], s[q]);
      bb.Ql[q] = Ql[p]; bb.Qr[q] = Qr[p]; bb.QlS[q] = k[q].QlS; bb.QrS[q] = k[q].QrS; bb.M[q] = M[p]; bb.N[q] = N[p];
      bb.scal[q] = k[q].scal; bb.dinv[q] = k[q].dinv;
      t1.t[q] = {k[q].QrS, N[p], N[p], dX[p], k[q].X1, M[p], (long)N[p], 1L, 0L, 0L};
      t1.dinv[q] = k[q].dinv; t2.dinv[q] = k[q].dinv + (long)((N[p] + 31) / 32) * 1024;
      t2.t[q] = {k[q].QlS, M[p], M[p], k[q].X1, k[q].Bt, N[p], 1L, (long)N[p], 0L, 0L};
      blk1 += (M[p] + 15) / 16; t1.blk_end[q] = blk1;      // 16-vector strips (k_trsm_ut_inv_batched)
      blk2 += (N[p] + 15) / 16; t2.blk_end[q] = blk2;
      if (M[p] > nmax) nmax = M[p];
      if (N[p] > nmax) nmax = N[p];
    }
    // K0 (:166-170) and the inverted diagonal blocks of K2 (:174) of every layer; also zeroes the scratch words
    hipLaunchKernelGGL(k_kron_balance_batched, dim3(kBalInvBlocks + 64, nb), dim3(kThreads), 0, st, bb);
    KRON_LAUNCH((int)hipGetLastError());
    GemmArgs g[kMaxBatch];
    long t64 = 0;
    for (int q = 0; q < nb; ++q) t64 += (long)((N[p0 + q] + 63) / 64) * ((M[p0 + q] + 63) / 64);
    const bool mixed = (g_stage_mix & 1) && t64 < 96 && g_small_deep && !g_trsm_lds;    // the kernels launch_gemm_batch / the solve would pick
    if (mixed) {
      // the products (:173) next to the solves (:174), stage by stage (their diagonal blocks were inverted by the first launch)
      for (int stage = 0; stage < 2; ++stage) {
        MixBatch mb;
        const TrsmBatch& tb = stage ? t2 : t1;
        mb.count = nb; mb.strips = stage ? blk2 : blk1;
        int tiles = 0;
        for (int q = 0; q < nb; ++q) {
          mb.blk_end[q] = tb.blk_end[q]; mb.t[q] = tb.t[q]; mb.dinv[q] = tb.dinv[q];
          mb.g[q] = s[q][stage];
          tiles += ((mb.g[q].N + 31) / 32) * ((mb.g[q].M + 31) / 32);
          mb.tile_end[q] = tiles;
        }
        hipLaunchKernelGGL(k_small_stage_mixed, dim3(mb.strips + tiles), dim3(kThreads), 0, st, mb);
        KRON_LAUNCH((int)hipGetLastError());
      }
    }
    for (int stage = 0; stage < 2 && !mixed; ++stage) {
      for (int q = 0; q < nb; ++q) g[q] = s[q][stage];
      KRON_LAUNCH(launch_gemm_batch(g, nb, st));
    }
    // K2 (:174) for every layer: the two solves (their diagonal blocks were inverted by the first launch)
    for (int pass = 0; pass < 2 && !mixed; ++pass) {
      TrsmBatch& tb = pass ? t2 : t1;
      if (!g_trsm_lds) {
        hipLaunchKernelGGL(k_trsm_ut_reg_batched, dim3(pass ? blk2 : blk1), dim3(kThreads), 0, st, tb);
        KRON_LAUNCH((int)hipGetLastError());
        continue;
      }
      const int pitch_max = ((nmax + 31) & ~31) + 2;
      static DeviceOnce attr_set;
      if (attr_set.needed()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trsm_ut_inv_batched),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((64 * (kStripN + 2) + 32 * 33) * sizeof(float))) != hipSuccess)
          return PSGD_ERR_LAUNCH;
        attr_set.done();
      }
      hipLaunchKernelGGL(k_trsm_ut_inv_batched, dim3(pass ? blk2 : blk1), dim3(kThreads),
                         (size_t)(16 * pitch_max + 32 * 33) * sizeof(float), st, tb, pitch_max);
      KRON_LAUNCH((int)hipGetLastError());
    }
    // the two gradient products of a layer are independent of each other, and so are its two factor updates: each pair
    // of stages is one launch over 2 nb problems
    for (int stage = 2; stage < 6; stage += 2) {
      for (int q = 0; q < nb; ++q) { g[2 * q] = s[q][stage]; g[2 * q + 1] = s[q][stage + 1]; }
      KRON_LAUNCH(launch_gemm_batch(g, 2 * nb, st));
    }
  }
  return PSGD_OK;
}

/* ---- sparse Kronecker formats (psgd.py:198-391): canonical (left, right) orientations ----
 * fmt 0: (dense, scaling)  Ql [M,M], qr [N]          psgd.py:276-322
 * fmt 1: (normalization, dense)  ql [2,M], Qr [N,N]  psgd.py:198-270
 * fmt 2: (normalization, scaling) ql [2,M], qr [N]   psgd.py:328-391
 * Data matrices are strided views (element (m,n) at p[m*rs + n*cs]); results are contiguous. */

struct SparseWs {
  float *scal, *LS, *RS, *T, *A, *Bt, *gsq, *v0, *v1, *v2, *v3, *dinv, *cpart;
  __bf16 *P0, *P1;              // operand planes of the dense factor's gradient when it is a few tiles with a long K
  float* sk_scratch; unsigned* sk_cnt;
  // f16 x 2 operand planes for the data-sized products (sparse_gemm): two sets (caller's stream, side stream) of an A and a B buffer
  __bf16* GP[2][2]; PlaneMeta* gmeta; float* gpart; int64_t gcap;
  // the dense factor's solve through its explicit inverse when there are many vectors (sparse_solve): planes and fp32 of tri_inverse
  __bf16 *IQc, *IIr, *IIc, *ITp; float *IInv, *ITf; PlaneMeta* imeta;
  int64_t total;
};

// the dense factor's gradient triu(X X' - Y Y') [rows x rows, K]: split-K on planes when it has few tiles and a long K
constexpr int kSplitkMaxTiles = 160, kSplitkScratchTiles = 640;
static inline bool sparse_splitk(int rows, int K) {
  const long T = (rows + 127) / 128;
  return T * (T + 1) / 2 <= kSplitkMaxTiles && K >= 4096;
}

static SparseWs sparse_layout(char* base, int fmt, int M, int N) {
  SparseWs k;
  const int64_t mn = (int64_t)M * N * 4;
  const int64_t lbytes = (fmt == 0) ? (int64_t)M * M * 4 : (int64_t)2 * M * 4;
  const int64_t rbytes = (fmt == 1) ? (int64_t)N * N * 4 : (int64_t)N * 4;
  const int64_t gbytes = (fmt == 0) ? (int64_t)M * M * 4 : (fmt == 1 ? (int64_t)N * N * 4 : 256);
  const int64_t vb = (int64_t)(M > N ? M : N) * 4;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { float* p = reinterpret_cast<float*>(base + off); off = align256(off + bytes); return p; };
  k.scal = take(256);
  k.LS = take(lbytes); k.RS = take(rbytes);
  k.T = take(mn); k.A = take(mn); k.Bt = take(mn);
  k.gsq = take(gbytes);
  k.v0 = take(vb); k.v1 = take(vb); k.v2 = take(vb); k.v3 = take(vb);
  k.dinv = take((int64_t)(((M > N ? M : N) + 31) / 32) * 1024 * 4);
  k.cpart = take((int64_t)kColRedRowBlocks * N * 4);             // row-block partials of k_col_reduce
  k.P0 = k.P1 = nullptr; k.sk_scratch = nullptr; k.sk_cnt = nullptr;
  const int grows = fmt == 0 ? M : N, gk = fmt == 0 ? N : M;     // the dense factor's gradient: [grows x grows], K = gk
  if (fmt != 2 && sparse_splitk(grows, gk)) {
    const int64_t pb = (int64_t)((grows + 127) & ~127) * ((gk + 127) & ~127) * 6;
    k.P0 = reinterpret_cast<__bf16*>(take(pb)); k.P1 = reinterpret_cast<__bf16*>(take(pb));
    k.sk_scratch = take((int64_t)kSplitkScratchTiles * 64 * kThreads * 4);
    k.sk_cnt = reinterpret_cast<unsigned*>(take(kSplitkMaxTiles * 4));
  }
  k.gmeta = nullptr; k.gpart = nullptr; k.gcap = 0;
  k.IQc = k.IIr = k.IIc = k.ITp = nullptr; k.IInv = k.ITf = nullptr; k.imeta = nullptr;
  k.GP[0][0] = k.GP[0][1] = k.GP[1][0] = k.GP[1][1] = nullptr;
  if (fmt != 2 && grows >= 512 && (int64_t)M * N >= (int64_t)1 << 20) {      // a dense factor worth the planes (sparse_gemm)
    const int64_t Mp = pad128(M), Np = pad128(N), dp = pad128(grows);
    k.gcap = Mp * Np > dp * dp ? Mp * Np : dp * dp;
    for (int s = 0; s < 2; ++s)
      for (int o = 0; o < 2; ++o) k.GP[s][o] = reinterpret_cast<__bf16*>(take(k.gcap * 4));
    k.gmeta = reinterpret_cast<PlaneMeta*>(take(256));
    k.gpart = take(6 * kPmPartMax * 4);                     // (two per plane set, one for sparse_solve's factor)
    k.IQc = reinterpret_cast<__bf16*>(take(dp * dp * 4)); k.IIr = reinterpret_cast<__bf16*>(take(dp * dp * 4));
    k.IIc = reinterpret_cast<__bf16*>(take(dp * dp * 4)); k.ITp = reinterpret_cast<__bf16*>(take(dp * dp * 4));
    k.IInv = take((int64_t)grows * grows * 4); k.ITf = take((int64_t)grows * grows * 4);
    k.imeta = reinterpret_cast<PlaneMeta*>(take(256));
  }
  k.total = off;
  return k;
}

// column sums of an M x N operand (k_col_reduce + k_col_reduce_fin): enough row blocks to put ~2048 workgroups on the chip,
// at least 256 rows each
static int col_reduce(const SparseWs& k, MatView Z, MatView Z2, const float* q0, const float* q1, int M, int N, int mode,
                      float* out, hipStream_t st) {
  const int cb = (N + 63) / 64;
  int rb = (2048 + cb - 1) / cb;
  if (rb > kColRedRowBlocks) rb = kColRedRowBlocks;
  if (rb > (M + 255) / 256) rb = (M + 255) / 256;
  if (rb < 1) rb = 1;
  const int rows = (M + rb - 1) / rb;
  rb = (M + rows - 1) / rows;
  hipLaunchKernelGGL(k_col_reduce, dim3(cb, rb), dim3(kThreads), 0, st, Z, Z2, q0, q1, M, N, mode, rows, rb == 1 ? out : k.cpart);
  if (hipGetLastError() != hipSuccess) return PSGD_ERR_LAUNCH;
  if (rb > 1) {
    hipLaunchKernelGGL(k_col_reduce_fin, dim3((N + kThreads - 1) / kThreads), dim3(kThreads), 0, st, (const float*)k.cpart, rb, N, out);
    if (hipGetLastError() != hipSuccess) return PSGD_ERR_LAUNCH;
  }
  return PSGD_OK;
}

// grad = triu(X X' - Y Y') for X, Y given as (row, k) views [rows x K]; fp32 result + max|.| as the fused GEMM epilogue does
// amaxX / amaxY (optional, f16 x 2 planes): device words with max|X|, max|Y| that the producing products left (sparse_gemm's
// `track`); without them a reduction launch per operand finds the maxima.
static int sparse_grad_splitk(const SparseWs& k, const float* X, const float* Y, long rs, long cs, int rows, int K, float* C,
                              float* maxout, hipStream_t st, const float* amaxX = nullptr, const float* amaxY = nullptr) {
  const long rp = (rows + 127) & ~127, kp = (K + 127) & ~127;
  const bool f16 = g_sparse_planes && g_planes_f16 && k.gmeta && (rs == 1 || cs == 1);
  P3Buf Xp = {k.P0, rp, kp, f16 ? k.gmeta + 4 : nullptr}, Yp = {k.P1, rp, kp, f16 ? k.gmeta + 5 : nullptr};
  int e;
  if (f16) {
    if (amaxX) { Xp.part = amaxX; Xp.npart = 1; }
    else if ((e = launch_absmax_view(X, rs, cs, rows, K, Xp, k.gpart + 5 * kPmPartMax, st))) return e;
  }
  if ((e = launch_split3(X, rs, cs, rows, K, Xp, st))) return e;
  if (f16) {
    if (amaxY) { Yp.part = amaxY; Yp.npart = 1; }
    else if ((e = launch_absmax_view(Y, rs, cs, rows, K, Yp, k.gpart + 5 * kPmPartMax, st))) return e;   // (the split above is done with the array)
  }
  if ((e = launch_split3(Y, rs, cs, rows, K, Yp, st))) return e;
  P3Args g = p3_args(Xp, Xp, rows, rows, K, 0);
  g.A2 = p3_of(Yp); g.B2 = p3_of(Yp); g.e.A2 = Y; g.e.K2 = K;
  g.e.epi = EPI_TRIU_MAX; g.e.maxout = maxout; g.e.C = C; g.e.ldc = rows;
  const int T = (rows + 127) / 128, nt = T * (T + 1) / 2, steps = (K + 31) / 32;
  int half = 256 / nt;                                  // chunks per operand pair: at most 512 work items, one round of blocks
  if (half < 1) half = 1;
  if (half > 32) half = 32;
  while (half > 1 && steps / half < 4) --half;
  const int nchunk = 2 * half;
  if ((long)nt * nchunk > kSplitkScratchTiles) return 1;
  if (hipMemsetAsync(k.sk_cnt, 0, (size_t)nt * 4, st) != hipSuccess) return 1;
  if (f16) hipLaunchKernelGGL(k_gemm_p3_splitk<1>, dim3(nt * nchunk), dim3(kThreads), 0, st, g, T, nchunk, k.sk_scratch, k.sk_cnt);
  else hipLaunchKernelGGL(k_gemm_p3_splitk<0>, dim3(nt * nchunk), dim3(kThreads), 0, st, g, T, nchunk, k.sk_scratch, k.sk_cnt);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(kThreads) void k_zero_below_diag(float* __restrict__ A, int n) {
  const long tot = (long)n * n;
  for (long e = (long)blockIdx.x * kThreads + threadIdx.x; e < tot; e += (long)gridDim.x * kThreads)
    if (e % n < e / n) A[e] = 0.0f;
}
static inline int ew_grid_fwd(long tot) {
  long g = (tot + kThreads - 1) / kThreads;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// One product of a sparse-format flow.  Data-sized products against a dense factor (an embedding's 1000 x 1000 factor times
// its 1000 x 30000 gradient) run on f16 x 2 operand planes like the dense (x) dense paths: both operands are split once
// (absmax + split each: four memory-bound launches) and the product is a plane GEMM with the fp32 epilogue of the caller's
// GemmArgs (column scales, D - A B, ...); the in-GEMM split kernel re-splits an operand element once per tile column it meets.
// set: 0 = the caller's stream, 1 = the side stream (their own plane buffers).  Small products, second operand pairs and views
// without a unit stride stay on launch_gemm.
// ONE predicate for "this product runs on planes": sparse_gemm and sparse_solve's choice of route both read it (the explicit-
// inverse solve is only in-place safe as a plane product: the planes are a copy of X, so Y may be X; launch_gemm reads its
// operands while other tiles write C).
static inline bool sparse_gemm_shape_on_planes(const SparseWs& k, long M, long N, long K) {
  const int64_t Mp = pad128(M), Np = pad128(N), Kp = pad128(K);
  const bool fits = k.gcap > 0 && Mp * Kp <= k.gcap && Np * Kp <= k.gcap;
  const bool big = (double)M * N * K >= 4e9 && M >= 256 && N >= 256 && K >= 256;
  return g_sparse_planes && g_planes && g_gemm_x3 && g_planes_f16 && fits && big;
}
static inline bool sparse_gemm_on_planes(const SparseWs& k, const GemmArgs& g) {
  const bool unit = (g.a_rs == 1 || g.a_cs == 1) && (g.b_rs == 1 || g.b_cs == 1);
  return !g.A2 && !g.sym && unit && sparse_gemm_shape_on_planes(k, g.M, g.N, g.K);
}
static int sparse_gemm(const SparseWs& k, const GemmArgs& g, hipStream_t st, int set = 0, PlaneMeta* track = nullptr,
                       bool* tracked = nullptr) {
  if (tracked) *tracked = false;
  if (!sparse_gemm_on_planes(k, g)) return launch_gemm(g, st);
  const int64_t Mp = pad128(g.M), Np = pad128(g.N), Kp = pad128(g.K);
  P3Buf A = {k.GP[set][0], Mp, Kp, k.gmeta + 2 * set}, B = {k.GP[set][1], Np, Kp, k.gmeta + 2 * set + 1};
  float* part = k.gpart + (long)set * 2 * kPmPartMax;
  int e;
  if ((e = launch_absmax_view(g.A, g.a_rs, g.a_cs, g.M, g.K, A, part, st))) return e;
  if ((e = launch_split3(g.A, g.a_rs, g.a_cs, g.M, g.K, A, st))) return e;                     // (x, k) = A(m, k)
  if ((e = launch_absmax_view(g.B, g.b_cs, g.b_rs, g.N, g.K, B, part + kPmPartMax, st))) return e;
  if ((e = launch_split3(g.B, g.b_cs, g.b_rs, g.N, g.K, B, st))) return e;                     // (x, k) = B(k, n)
  P3Args p = p3_args(A, B, g.M, g.N, g.K, g.kmode);
  p.e = g;
  p.e.A2 = nullptr; p.e.kblk = 0;
  if (track && !g.colv && g.epi == EPI_STORE) {            // max|C| for a consumer that splits C into planes (accumulated: zeroed by the caller)
    p.ometa = track;
    if (tracked) *tracked = true;
  }
  return launch_p3_auto(p, k.sk_scratch, k.sk_cnt, st);
}

// The solve y Q = x of a sparse-format flow (see trsm_ut for the arguments).  With many vectors per column of Q -- an embedding's
// 30000 rows against its 1000 x 1000 factor -- the substitution strips are throughput-bound (1875 workgroups per strip: 0.83 ms
// for the two strips and the update between them), and the solve is cheaper as ONE product with the explicit inverse (tri_inverse:
// ~13 launches for n = 1000, then a plane product through sparse_gemm): from 8 vectors per column on, 512 <= n <= 8192.
static inline bool sparse_solve_inverse(const SparseWs& k, int n, int nvec) {
  return g_trsm_inv && k.IInv && n >= 512 && n <= 8192 && (long)nvec >= 8L * n &&
         sparse_gemm_shape_on_planes(k, nvec, n, n);          // (the route exists for the plane product; see sparse_solve)
}
// The factor-only half (the inversion: a chain of small launches that depends on Q alone), for callers that have other work to
// put behind it on the stream; sparse_solve(..., prepared = true) then only runs the product.  No-op when the strips are used.
static int sparse_solve_prepare(const SparseWs& k, const float* Q, int n, int nvec, float* dinv, hipStream_t st) {
  if (!sparse_solve_inverse(k, n, nvec)) return 0;
  const long np = pad128(n);
  int e;
  hipLaunchKernelGGL(k_tri_inv32, dim3((n + 31) / 32), dim3(64), 0, st, Q, n, n, dinv);
  if (hipGetLastError() != hipSuccess) return 1;
  if (hipMemsetAsync(k.imeta, 0, 256, st) != hipSuccess) return 1;                      // (the running maxima start from zero)
  P3Buf Qc = {k.IQc, np, np, k.imeta + 0};
  if ((e = launch_absmax(Q, (long)n * n, Qc, k.gpart + 4 * kPmPartMax, st))) return e;
  if ((e = launch_split3(Q, 1, n, n, n, Qc, st))) return e;                             // (x, k) = Q[k][x]
  InvSide f = {Q, n, dinv, k.IInv, k.ITf, Qc, P3Buf{k.IIr, np, np, k.imeta + 1}, P3Buf{k.IIc, np, np, k.imeta + 1},
               P3Buf{k.ITp, np, np, nullptr}, k.imeta + 2};
  if ((e = tri_inverse_blocks(f, st))) return e;
  int level = 0;
  for (int b = f.b0; b < n; b *= 2, ++level)
    if ((e = tri_inverse_level(f, b, level, st))) return e;
  // the strict lower triangle of Inv was never written: the product's K range (k <= column) and the split's triangle mask
  // would need it zero -- sparse_gemm splits the whole matrix, so clear it by writing the upper triangle's complement here
  hipLaunchKernelGGL(k_zero_below_diag, dim3(ew_grid_fwd((long)n * n)), dim3(kThreads), 0, st, k.IInv, n);
  return (int)hipGetLastError();
}
static int sparse_solve(const SparseWs& k, const float* Q, int n, const float* X, float* Y, int nvec, long si, long sj, float* dinv,
                        hipStream_t st, long xi = 0, long xj = 0, PlaneMeta* track = nullptr, bool* tracked = nullptr,
                        bool prepared = false) {
  if (tracked) *tracked = false;
  if (xi == 0 && xj == 0) { xi = si; xj = sj; }
  if (!sparse_solve_inverse(k, n, nvec))
    return trsm_ut(Q, n, X, Y, nvec, si, sj, dinv, st, xi == si && xj == sj ? 0 : xi, xi == si && xj == sj ? 0 : xj);
  int e;
  if (!prepared && (e = sparse_solve_prepare(k, Q, n, nvec, dinv, st))) return e;
  GemmArgs g = {};
  g.A = X; g.a_rs = xi; g.a_cs = xj;
  g.B = k.IInv; g.b_rs = n; g.b_cs = 1;
  g.C = Y; g.ldc = si; g.c_cs = sj;
  g.M = nvec; g.N = n; g.K = n; g.kmode = KHI_N; g.epi = EPI_STORE;
  // in place (Y = X) the product must read a COPY of X: only the plane path does.  A view the planes cannot take (no unit
  // stride) falls back to the substitution strips, which are in-place safe (the inversion above was then wasted, not wrong).
  if (static_cast<const float*>(Y) == X && !sparse_gemm_on_planes(k, g)) return trsm_ut(Q, n, X, Y, nvec, si, sj, dinv, st);
  return sparse_gemm(k, g, st, 0, track, tracked);
}

static inline int ew_grid(long tot) {
  long g = (tot + kThreads - 1) / kThreads;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

#define SP_LAUNCH_WIDE(kernel, ...)   /* one block of kFinThreads */                \
  do {                                                                          \
    hipLaunchKernelGGL(kernel, dim3(1), dim3(kFinThreads), 0, st, __VA_ARGS__); \
    if (hipGetLastError() != hipSuccess) return PSGD_ERR_LAUNCH;                \
  } while (0)
#define SP_LAUNCH(kernel, grid, ...)                                            \
  do {                                                                          \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, st, __VA_ARGS__); \
    if (hipGetLastError() != hipSuccess) return PSGD_ERR_LAUNCH;                \
  } while (0)

int64_t psgd_kron_sparse_workspace_bytes(int fmt, int M, int N) {
  if (fmt < 0 || fmt > 2 || M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  return sparse_layout(nullptr, fmt, M, N).total;
}

static int sparse_open(int fmt, int M, int N, void* ws, int64_t ws_bytes, SparseWs* k) {
  if (M <= 0 || N <= 0) return PSGD_ERR_SHAPE;
  if (kron_ws_check(ws, ws_bytes, sparse_layout(nullptr, fmt, M, N).total)) return PSGD_ERR_WORKSPACE;
  *k = sparse_layout(static_cast<char*>(ws), fmt, M, N);
  return PSGD_OK;
}

// (dense, scaling) update, psgd.py:276-307
int psgd_kron_ds_update_f32(const float* Ql, const float* qr, const float* dX, const float* dG, int64_t xrs, int64_t xcs,
                            float* QlOut, float* qrOut, int M, int N, float step, float tiny, void* ws,
                            int64_t ws_bytes, void* stream) {
  if (!Ql || !qr || !dX || !dG || !QlOut || !qrOut) return PSGD_ERR_BAD_ARG;
  SparseWs k;
  const int rc = sparse_open(0, M, N, ws, ws_bytes, &k);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(k.scal, 0, 256, st) != hipSuccess) return PSGD_ERR_LAUNCH;
  SP_LAUNCH(k_balance_generic, ew_grid((long)M * M), Ql, (long)M + 1, M, (long)M * M, qr, 1L, N, (long)N, k.LS, k.RS);
  // the product of :295-296 on the side stream next to the solve of :298-299 (kron_fork: they meet at the gradient)
  KronFork* fk = kron_overlap_chains(M, M) ? kron_fork(st) : nullptr;
  KronForkScope fork_scope(fk, st);          // joins on every exit path, early error returns included
  {                                                          // A = (QlS dG) .* qrS          (:295-296)
    GemmArgs g = gemm_args(k.LS, M, false, dG, 0, false, k.A, N, M, N, M, KLO_M);
    g.b_rs = xrs; g.b_cs = xcs;
    g.colv = k.RS;
    KRON_LAUNCH(sparse_gemm(k, g, fk ? fk->side : st, fk ? 1 : 0));
  }
  // Bt = (QlS^-T dX) .* (1/qrS)                             (:298-299); columns independent
  KRON_LAUNCH(sparse_solve(k, k.LS, M, dX, k.Bt, N, 1L, (long)N, k.dinv, st, (long)xcs, (long)xrs));
  SP_LAUNCH(k_col_inv_scale, ew_grid((long)M * N), k.Bt, k.RS, M, N);
  KRON_LAUNCH(fork_scope.join());
  if (k.P0 && g_planes && g_gemm_x3) {                       // grad1 = triu(A A' - Bt Bt')  (:301): few tiles, long K
    KRON_LAUNCH(sparse_grad_splitk(k, k.A, k.Bt, (long)N, 1L, M, N, k.gsq, k.scal, st));
  } else {
    GemmArgs g = gemm_args(k.A, N, false, k.A, N, true, k.gsq, M, M, M, N);
    g.A2 = k.Bt; g.a2_rs = N; g.a2_cs = 1; g.B2 = k.Bt; g.b2_rs = 1; g.b2_cs = N; g.K2 = N;
    g.epi = EPI_TRIU_MAX; g.maxout = k.scal;
    KRON_LAUNCH(sparse_gemm(k, g, st));
  }
  {                                                          // grad2 = colsum(A^2) - colsum(Bt^2)   (:304)
    MatView a = {k.A, N, 1}, b = {k.Bt, N, 1};
    if (col_reduce(k, a, b, nullptr, nullptr, M, N, 2, k.v0, st)) return PSGD_ERR_LAUNCH;
  }
  {                                                          // Ql - (step1 grad1) Ql         (:307)
    GemmArgs g = gemm_args(k.gsq, M, false, k.LS, M, false, QlOut, M, M, M, M, KLO_M | KHI_N);
    g.epi = EPI_D_MINUS; g.D = k.LS; g.ldd = M; g.scale_max = k.scal; g.step = step; g.tiny = tiny;
    KRON_LAUNCH(sparse_gemm(k, g, st));
  }
  SP_LAUNCH_WIDE(k_scale_finalize, k.RS, k.v0, N, step, tiny, qrOut);
  return PSGD_OK;
}

// (dense, scaling) apply, psgd.py:310-322; out contiguous [M,N]
int psgd_kron_ds_apply_f32(const float* Ql, const float* qr, const float* G, int64_t grs, int64_t gcs, float* out, int M,
                           int N, void* ws, int64_t ws_bytes, void* stream) {
  if (!Ql || !qr || !G || !out) return PSGD_ERR_BAD_ARG;
  SparseWs k;
  const int rc = sparse_open(0, M, N, ws, ws_bytes, &k);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (M < N) {                                               // (Ql'Ql) G                     (:318-319)
    KRON_LAUNCH(sparse_gemm(k, gemm_args(Ql, M, true, Ql, M, false, k.gsq, M, M, M, M, KHI_M | KHI_N), st));
    GemmArgs g = gemm_args(k.gsq, M, false, G, 0, false, out, N, M, N, M);
    g.b_rs = grs; g.b_cs = gcs; g.colv = qr; g.colsq = 1;
    KRON_LAUNCH(sparse_gemm(k, g, st));
  } else {                                                   // Ql' (Ql G)                    (:320-321)
    GemmArgs g1 = gemm_args(Ql, M, false, G, 0, false, k.T, N, M, N, M, KLO_M);
    g1.b_rs = grs; g1.b_cs = gcs;
    KRON_LAUNCH(sparse_gemm(k, g1, st));
    GemmArgs g2 = gemm_args(Ql, M, true, k.T, N, false, out, N, M, N, M, KHI_M);
    g2.colv = qr; g2.colsq = 1;                              // .* (qr qr)                     (:322)
    KRON_LAUNCH(sparse_gemm(k, g2, st));
  }
  return PSGD_OK;
}

// shared by the two normalization-left updates: A0 = Ql dG (opt. .* colv), Bt0 = Ql^-T dX (opt. .* 1/colv)
static int norm_left_pair(const SparseWs& k, const float* dX, const float* dG, long xrs, long xcs, int M, int N,
                          const float* colv, float* A0, float* Bt0, hipStream_t st) {
  const float* q0 = k.LS;
  const float* q1 = k.LS + M;
  MatView vx = {dX, xrs, xcs}, vg = {dG, xrs, xcs};
  SP_LAUNCH(k_norm_left, ew_grid((long)M * N), vg, q0, q1, M, N, colv, 0, A0);
  if (col_reduce(k, vx, vx, q0, q1, M, N, 0, k.v0, st)) return PSGD_ERR_LAUNCH;
  SP_LAUNCH(k_norm_left_invT, ew_grid((long)M * N), vx, q0, (const float*)k.v0, M, N, colv, Bt0);
  return PSGD_OK;
}

// (normalization, dense) update, psgd.py:198-246
int psgd_kron_nd_update_f32(const float* ql, const float* Qr, const float* dX, const float* dG, int64_t xrs, int64_t xcs,
                            float* qlOut, float* QrOut, int M, int N, float step, float tiny, void* ws,
                            int64_t ws_bytes, void* stream) {
  if (!ql || !Qr || !dX || !dG || !qlOut || !QrOut) return PSGD_ERR_BAD_ARG;
  SparseWs k;
  const int rc = sparse_open(1, M, N, ws, ws_bytes, &k);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(k.scal, 0, 256, st) != hipSuccess) return PSGD_ERR_LAUNCH;
  SP_LAUNCH(k_balance_generic, ew_grid((long)N * N), ql, 1L, M, (long)2 * M, Qr, (long)N + 1, N, (long)N * N, k.LS, k.RS);
  bool trackA = false, trackB = false;
  // two chains that meet at the row statistics: T = Ql dG -> A = T QrS' (:218-220) on the side stream (kron_fork),
  // Bt = Ql^-T dX -> Bt QrS^-1 (:222-233, solved in place) on the caller's
  KronFork* fk = kron_overlap_chains(N, N) ? kron_fork(st) : nullptr;
  KronForkScope fork_scope(fk, st);          // joins on every exit path, early error returns included
  {
    hipStream_t sf = fk ? fk->side : st;
    MatView vg = {dG, xrs, xcs};
    hipLaunchKernelGGL(k_norm_left, dim3(ew_grid((long)M * N)), dim3(kThreads), 0, sf, vg, (const float*)k.LS,
                       (const float*)(k.LS + M), M, N, (const float*)nullptr, 0, k.T);
    if (hipGetLastError() != hipSuccess) return PSGD_ERR_LAUNCH;
    // (max|A| and max|Bt| ride on the producing products when they run on planes: words of k.scal, zeroed above)
    KRON_LAUNCH(sparse_gemm(k, gemm_args(k.T, N, false, k.RS, N, true, k.A, N, M, N, N, KLO_N), sf, fk ? 1 : 0,
                            reinterpret_cast<PlaneMeta*>(k.scal + 32), &trackA));                        // A = T QrS'  (:220)
  }
  // (the inversion behind Bt's solve depends on QrS alone: queued first, its small launches -- k_tri_inv128 needs whole CUs --
  //  are through before the side stream's full-chip product starts: 173 -> ~50 us for that launch)
  KRON_LAUNCH(sparse_solve_prepare(k, k.RS, N, M, k.dinv, st));
  {
    MatView vx = {dX, xrs, xcs};
    if (col_reduce(k, vx, vx, k.LS, k.LS + M, M, N, 0, k.v0, st)) return PSGD_ERR_LAUNCH;
    SP_LAUNCH(k_norm_left_invT, ew_grid((long)M * N), vx, (const float*)k.LS, (const float*)k.v0, M, N, (const float*)nullptr, k.Bt);
  }
  KRON_LAUNCH(sparse_solve(k, k.RS, N, k.Bt, k.Bt, M, (long)N, 1L, k.dinv, st, 0, 0, reinterpret_cast<PlaneMeta*>(k.scal + 36),
                           &trackB, true));                                                          // Bt QrS^-1, in place  (:233)
  KRON_LAUNCH(fork_scope.join());
  SP_LAUNCH(k_row_stats, (M + 3) / 4, (const float*)k.A, (const float*)k.Bt, M, N, k.v1, k.v2);   // (:235-237)
  SP_LAUNCH_WIDE(k_norm_finalize, (const float*)k.LS, (const float*)k.v1, (const float*)k.v2, M, step, tiny, qlOut);
  if (k.P0 && g_planes && g_gemm_x3) {                       // grad2 = triu(A'A - Bt'Bt)     (:243): few tiles, long K
    KRON_LAUNCH(sparse_grad_splitk(k, k.A, k.Bt, 1L, (long)N, N, M, k.gsq, k.scal, st, trackA ? k.scal + 34 : nullptr,
                                   trackB ? k.scal + 38 : nullptr));
  } else {
    GemmArgs g = gemm_args(k.A, N, true, k.A, N, false, k.gsq, N, N, N, M);
    g.A2 = k.Bt; g.a2_rs = 1; g.a2_cs = N; g.B2 = k.Bt; g.b2_rs = N; g.b2_cs = 1; g.K2 = M;
    g.epi = EPI_TRIU_MAX; g.maxout = k.scal;
    KRON_LAUNCH(sparse_gemm(k, g, st));
  }
  {                                                          // Qr - (step2 grad2) Qr         (:246)
    GemmArgs g = gemm_args(k.gsq, N, false, k.RS, N, false, QrOut, N, N, N, N, KLO_M | KHI_N);
    g.epi = EPI_D_MINUS; g.D = k.RS; g.ldd = N; g.scale_max = k.scal; g.step = step; g.tiny = tiny;
    KRON_LAUNCH(sparse_gemm(k, g, st));
  }
  return PSGD_OK;
}

// (normalization, dense) apply, psgd.py:249-270
int psgd_kron_nd_apply_f32(const float* ql, const float* Qr, const float* G, int64_t grs, int64_t gcs, float* out, int M,
                           int N, void* ws, int64_t ws_bytes, void* stream) {
  if (!ql || !Qr || !G || !out) return PSGD_ERR_BAD_ARG;
  SparseWs k;
  const int rc = sparse_open(1, M, N, ws, ws_bytes, &k);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MatView vg = {G, grs, gcs};
  SP_LAUNCH(k_norm_left, ew_grid((long)M * N), vg, ql, ql + M, M, N, (const float*)nullptr, 0, k.T);   // Ql G (:258-259)
  if (M < N) {                                               // (P Qr') Qr                    (:260-261)
    KRON_LAUNCH(sparse_gemm(k, gemm_args(k.T, N, false, Qr, N, true, k.A, N, M, N, N, KLO_N), st));
    KRON_LAUNCH(sparse_gemm(k, gemm_args(k.A, N, false, Qr, N, false, k.Bt, N, M, N, N, KHI_N), st));
  } else {                                                   // P (Qr'Qr)                     (:263)
    KRON_LAUNCH(sparse_gemm(k, gemm_args(Qr, N, true, Qr, N, false, k.gsq, N, N, N, N, KHI_M | KHI_N), st));
    KRON_LAUNCH(sparse_gemm(k, gemm_args(k.T, N, false, k.gsq, N, false, k.Bt, N, M, N, N), st));
  }
  MatView vz = {k.Bt, N, 1};
  if (col_reduce(k, vz, vz, ql, ql + M, M, N, 1, k.v0, st)) return PSGD_ERR_LAUNCH;                      // (:265)
  SP_LAUNCH(k_norm_leftT, ew_grid((long)M * N), (const float*)k.Bt, ql, (const float*)k.v0, M, N, out);  // (:266-268)
  return PSGD_OK;
}

// (normalization, scaling) update, psgd.py:328-369
int psgd_kron_ns_update_f32(const float* ql, const float* qr, const float* dX, const float* dG, int64_t xrs, int64_t xcs,
                            float* qlOut, float* qrOut, int M, int N, float step, float tiny, void* ws,
                            int64_t ws_bytes, void* stream) {
  if (!ql || !qr || !dX || !dG || !qlOut || !qrOut) return PSGD_ERR_BAD_ARG;
  SparseWs k;
  const int rc = sparse_open(2, M, N, ws, ws_bytes, &k);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  SP_LAUNCH_WIDE(k_balance_generic, ql, 1L, M, (long)2 * M, qr, 1L, N, (long)N, k.LS, k.RS);
  int e = norm_left_pair(k, dX, dG, xrs, xcs, M, N, k.RS, k.A, k.Bt, st);          // (:349-356)
  if (e) return e;
  SP_LAUNCH(k_row_stats, (M + 3) / 4, (const float*)k.A, (const float*)k.Bt, M, N, k.v1, k.v2);   // (:358-360)
  SP_LAUNCH_WIDE(k_norm_finalize, (const float*)k.LS, (const float*)k.v1, (const float*)k.v2, M, step, tiny, qlOut);
  MatView a = {k.A, N, 1}, b = {k.Bt, N, 1};
  if (col_reduce(k, a, b, nullptr, nullptr, M, N, 2, k.v3, st)) return PSGD_ERR_LAUNCH;                  // (:366)
  SP_LAUNCH_WIDE(k_scale_finalize, (const float*)k.RS, (const float*)k.v3, N, step, tiny, qrOut);
  return PSGD_OK;
}

// (normalization, scaling) apply, psgd.py:372-391
int psgd_kron_ns_apply_f32(const float* ql, const float* qr, const float* G, int64_t grs, int64_t gcs, float* out, int M,
                           int N, void* ws, int64_t ws_bytes, void* stream) {
  if (!ql || !qr || !G || !out) return PSGD_ERR_BAD_ARG;
  SparseWs k;
  const int rc = sparse_open(2, M, N, ws, ws_bytes, &k);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  MatView vg = {G, grs, gcs};
  SP_LAUNCH(k_norm_left, ew_grid((long)M * N), vg, ql, ql + M, M, N, qr, 1, k.T);                       // (:383-385)
  MatView vz = {k.T, N, 1};
  if (col_reduce(k, vz, vz, ql, ql + M, M, N, 1, k.v0, st)) return PSGD_ERR_LAUNCH;                      // (:386)
  SP_LAUNCH(k_norm_leftT, ew_grid((long)M * N), (const float*)k.T, ql, (const float*)k.v0, M, N, out);   // (:387-389)
  return PSGD_OK;
}

}  // extern "C"
