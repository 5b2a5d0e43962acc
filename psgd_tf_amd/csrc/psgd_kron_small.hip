// psgd_kron_small.hip -- dense (x) dense Kronecker preconditioner for SMALL layers (M, N <= 512; LeNet5: mnist_with_lenet5.py:12-16)
// through the reference's own per-layer call pattern (mnist_with_lenet5.py:51,53: one update_precond_kron / precond_grad_kron
// call per layer).  Such a call is bound by its chain of dependent launches (a trivial kernel occupies ~5 us there), not by
// arithmetic, so a call is ONE launch for layers one workgroup can finish quickly, and one launch per phase (2 for the apply, 4
// for the update) for layers worth spreading over many workgroups -- instead of 3 / 5 launches of generic stage kernels.
//
//   _precond_grad_dense_dense (psgd.py:182-192)     out = Ql' (Ql G Qr') Qr            two phases
//   _update_precond_dense_dense (psgd.py:156-179)   balance, A = L dG R', Bt = L^-T dX R^-1, two gradients, two factor updates
//
// Everything is exact fp32 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: fp32 fma chains), 16 x 16 tiles.  The unit of
// work is a 16-row STRIP of the data matrix: a strip's chained products never leave the registers / LDS of one workgroup.  The
// idiom that makes this cheap: the 16 x 16 C/D register layout of the MFMA (lane holds C[4 (lane >> 4) + e][lane & 15]) IS a
// valid B operand (B[k = lane >> 4][n = lane & 15], one MFMA per e), contracting over the tile's ROW index.  So every
// intermediate is produced as the TRANSPOSED tile whose rows are the next product's contraction index:
//     Y'[l, i] = sum_k G[k, l] Ql[i, k]       (A operand G' from memory, B operand Ql' from memory)       C layout [l][i]
//     X'[j, i] = sum_l Qr[j, l] Y'[l, i]      (A operand Qr from memory,  B operand = the Y' tiles as they are)
// and the triangular solves run right-looking over 16-blocks the same way (W[t'] -= Q[t, t']' Y[t], Y[t] = Dinv_t' W[t]).
// The products are re-associated against the reference ((L dG) R' for L (dG R'), (L^-T dX) R^-1 for L^-T (dX R^-1), no Gram in the
// apply): same mathematics, fp32 rounding differs at the 1e-7 level; the parity tests hold the same 1e-5 / 2e-3 bars.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kron_shared.h"
#include "nanmax.h"

namespace psgdk {
namespace {

using psgd::amaxf;
using psgd::nmaxf;

constexpr int kTh = 256;
constexpr int kMaxT = 8;                     // tiles of a 32-tile row / column (512 / 16) per wave of four
constexpr int kStripT = 4;                   // tiles of a data strip per wave: kron_small_fused keeps N <= 256 (16 tiles over 4 waves)
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// An upper-triangular factor, optionally balanced on the fly (psgd.py:169-170): mode 0 = q / rho (left), 1 = rho * q (right),
// 2 = q.  Entries below the diagonal are never used (the reference's invariant); addresses are clamped, values selected.
// Every operand is read through a buffer descriptor (32-bit byte offsets in one VGPR, the hardware range check returns 0 past the
// end): a masked element is a load at an offset past the buffer -- no clamped addresses, no selects behind the load, and the
// address of a load in flight costs one register instead of two.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOob = 0x7ff00000;             // byte offset past every buffer (matrices here are at most 512 x 512 floats)
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, int floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, floats * 4, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// An upper-triangular factor [n x n], optionally balanced on the fly (psgd.py:169-170; mul = 1 / rho, rho, or 1).  Entries below the
// diagonal are never read (the reference's invariant).
struct Tri { rsrc_t rs; int n; float mul; };
__device__ __forceinline__ Tri make_tri(const float* Q, int n, float mul) { return Tri{make_rsrc(Q, n * n), n, mul}; }

// (row r0 + e, column c), e = 0..3
__device__ __forceinline__ void tri_col4(const Tri& t, int r0, int c, float (&v)[4]) {
  const int base = c < t.n ? (r0 * t.n + c) * 4 : kOob;                // (rows past n run off the end of the buffer by themselves)
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = bload(t.rs, r0 + e <= c ? base : kOob, e * t.n * 4) * t.mul;
}
// (row r, column c0 + e)
__device__ __forceinline__ void tri_row4(const Tri& t, int r, int c0, float (&v)[4]) {
  const int base = r < t.n ? (r * t.n + c0) * 4 : kOob;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = bload(t.rs, (c0 + e >= r && c0 + e < t.n) ? base + 4 * e : kOob, 0) * t.mul;
}
// dense [R x C] row-major: (row r0 + e, column c), zero outside
struct Dense { rsrc_t rs; int R, C; };
__device__ __forceinline__ Dense make_dense(const float* D, int R, int C) { return Dense{make_rsrc(D, R * C), R, C}; }
__device__ __forceinline__ void dense_col4(const Dense& d, int r0, int c, float (&v)[4]) {
  const int base = c < d.C ? (r0 * d.C + c) * 4 : kOob;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = bload(d.rs, base, e * d.C * 4);
}
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

constexpr int kRing = 4;                     // chunks in flight per wave: these layers are bound by load LATENCY (a dependent round trip
                                             // costs as much as dozens of MFMAs), so every chunk loop keeps kRing chunks requested ahead

template <int N> struct IC { static constexpr int value = N; };
// f(IC<n>) for the smallest n in {1, 2, 3, 4} (MT = 4) or {1, 2, 4, 8} (MT = 8) that is >= nt: the chunk loops are straight-line code
// per tile count (loads under data-dependent branches would turn every counted wait into a full drain)
template <int MT, class F>
__device__ __forceinline__ void for_tiles(int nt, F f) {
  if constexpr (MT == 4) {
    if (nt <= 1) f(IC<1>{}); else if (nt == 2) f(IC<2>{}); else if (nt == 3) f(IC<3>{}); else f(IC<4>{});
  } else {
    if (nt <= 1) f(IC<1>{}); else if (nt == 2) f(IC<2>{}); else if (nt <= 4) f(IC<4>{}); else f(IC<8>{});
  }
}

// ---- a strip's first product: acc[t] += sum over 16-chunks c in [c_lo, c_hi) of a(t, c) x b(c), t < NT (exact)
template <int NT, int MT, class LA, class LB>
__device__ __forceinline__ void strip_product_n(f32x4 (&acc)[MT], int c_lo, int c_hi, LA la, LB lb) {
  if (c_lo >= c_hi) return;
  float br[kRing][4], ar[kRing][NT][4];
#pragma unroll
  for (int d = 0; d < kRing; ++d) {
    const int cc = min(c_lo + d, c_hi - 1);
    lb(cc, br[d]);
#pragma unroll
    for (int t = 0; t < NT; ++t) la(t, cc, ar[d][t]);
  }
  for (int c = c_lo; c < c_hi; c += kRing) {
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
      const bool live = c + d < c_hi;                                    // (uniform)
      float b[4], a[NT][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = live ? br[d][e] : 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[t][e] = ar[d][t][e];
      const int cn = min(c + d + kRing, c_hi - 1);
      lb(cn, br[d]);
#pragma unroll
      for (int t = 0; t < NT; ++t) la(t, cn, ar[d][t]);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t] = MFMA4(a[t][e], b[e], acc[t]);
    }
  }
}
template <int MT, class LA, class LB>
__device__ __forceinline__ void strip_product(f32x4 (&acc)[MT], int nt, int c_lo, int c_hi, LA la, LB lb) {
  if (nt <= 0) return;
  for_tiles<MT>(nt, [&](auto n) { strip_product_n<decltype(n)::value, MT>(acc, c_lo, c_hi, la, lb); });
}

// ---- a strip's second product: out[t] (tile jt = wl + t * WS) = sum over tiles lt in [lo(jt), hi(jt)) of a(jt, lt) x TL[lt], the tiles
// of the first product as they lie in LDS (C layout = B operand)
template <int MT, class LA, class RG>
__device__ __forceinline__ void strip_product2(f32x4 (&out)[MT], int nt, int wl, int WS, const f32x4* __restrict__ TL, int lane,
                                               LA la, RG range) {
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    if (t >= nt) continue;
    const int jt = wl + t * WS;
    int lo, hi;
    range(jt, lo, hi);
    f32x4 acc = zero4();
    if (lo < hi) {
      float ar[kRing][4];
#pragma unroll
      for (int d = 0; d < kRing; ++d) la(jt, min(lo + d, hi - 1), ar[d]);
      for (int l0 = lo; l0 < hi; l0 += kRing) {
#pragma unroll
        for (int d = 0; d < kRing; ++d) {
          const int lt = min(l0 + d, hi - 1);
          const bool live = l0 + d < hi;
          float a[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = live ? ar[d][e] : 0.0f;
          la(jt, min(l0 + d + kRing, hi - 1), ar[d]);
          const f32x4 b = TL[lt * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = MFMA4(a[e], b[e], acc);
        }
      }
    }
    out[t] = acc;
  }
}

// ---- the mirror image of strip_product for tile ROWS: acc[t] += sum over chunks c in [c_lo, c_hi) of (amul a(c)) x b(t, c) -- the A
// operand (the strip's rows) is shared by the wave's tiles, the B operand is per tile.  b(t, c) must be valid for every t < NT (the
// caller clamps tile indices past its count onto its last tile and ignores their results).
template <int NT, int MT, class LA, class LB>
__device__ __forceinline__ void strip_product_bt_n(f32x4 (&acc)[MT], int c_lo, int c_hi, float amul, LA la, LB lb) {
  if (c_lo >= c_hi) return;
  float ar[kRing][4], br[kRing][NT][4];
#pragma unroll
  for (int d = 0; d < kRing; ++d) {
    const int cc = min(c_lo + d, c_hi - 1);
    la(cc, ar[d]);
#pragma unroll
    for (int t = 0; t < NT; ++t) lb(t, cc, br[d][t]);
  }
  for (int c = c_lo; c < c_hi; c += kRing) {
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
      const bool live = c + d < c_hi;
      float a[4], b[NT][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = live ? ar[d][e] * amul : 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) b[t][e] = br[d][t][e];
      const int cn = min(c + d + kRing, c_hi - 1);
      la(cn, ar[d]);
#pragma unroll
      for (int t = 0; t < NT; ++t) lb(t, cn, br[d][t]);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t] = MFMA4(a[e], b[t][e], acc[t]);
    }
  }
}
template <int MT, class LA, class LB>
__device__ __forceinline__ void strip_product_bt(f32x4 (&acc)[MT], int nt, int c_lo, int c_hi, float amul, LA la, LB lb) {
  if (nt <= 0) return;
  for_tiles<MT>(nt, [&](auto n) { strip_product_bt_n<decltype(n)::value, MT>(acc, c_lo, c_hi, amul, la, lb); });
}

// Group barrier: the WS waves that share a strip.  WS == 4: the whole workgroup; WS == 1: the wave alone (LDS traffic of one wave
// is in order; the fence keeps the compiler from moving accesses across).
__device__ __forceinline__ void group_sync(int WS) {
  if (WS == 4) __syncthreads();
  else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// ---- Inverses of the 16 x 16 diagonal blocks of a (balanced) factor into DI[blk][r][c] (LDS), identity-padded at the ragged end;
// 16 lanes per block (lane r computes row r by substitution), 16 blocks per pass of the workgroup.  Dt: staging, 16 x [16][17].
__device__ __forceinline__ void diag_inverses(const Tri& q, int nb, float* __restrict__ DI, float* __restrict__ Dt) {
  const int tid = threadIdx.x, slot = tid >> 4, r = tid & 15;
  for (int b0 = 0; b0 < nb; b0 += 16) {
    const int blk = b0 + slot;
    float* D = Dt + slot * (16 * 17);
    if (blk < nb) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int gr = 16 * blk + r, gc = 16 * blk + c;
        const float v = bload(q.rs, (gr <= gc && gc < q.n) ? (gr * q.n + gc) * 4 : kOob, 0) * q.mul;
        D[r * 17 + c] = (gr >= q.n && r == c) ? 1.0f : v;
      }
    }
    __syncthreads();
    if (blk < nb) {
      float rr[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) rr[j] = (j == r) ? 1.0f : 0.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float y = rr[j] / D[j * 17 + j];
        rr[j] = y;
#pragma unroll
        for (int j2 = j + 1; j2 < 16; ++j2) rr[j2] = fmaf(-y, D[j * 17 + j2], rr[j2]);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) DI[blk * 256 + r * 16 + j] = rr[j];
    }
    __syncthreads();
  }
}

// ---- Right-looking triangular solve of one 16-wide strip:  Q' Y = X  (Q upper triangular [n x n], balanced on the fly), tiles
// t = 0 .. nt-1 along Q's dimension, C layout [q = 16 t + 4 (lane >> 4) + e][v = lane & 15].  The WS waves of the group own the tiles
// t == wl (mod WS).  step t: the owner finishes Y[t] = Dinv_t' W[t], publishes it in TL[t]; everybody updates its later tiles,
// W[t'] -= Q[t, t']' Y[t].  loadx(t) -> W[t]; storey(t, Y[t]).
template <int MT, class LX, class SY>
__device__ __forceinline__ void solve_strip(const Tri& q, int nt, int wl, int WS, const float* __restrict__ DI, f32x4* __restrict__ TL,
                                            int lane, LX loadx, SY storey) {
  const int l15 = lane & 15, g4 = 4 * (lane >> 4);
  f32x4 W[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) W[t] = loadx(min(wl + t * WS, nt - 1));       // (tiles past nt: a copy of the last one, never stored)
  // Q[tile s, tile tt] for this wave's tiles, requested one step ahead of its use (the loads depend on nothing computed)
  float an[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t) tri_col4(q, g4, 16 * (wl + t * WS) + l15, an[t]);
  for (int s = 0; s < nt; ++s) {
    float a[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[t][e] = an[t][e];
#pragma unroll
    for (int t = 0; t < MT; ++t) tri_col4(q, 16 * min(s + 1, nt - 1) + g4, 16 * (wl + t * WS) + l15, an[t]);
    if (s % WS == wl) {
      const int ts = s / WS;
      f32x4 w = zero4();
#pragma unroll
      for (int t = 0; t < MT; ++t)
        if (t == ts) w = W[t];
      f32x4 y = zero4();
#pragma unroll
      for (int e = 0; e < 4; ++e) y = MFMA4(DI[s * 256 + (g4 + e) * 16 + l15], w[e], y);      // A[m = q][k = q'] = Dinv[q'][q]
      TL[s * 64 + lane] = y;
      storey(s, y);
    }
    group_sync(WS);
    const f32x4 y = TL[s * 64 + lane];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int tt = wl + t * WS;
      if (tt > s && tt < nt) {                                          // (uniform; MFMAs only)
#pragma unroll
        for (int e = 0; e < 4; ++e) W[t] = MFMA4(-a[t][e], y[e], W[t]);   // A[m = q' (tile tt)][k = q (tile s)] = Q[q][q']
      }
    }
  }
  group_sync(WS);                                                      // TL may be reused
}

// C-layout tile -> row-major Y[(r0 + g4 + e) * ld + c0 + l15]  (whole tile: padded scratch)
__device__ __forceinline__ void store_tile(float* __restrict__ Y, int ld, int r0, int c0, int lane, const f32x4& v) {
  const int l15 = lane & 15, g4 = 4 * (lane >> 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) Y[(long)(r0 + g4 + e) * ld + c0 + l15] = v[e];
}
// C-layout tile -> its transpose, row-major YT[(c0 + l15) * ld + r0 + g4 .. + 3]  (16-byte store; ld, r0 multiples of 4)
__device__ __forceinline__ void store_tile_t(float* __restrict__ YT, int ld, int r0, int c0, int lane, const f32x4& v) {
  const int l15 = lane & 15, g4 = 4 * (lane >> 4);
  *reinterpret_cast<f32x4*>(YT + (long)(c0 + l15) * ld + r0 + g4) = v;
}
__device__ __forceinline__ void row4(const float* __restrict__ P, int ld, int r, int c0, float (&v)[4]) {
  const f32x4 x = *reinterpret_cast<const f32x4*>(P + (long)r * ld + c0);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = x[e];
}

// ---- balance: rho = sqrt(max diag Ql / max diag Qr)  (psgd.py:166-168; signed maxima, NaN propagates)
__device__ __forceinline__ float balance_rho(const float* __restrict__ Ql, const float* __restrict__ Qr, int M, int N, float* red) {
  float ml = -INFINITY, mr = -INFINITY;
  for (int i = threadIdx.x; i < M; i += kTh) ml = nmaxf(ml, Ql[(long)i * M + i]);
  for (int i = threadIdx.x; i < N; i += kTh) mr = nmaxf(mr, Qr[(long)i * N + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ml = nmaxf(ml, __shfl_down(ml, off, 64));
    mr = nmaxf(mr, __shfl_down(mr, off, 64));
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w] = ml; red[4 + w] = mr; }
  __syncthreads();
  ml = nmaxf(nmaxf(red[0], red[1]), nmaxf(red[2], red[3]));
  mr = nmaxf(nmaxf(red[4], red[5]), nmaxf(red[6], red[7]));
  __syncthreads();
  return sqrtf(ml / mr);
}

// =============================================================================================================== apply
struct SmallApply {
  const float *Ql, *Qr, *G;
  float* out;
  float* XT;                 // scratch [Np][Mp]: (Ql G Qr')'
  int M, N, Mp, Np, S, NT, WS;
};

// phases: bit 0 = X' = (Ql G Qr')' by strips, bit 1 = out = Ql' X Qr by strips; 3 = both in one workgroup (gridDim.x == 1)
__global__ __launch_bounds__(kTh) void k_kron_small_apply(SmallApply p, int phases) {
  extern __shared__ __attribute__((aligned(16))) float small_lds[];
  f32x4* TLall = reinterpret_cast<f32x4*>(small_lds);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, g4 = 4 * (lane >> 4);
  const int WS = p.WS, groups = 4 / WS, grp = w / WS, wl = w % WS;
  f32x4* TL = TLall + (WS == 4 ? 0 : grp * 64);        // (WS == 1 only with NT == 1: one tile per group)
  const int per_round = gridDim.x * groups;
  const int rounds = (p.S + per_round - 1) / per_round;
  const Tri L = make_tri(p.Ql, p.M, 1.0f), R = make_tri(p.Qr, p.N, 1.0f);
  const Dense Gd = make_dense(p.G, p.M, p.N);
  int ntw = 0;                                           // this wave's tiles of a strip
  for (int t = 0; t < kStripT; ++t) ntw += (wl + t * WS < p.NT) ? 1 : 0;
  for (int ph = 1; ph <= 2; ++ph) {
    if (!(phases & ph)) continue;
    for (int r = 0; r < rounds; ++r) {
      const int s = (r * gridDim.x + blockIdx.x) * groups + grp;
      const bool act = s < p.S;                          // (an idle group still meets the workgroup's barriers)
      const int i0 = 16 * s;
      f32x4 acc[kStripT];
#pragma unroll
      for (int t = 0; t < kStripT; ++t) acc[t] = zero4();
      if (act) {
        if (ph == 1) {
          // Y'[l, i] = sum_{k >= i0} G[k, l] Ql[i, k]
          strip_product<kStripT>(acc, ntw, s, p.S,
                        [&](int t, int c, float (&a)[4]) { dense_col4(Gd, 16 * c + g4, 16 * (wl + t * WS) + l15, a); },
                        [&](int c, float (&b)[4]) { tri_row4(L, i0 + l15, 16 * c + g4, b); });
        } else {
          // Z'[l, i] = sum_{k <= i0 + 15} X'[l, k] Ql[k, i]
          strip_product<kStripT>(acc, ntw, 0, s + 1,
                        [&](int t, int c, float (&a)[4]) { row4(p.XT, p.Mp, 16 * (wl + t * WS) + l15, 16 * c + g4, a); },
                        [&](int c, float (&b)[4]) { tri_col4(L, 16 * c + g4, i0 + l15, b); });
        }
#pragma unroll
        for (int t = 0; t < kStripT; ++t)
          if (t < ntw) TL[(wl + t * WS) * 64 + lane] = acc[t];
      }
      group_sync(WS);
      if (act) {
        f32x4 out[kStripT];
        if (ph == 1) {
          // X'[j, i] = sum_{l >= j0} Qr[j, l] Y'[l, i]
          strip_product2<kStripT>(out, ntw, wl, WS, TL, lane,
                         [&](int jt, int lt, float (&a)[4]) { tri_row4(R, 16 * jt + l15, 16 * lt + g4, a); },
                         [&](int jt, int& lo, int& hi) { lo = jt; hi = p.NT; });
#pragma unroll
          for (int t = 0; t < kStripT; ++t)
            if (t < ntw) store_tile(p.XT, p.Mp, 16 * (wl + t * WS), i0, lane, out[t]);
        } else {
          // out'[j, i] = sum_{l <= j0 + 15} Qr[l, j] Z'[l, i]
          strip_product2<kStripT>(out, ntw, wl, WS, TL, lane,
                         [&](int jt, int lt, float (&a)[4]) { tri_col4(R, 16 * lt + g4, 16 * jt + l15, a); },
                         [&](int jt, int& lo, int& hi) { lo = 0; hi = jt + 1; });
#pragma unroll
          for (int t = 0; t < kStripT; ++t)
            if (t < ntw) {
              const int i = i0 + l15, j0 = 16 * (wl + t * WS) + g4;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (i < p.M && j0 + e < p.N) p.out[(long)i * p.N + j0 + e] = out[t][e];
            }
        }
      }
      group_sync(WS);                                    // TL is free for the next round
    }
    if (phases == 3 && ph == 1) {                        // one workgroup: X' written above is read below by other waves
      __threadfence();
      __syncthreads();
    }
  }
}

// =============================================================================================================== update
struct SmallUpdate {
  const float *Ql, *Qr, *dX, *dG;
  float *QlOut, *QrOut;
  float *A, *AT, *Bt, *BtT, *P, *g1, *g2;    // scratch: A, Bt, P [Mp][Np]; AT, BtT [Np][Mp]; g1 [Mp][Mp]; g2 [Np][Np]
  unsigned* sync;                            // [0], [1]: bits of max|grad1|, max|grad2|
  int M, N, Mp, Np, S, NT, WS;
  float step, tiny;
};

// phases: bit 0 = A (row strips) and P = L^-T dX (column strips); bit 1 = Bt = P R^-1 (row strips); bit 2 = the two gradients and
// their maxima; bit 3 = the two factor updates.  15 = everything in ONE workgroup (gridDim.x == 1: barriers instead of launches).
__global__ __launch_bounds__(kTh) void k_kron_small_update(SmallUpdate p, int phases) {
  extern __shared__ __attribute__((aligned(16))) float small_lds[];
  f32x4* TLall = reinterpret_cast<f32x4*>(small_lds);                 // 32 tiles
  float* DI = small_lds + 32 * 256;                                   // 32 inverted diagonal blocks
  float* Dt = DI + 32 * 256;                                          // staging of diag_inverses: 16 x [16][17]
  float* red = Dt + 16 * 16 * 17;                                     // 8 floats
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, g4 = 4 * (lane >> 4);
  const bool one = gridDim.x == 1;
  // A = L dG R' and Bt = L^-T dX R^-1 do not depend on the balance (rho cancels: L = Ql / rho, R = rho Qr), so the phases ahead of
  // the factor updates work on Ql, Qr as they are and the maxima of the diagonals are off their critical path
  const Tri L = make_tri(p.Ql, p.M, 1.0f), R = make_tri(p.Qr, p.N, 1.0f);
  const Dense dGd = make_dense(p.dG, p.M, p.N), dXd = make_dense(p.dX, p.M, p.N);
  if ((phases & 1) && blockIdx.x == 0 && threadIdx.x < 2) p.sync[threadIdx.x] = 0u;     // the maxima of phase 3 (a launch / barrier ahead)

  if (phases & 1) {
    // ---- A = (L dG) R' by row strips: A'[j, i] = sum_{l >= j} R[j, l] Y'[l, i],  Y'[l, i] = sum_{k >= i0} dG[k, l] L[i, k]   (:173)
    {
      const int WS = p.WS, groups = 4 / WS, grp = w / WS, wl = w % WS;
      f32x4* TL = TLall + (WS == 4 ? 0 : grp * 64);
      int ntw = 0;
      for (int t = 0; t < kStripT; ++t) ntw += (wl + t * WS < p.NT) ? 1 : 0;
      const int per_round = gridDim.x * groups, rounds = (p.S + per_round - 1) / per_round;
      for (int r = 0; r < rounds; ++r) {
        const int s = (r * gridDim.x + blockIdx.x) * groups + grp;
        const bool act = s < p.S;
        const int i0 = 16 * s;
        if (act) {
          f32x4 acc[kStripT];
#pragma unroll
          for (int t = 0; t < kStripT; ++t) acc[t] = zero4();
          strip_product<kStripT>(acc, ntw, s, p.S,
                        [&](int t, int c, float (&a)[4]) { dense_col4(dGd, 16 * c + g4, 16 * (wl + t * WS) + l15, a); },
                        [&](int c, float (&b)[4]) { tri_row4(L, i0 + l15, 16 * c + g4, b); });
#pragma unroll
          for (int t = 0; t < kStripT; ++t)
            if (t < ntw) TL[(wl + t * WS) * 64 + lane] = acc[t];
        }
        group_sync(WS);
        if (act) {
          f32x4 out[kStripT];
          strip_product2<kStripT>(out, ntw, wl, WS, TL, lane,
                         [&](int jt, int lt, float (&a)[4]) { tri_row4(R, 16 * jt + l15, 16 * lt + g4, a); },
                         [&](int jt, int& lo, int& hi) { lo = jt; hi = p.NT; });
#pragma unroll
          for (int t = 0; t < kStripT; ++t)
            if (t < ntw) {
              store_tile(p.AT, p.Mp, 16 * (wl + t * WS), i0, lane, out[t]);
              store_tile_t(p.A, p.Np, 16 * (wl + t * WS), i0, lane, out[t]);
            }
        }
        group_sync(WS);
      }
    }
    // ---- P = L^-T dX by column strips (L' P = dX): tiles along the rows, all four waves on one strip   (:174, left solve first)
    {
      bool mine = false;
      for (int js = 0; js < p.NT; ++js) mine = mine || ((int)gridDim.x - 1 - js % (int)gridDim.x) == (int)blockIdx.x;
      if (mine) {                                            // (uniform per workgroup)
        __syncthreads();
        diag_inverses(L, p.S, DI, Dt);
        for (int js = 0; js < p.NT; ++js) {
          if (((int)gridDim.x - 1 - js % (int)gridDim.x) != (int)blockIdx.x) continue;   // dealt from the last workgroup down
          const int j0 = 16 * js;
          solve_strip<kMaxT>(L, p.S, w, 4, DI, TLall, lane,
                      [&](int t) {
                        float x[4];
                        dense_col4(dXd, 16 * t + g4, j0 + l15, x);
                        return f32x4{x[0], x[1], x[2], x[3]};
                      },
                      [&](int t, const f32x4& y) { store_tile(p.P, p.Np, 16 * t, j0, lane, y); });
        }
      }
    }
    if (one) { __threadfence(); __syncthreads(); }
  }

  if (phases & 2) {
    // ---- Bt = P R^-1 by row strips (R' Bt' = P'): tiles along the columns   (:174)
    const int WS = p.WS, groups = 4 / WS, grp = w / WS, wl = w % WS;
    f32x4* TL = TLall + (WS == 4 ? 0 : grp * 64);
    diag_inverses(R, p.NT, DI, Dt);
    const int per_round = gridDim.x * groups, rounds = (p.S + per_round - 1) / per_round;
    for (int r = 0; r < rounds; ++r) {
      const int s = (r * gridDim.x + blockIdx.x) * groups + grp;
      const int i0 = 16 * s;
      if (s < p.S)                                       // (uniform for the waves that share the strip's barriers)
        solve_strip<kStripT>(R, p.NT, wl, WS, DI, TL, lane,
                    [&](int t) {
                      float x[4];
                      row4(p.P, p.Np, i0 + l15, 16 * t + g4, x);
                      return f32x4{x[0], x[1], x[2], x[3]};
                    },
                    [&](int t, const f32x4& y) {
                      store_tile(p.BtT, p.Mp, 16 * t, i0, lane, y);
                      store_tile_t(p.Bt, p.Np, 16 * t, i0, lane, y);
                    });
    }
    if (one) { __threadfence(); __syncthreads(); }
  }

  if (phases & 4) {
    // ---- grad1 = triu(A A' - Bt Bt') [M x M], grad2 = triu(A'A - Bt'Bt) [N x N]   (:175-176).  A wave task = one tile ROW a of a
    // gradient and up to 8 of its tiles b >= a: the row's operand is loaded once per chunk, the tiles' loads are all in flight together
    // (a task per 16 x 16 tile would be a chain of exposed load latencies: these layers are latency-bound, not flop-bound).
    float vmax1 = 0.0f, vmax2 = 0.0f;
    int ntask1 = 0, ntask2 = 0;
    for (int a = 0; a < p.S; ++a) ntask1 += (p.S - a + kMaxT - 1) / kMaxT;
    for (int a = 0; a < p.NT; ++a) ntask2 += (p.NT - a + kMaxT - 1) / kMaxT;
    for (int id = blockIdx.x * 4 + w; id < ntask1 + ntask2; id += gridDim.x * 4) {
      const bool first = id < ntask1;
      const int n = first ? p.S : p.NT;
      int rem = first ? id : id - ntask1, a = 0;
      while (rem >= (n - a + kMaxT - 1) / kMaxT) { rem -= (n - a + kMaxT - 1) / kMaxT; ++a; }
      const int b0 = a + rem * kMaxT, nt = min(kMaxT, n - b0);
      const float* __restrict__ X = first ? p.A : p.AT;
      const float* __restrict__ Y = first ? p.Bt : p.BtT;
      const int ld = first ? p.Np : p.Mp, nc = first ? p.NT : p.S;
      f32x4 acc[kMaxT];
#pragma unroll
      for (int t = 0; t < kMaxT; ++t) acc[t] = zero4();
      // (tile slots past nt read the last tile again; their sums are not stored)
      strip_product_bt<kMaxT>(acc, nt, 0, nc, 1.0f, [&](int c, float (&x)[4]) { row4(X, ld, 16 * a + l15, 16 * c + g4, x); },
                              [&](int t, int c, float (&x)[4]) { row4(X, ld, 16 * (b0 + min(t, nt - 1)) + l15, 16 * c + g4, x); });
      strip_product_bt<kMaxT>(acc, nt, 0, nc, -1.0f, [&](int c, float (&x)[4]) { row4(Y, ld, 16 * a + l15, 16 * c + g4, x); },
                              [&](int t, int c, float (&x)[4]) { row4(Y, ld, 16 * (b0 + min(t, nt - 1)) + l15, 16 * c + g4, x); });
      float* __restrict__ Gd = first ? p.g1 : p.g2;
      const int ldg = first ? p.Mp : p.Np, nn = first ? p.M : p.N;
      float m = 0.0f;
#pragma unroll
      for (int t = 0; t < kMaxT; ++t)
        if (t < nt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = 16 * a + g4 + e, col = 16 * (b0 + t) + l15;
            const float v = (col >= row && col < nn) ? acc[t][e] : 0.0f;     // triu; (rows / columns past the matrix are zero anyway)
            Gd[(long)row * ldg + col] = v;
            m = amaxf(m, fabsf(v));
          }
        }
      if (first) vmax1 = amaxf(vmax1, m); else vmax2 = amaxf(vmax2, m);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      vmax1 = amaxf(vmax1, __shfl_down(vmax1, off, 64));
      vmax2 = amaxf(vmax2, __shfl_down(vmax2, off, 64));
    }
    // integer max on the bits of a non-negative float: order-independent (deterministic), NaN (above +inf) propagates
    if (lane == 0 && __float_as_uint(vmax1) != 0u) atomicMax(p.sync + 0, __float_as_uint(vmax1));
    if (lane == 0 && __float_as_uint(vmax2) != 0u) atomicMax(p.sync + 1, __float_as_uint(vmax2));
    if (one) { __threadfence(); __syncthreads(); }
  }

  if (phases & 8) {
    // ---- Lnew = L - (step1 grad1) L,  Rnew = R - (step2 grad2) R   (:177-179).  A wave task = tile row a of a factor and up to 8 of
    // its tiles b (all of them: below the diagonal the balanced factor is copied, as the reference's elementwise ops would);
    // chunks kt = a .. b of grad[a, kt] Q[kt, b] (grad upper: kt >= a, Q upper: kt <= b).
    const float rho = balance_rho(p.Ql, p.Qr, p.M, p.N, red);           // (:166-170)
    const Tri Lb = make_tri(p.Ql, p.M, 1.0f / rho), Rb = make_tri(p.Qr, p.N, rho);
    const float s1 = p.step / (__uint_as_float(__hip_atomic_load(p.sync + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + p.tiny);
    const float s2 = p.step / (__uint_as_float(__hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + p.tiny);
    const int g1n = (p.S + kMaxT - 1) / kMaxT, g2n = (p.NT + kMaxT - 1) / kMaxT;
    const int ntask1 = p.S * g1n, ntask2 = p.NT * g2n;
    for (int id = blockIdx.x * 4 + w; id < ntask1 + ntask2; id += gridDim.x * 4) {
      const bool first = id < ntask1;
      const int n = first ? p.S : p.NT, gn = first ? g1n : g2n, lid = first ? id : id - ntask1;
      const int a = lid / gn, b0 = (lid % gn) * kMaxT, nt = min(kMaxT, n - b0);
      const Tri& Q = first ? Lb : Rb;
      const float* __restrict__ Gd = first ? p.g1 : p.g2;
      const int ldg = first ? p.Mp : p.Np;
      float d[kMaxT][4];                                   // D up front (every load ahead of the chunk loop)
#pragma unroll
      for (int t = 0; t < kMaxT; ++t)
        if (t < nt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = 16 * a + g4 + e, col = 16 * (b0 + t) + l15;
            d[t][e] = bload(Q.rs, (row * Q.n + min(col, Q.n - 1)) * 4, 0) * Q.mul;       // (lower part included; stores are masked)
          }
        }
      f32x4 acc[kMaxT];
#pragma unroll
      for (int t = 0; t < kMaxT; ++t) acc[t] = zero4();
      // (a tile b < c of the row contributes zeros: Q is upper triangular, tri_col4 masks; slots past nt repeat the last tile)
      strip_product_bt<kMaxT>(acc, nt, a, b0 + nt, first ? s1 : s2,
                              [&](int c, float (&x)[4]) { row4(Gd, ldg, 16 * a + l15, 16 * c + g4, x); },
                              [&](int t, int c, float (&x)[4]) { tri_col4(Q, 16 * c + g4, 16 * (b0 + min(t, nt - 1)) + l15, x); });
      float* __restrict__ O = first ? p.QlOut : p.QrOut;
#pragma unroll
      for (int t = 0; t < kMaxT; ++t)
        if (t < nt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = 16 * a + g4 + e, col = 16 * (b0 + t) + l15;
            if (row < Q.n && col < Q.n) O[(long)row * Q.n + col] = d[t][e] - acc[t][e];
          }
        }
    }
  }
}

inline int pad16(int x) { return (x + 15) & ~15; }
inline int64_t al256(int64_t x) { return (x + 255) & ~int64_t(255); }

struct SmallWs { float *A, *AT, *Bt, *BtT, *P, *g1, *g2; unsigned* sync; int64_t total; };
SmallWs small_layout(char* base, int M, int N) {
  SmallWs k;
  const int64_t Mp = pad16(M), Np = pad16(N);
  int64_t off = 0;
  auto take = [&](int64_t bytes) { float* p = reinterpret_cast<float*>(base + off); off = al256(off + bytes); return p; };
  k.sync = reinterpret_cast<unsigned*>(take(256));
  k.A = take(Mp * Np * 4); k.AT = take(Mp * Np * 4); k.Bt = take(Mp * Np * 4); k.BtT = take(Mp * Np * 4); k.P = take(Mp * Np * 4);
  k.g1 = take(Mp * Mp * 4); k.g2 = take(Np * Np * 4);
  k.total = off;
  return k;
}

constexpr size_t kUpdateLds = (32 * 256 + 32 * 256 + 16 * 16 * 17 + 8) * sizeof(float);   // tiles, inverted blocks, staging, red

int set_lds_once() {
  static DeviceOnce once;                       // (the attribute is per device)
  if (!once.needed()) return 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kron_small_update), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kUpdateLds) != hipSuccess) return 1;
  once.done();
  return 0;
}

}  // namespace

// Shapes the fused small-layer kernels take (a pure function of the shape: the workspace layout follows it).  The strip chains
// are one workgroup deep in their second dimension, so the rule bounds the tiles of a strip as well as the strips.
bool kron_small_fused(int M, int N) {
  if (M > 512 || N > 512 || M <= 0 || N <= 0) return false;
  const int S = (M + 15) / 16, NT = (N + 15) / 16;
  return S * NT <= 160 && NT <= 16;
}
int64_t kron_small_ws_bytes(int M, int N) { return kron_small_fused(M, N) ? small_layout(nullptr, M, N).total : 0; }

// workgroups of a phase-per-launch call; 1 = the whole call in one workgroup (one launch)
static int small_grid_apply(int S, int NT) {
  const int WS = NT == 1 ? 1 : 4;
  long work = 0;                                   // 16-chunk x tile steps of the longest schedule in one workgroup
  for (int s = 0; s < S; ++s) work += (long)(S - s) * ((NT + 3) / 4) + (long)NT * (NT + 1) / 8 + 2;
  if (WS == 1) work = (work + 3) / 4;
  return work <= 48 ? 1 : (S + 4 / WS - 1) / (4 / WS);
}

int kron_small_apply(const float* Ql, const float* Qr, const float* G, float* out, int M, int N, void* scratch, hipStream_t st) {
  const SmallWs k = small_layout(static_cast<char*>(scratch), M, N);
  SmallApply p;
  p.Ql = Ql; p.Qr = Qr; p.G = G; p.out = out; p.XT = k.AT;
  p.M = M; p.N = N; p.Mp = pad16(M); p.Np = pad16(N); p.S = p.Mp / 16; p.NT = p.Np / 16; p.WS = p.NT == 1 ? 1 : 4;
  const int grid = small_grid_apply(p.S, p.NT);
  const size_t lds = (size_t)(p.NT > 4 ? p.NT : 4) * 256 * 4;
  if (grid == 1) {
    hipLaunchKernelGGL(k_kron_small_apply, dim3(1), dim3(kTh), lds, st, p, 3);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(k_kron_small_apply, dim3(grid), dim3(kTh), lds, st, p, 1);
  if (hipGetLastError() != hipSuccess) return 1;
  hipLaunchKernelGGL(k_kron_small_apply, dim3(grid), dim3(kTh), lds, st, p, 2);
  return (int)hipGetLastError();
}

int kron_small_update(const float* Ql, const float* Qr, const float* dX, const float* dG, float* QlOut, float* QrOut, int M, int N,
                      float step, float tiny, void* scratch, hipStream_t st) {
  if (set_lds_once()) return 1;
  const SmallWs k = small_layout(static_cast<char*>(scratch), M, N);
  SmallUpdate p;
  p.Ql = Ql; p.Qr = Qr; p.dX = dX; p.dG = dG; p.QlOut = QlOut; p.QrOut = QrOut;
  p.A = k.A; p.AT = k.AT; p.Bt = k.Bt; p.BtT = k.BtT; p.P = k.P; p.g1 = k.g1; p.g2 = k.g2; p.sync = k.sync;
  p.M = M; p.N = N; p.Mp = pad16(M); p.Np = pad16(N); p.S = p.Mp / 16; p.NT = p.Np / 16; p.WS = p.NT == 1 ? 1 : 4;
  p.step = step; p.tiny = tiny;
  const int S = p.S, NT = p.NT;
  // one workgroup when its whole schedule is short: strips, the column solve, the row tasks of the gradients and of the updates
  int t3 = 0, t4 = S * ((S + kMaxT - 1) / kMaxT) + NT * ((NT + kMaxT - 1) / kMaxT);
  for (int a = 0; a < S; ++a) t3 += (S - a + kMaxT - 1) / kMaxT;
  for (int a = 0; a < NT; ++a) t3 += (NT - a + kMaxT - 1) / kMaxT;
  const bool one = small_grid_apply(S, NT) == 1 && t4 <= 32;
  if (one) {
    hipLaunchKernelGGL(k_kron_small_update, dim3(1), dim3(kTh), kUpdateLds, st, p, 15);
    return (int)hipGetLastError();
  }
  const int groups = 4 / p.WS;
  const int g1 = (S + groups - 1) / groups + NT;                           // strips of A + column strips of P
  const int g2 = (S + groups - 1) / groups;
  const int g3 = (t3 + 3) / 4, g4 = (t4 + 3) / 4;
  hipLaunchKernelGGL(k_kron_small_update, dim3(g1), dim3(kTh), kUpdateLds, st, p, 1);
  if (hipGetLastError() != hipSuccess) return 1;
  hipLaunchKernelGGL(k_kron_small_update, dim3(g2), dim3(kTh), kUpdateLds, st, p, 2);
  if (hipGetLastError() != hipSuccess) return 1;
  hipLaunchKernelGGL(k_kron_small_update, dim3(g3), dim3(kTh), kUpdateLds, st, p, 4);
  if (hipGetLastError() != hipSuccess) return 1;
  hipLaunchKernelGGL(k_kron_small_update, dim3(g4), dim3(kTh), kUpdateLds, st, p, 8);
  return (int)hipGetLastError();
}

}  // namespace psgdk
