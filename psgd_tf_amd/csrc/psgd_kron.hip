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
#include "psgd_hip.h"

namespace psgdk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 16, LD = 80;   // LDS row pitch 80 floats: conflict-free MFMA reads
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
  long c_cs;
};

enum { KLO_M = 1, KHI_M = 2, KLO_N = 4, KHI_N = 8 };

__device__ __forceinline__ void stage_tiles(float (*As)[LD], float (*Bs)[LD], const float* A, long a_rs, long a_cs,
                                            const float* B, long b_rs, long b_cs, int M, int N, int K, int m0,
                                            int n0, int k0, float a_mul) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int t = 0; t < (BM * BK) / kThreads; ++t) {
    const int e = tid + kThreads * t;
    int m, k;
    if (a_cs == 1) { k = e % BK; m = e / BK; } else { m = e % BM; k = e / BM; }
    const int gm = m0 + m, gk = k0 + k;
    As[k][m] = (gm < M && gk < K) ? A[gm * a_rs + gk * a_cs] * a_mul : 0.0f;
  }
#pragma unroll
  for (int t = 0; t < (BN * BK) / kThreads; ++t) {
    const int e = tid + kThreads * t;
    int n, k;
    if (b_cs == 1) { n = e % BN; k = e / BN; } else { k = e % BK; n = e / BK; }
    const int gn = n0 + n, gk = k0 + k;
    Bs[k][n] = (gn < N && gk < K) ? B[gk * b_rs + gn * b_cs] : 0.0f;
  }
}

__global__ __launch_bounds__(kThreads) void k_gemm_f32(GemmArgs g) {
  __shared__ float As[BK][LD];
  __shared__ float Bs[BK][LD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // upper-triangular outputs: tiles strictly below the diagonal are all zero
  const bool tri_skip = (g.epi == EPI_TRIU_MAX) && (m0 >= n0 + BN);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float a_mul = 1.0f;
  if (g.scale_max) a_mul = g.step / (g.scale_max[0] + g.tiny);

  if (!tri_skip) {
    for (int pass = 0; pass < 2; ++pass) {
      const float* A = pass ? g.A2 : g.A;
      const float* B = pass ? g.B2 : g.B;
      if (!A) break;
      const long a_rs = pass ? g.a2_rs : g.a_rs, a_cs = pass ? g.a2_cs : g.a_cs;
      const long b_rs = pass ? g.b2_rs : g.b_rs, b_cs = pass ? g.b2_cs : g.b_cs;
      const int K = pass ? g.K2 : g.K;
      const float mul = pass ? -a_mul : a_mul;
      const int km = pass ? g.kmode2 : g.kmode;
      int klo = 0, khi = K;
      if (km & KLO_M) klo = max(klo, m0);
      if (km & KLO_N) klo = max(klo, n0);
      if (km & KHI_M) khi = min(khi, m0 + BM);
      if (km & KHI_N) khi = min(khi, n0 + BN);
      klo = (klo / BK) * BK;
      for (int k0 = klo; k0 < khi; k0 += BK) {
        __syncthreads();
        stage_tiles(As, Bs, A, a_rs, a_cs, B, b_rs, b_cs, g.M, g.N, K, m0, n0, k0, mul);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
          const int kr = kk * 4 + (lane >> 4);
          float a[2], b[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) a[i] = As[kr][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
          for (int j = 0; j < 2; ++j) b[j] = Bs[kr][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  float vmax = 0.0f;
  const long ccs = g.c_cs ? g.c_cs : 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + e;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][e];
          if (g.epi == EPI_TRIU_MAX) {
            v = (col >= row) ? v : 0.0f;
            vmax = fmaxf(vmax, fabsf(v));
          } else if (g.epi == EPI_D_MINUS) {
            v = g.D[(long)row * g.ldd + col * ccs] - v;
          }
          g.C[(long)row * g.ldc + col * ccs] = v;
        }
      }
  if (g.epi == EPI_TRIU_MAX) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
    if (lane == 0 && vmax > 0.0f) atomicMax(reinterpret_cast<int*>(g.maxout), __float_as_int(vmax));
  }
}

// ---------------------------------------------------------------------------------------------
// Large-tile variant of k_gemm_f32 (same GemmArgs, same epilogues): 128 x 128 x 16 tile, wave tile
// 64 x 64 = 4 x 4 MFMA tiles (16 MFMAs per 8 LDS operand reads), LDS double-buffered with the next
// K tile prefetched into registers while the current one is multiplied (one barrier per K tile),
// 16-byte global loads on interior tiles whenever the operand's contiguous dimension allows it.
// LDS pitch 144 floats: the 4 k-rows of an MFMA operand read land 16 banks apart (conflict-free).
constexpr int GM = 128, GN = 128, GK = 16, GP = 144;

struct TileSrc {
  const float* P; long rs, cs;   // element (x, k) at P[x*rs + k*cs], x = m (A) or n (B)
  int X, K;                      // extents
};

// loads one 128 x 16 operand tile into 8 registers per thread
__device__ __forceinline__ void g2r_tile(const TileSrc& t, int x0, int k0, int khi, float mul, float (&r)[8]) {
  const int tid = threadIdx.x;
  const bool interior = (x0 + GM <= t.X) && (k0 + GK <= khi);
  if (interior && t.cs == 1 && (t.rs & 3) == 0 && ((reinterpret_cast<uintptr_t>(t.P) & 15) == 0)) {
    // K-contiguous: two float4 along k per thread
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + kThreads * u, row = f >> 2, k4 = f & 3;
      const f32x4 v = *reinterpret_cast<const f32x4*>(t.P + (long)(x0 + row) * t.rs + k0 + 4 * k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[4 * u + e] = v[e] * mul;
    }
  } else if (interior && t.rs == 1 && (t.cs & 3) == 0 && ((reinterpret_cast<uintptr_t>(t.P) & 15) == 0)) {
    // X-contiguous: two float4 along x per thread
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + kThreads * u, k = f >> 5, x4 = f & 31;
      const f32x4 v = *reinterpret_cast<const f32x4*>(t.P + (long)(k0 + k) * t.cs + x0 + 4 * x4);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[4 * u + e] = v[e] * mul;
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + kThreads * u;
      int x, k;
      if (t.cs == 1) { k = e % GK; x = e / GK; } else { x = e % GM; k = e / GM; }
      const int gx = x0 + x, gk = k0 + k;
      r[u] = (gx < t.X && gk < khi) ? t.P[(long)gx * t.rs + (long)gk * t.cs] * mul : 0.0f;
    }
  }
}

// writes the registers of g2r_tile into an LDS tile T[k][x]; must take the same branch
__device__ __forceinline__ void r2s_tile(const TileSrc& t, int x0, int k0, int khi, const float (&r)[8], float (*T)[GP]) {
  const int tid = threadIdx.x;
  const bool interior = (x0 + GM <= t.X) && (k0 + GK <= khi);
  if (interior && t.cs == 1 && (t.rs & 3) == 0 && ((reinterpret_cast<uintptr_t>(t.P) & 15) == 0)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + kThreads * u, row = f >> 2, k4 = f & 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) T[4 * k4 + e][row] = r[4 * u + e];
    }
  } else if (interior && t.rs == 1 && (t.cs & 3) == 0 && ((reinterpret_cast<uintptr_t>(t.P) & 15) == 0)) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + kThreads * u, k = f >> 5, x4 = f & 31;
      *reinterpret_cast<f32x4*>(&T[k][4 * x4]) = f32x4{r[4 * u], r[4 * u + 1], r[4 * u + 2], r[4 * u + 3]};
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + kThreads * u;
      int x, k;
      if (t.cs == 1) { k = e % GK; x = e / GK; } else { x = e % GM; k = e / GM; }
      T[k][x] = r[u];
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_gemm_f32_big(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][GK][GP];
  __shared__ __attribute__((aligned(16))) float Bs[2][GK][GP];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
  const bool tri_skip = (g.epi == EPI_TRIU_MAX) && (m0 >= n0 + GN);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float a_mul = 1.0f;
  if (g.scale_max) a_mul = g.step / (g.scale_max[0] + g.tiny);

  // K ranges of the (up to) two passes
  int klo[2] = {0, 0}, khi[2] = {0, 0}, nk[2] = {0, 0};
  if (!tri_skip) {
    for (int p = 0; p < 2; ++p) {
      if (p == 1 && !g.A2) break;
      const int K = p ? g.K2 : g.K, km = p ? g.kmode2 : g.kmode;
      int lo = 0, hi = K;
      if (km & KLO_M) lo = max(lo, m0);
      if (km & KLO_N) lo = max(lo, n0);
      if (km & KHI_M) hi = min(hi, m0 + GM);
      if (km & KHI_N) hi = min(hi, n0 + GN);
      lo = (lo / GK) * GK;
      klo[p] = lo; khi[p] = hi; nk[p] = hi > lo ? (hi - lo + GK - 1) / GK : 0;
    }
  }
  const int ntile = nk[0] + nk[1];
  const TileSrc ta[2] = {{g.A, g.a_rs, g.a_cs, g.M, g.K}, {g.A2, g.a2_rs, g.a2_cs, g.M, g.K2}};
  const TileSrc tb[2] = {{g.B, g.b_cs, g.b_rs, g.N, g.K}, {g.B2, g.b2_cs, g.b2_rs, g.N, g.K2}};   // (n, k) view of B

  float ra[8], rb[8];
  auto fetch = [&](int t) {
    const int p = t < nk[0] ? 0 : 1;
    const int k0 = klo[p] + (p ? t - nk[0] : t) * GK;
    g2r_tile(ta[p], m0, k0, khi[p], p ? -a_mul : a_mul, ra);
    g2r_tile(tb[p], n0, k0, khi[p], 1.0f, rb);
  };
  auto commit = [&](int t, int buf) {
    const int p = t < nk[0] ? 0 : 1;
    const int k0 = klo[p] + (p ? t - nk[0] : t) * GK;
    r2s_tile(ta[p], m0, k0, khi[p], ra, As[buf]);
    r2s_tile(tb[p], n0, k0, khi[p], rb, Bs[buf]);
  };

  if (ntile > 0) {
    fetch(0);
    commit(0, 0);
  }
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntile) fetch(t + 1);
#pragma unroll
    for (int kk = 0; kk < GK / 4; ++kk) {
      const int kr = kk * 4 + (lane >> 4);
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[buf][kr][wm * 64 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[buf][kr][wn * 64 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < ntile) commit(t + 1, buf ^ 1);
    __syncthreads();
  }

  float vmax = 0.0f;
  const long ccs = g.c_cs ? g.c_cs : 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + e;
        const int col = n0 + wn * 64 + j * 16 + (lane & 15);
        if (row < g.M && col < g.N) {
          float v = acc[i][j][e];
          if (g.epi == EPI_TRIU_MAX) {
            v = (col >= row) ? v : 0.0f;
            vmax = fmaxf(vmax, fabsf(v));
          } else if (g.epi == EPI_D_MINUS) {
            v = g.D[(long)row * g.ldd + col * ccs] - v;
          }
          g.C[(long)row * g.ldc + col * ccs] = v;
        }
      }
  if (g.epi == EPI_TRIU_MAX) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
    if (lane == 0 && vmax > 0.0f) atomicMax(reinterpret_cast<int*>(g.maxout), __float_as_int(vmax));
  }
}

// Solve  y[i,:] Q = x[i,:]  for nvec independent vectors i, Q upper-triangular [n,n] row-major:
//   y[i,j] = (x[i,j] - sum_{k<j} y[i,k] Q[k,j]) / Q[j,j]
// Element (i,j) of X / Y lives at  i*si + j*sj.  With (si,sj) = (ld,1) this is the right solve
// Y Q = X on row-major [nvec,n]; with (si,sj) = (1,ld) it is Q'Y = X on row-major [n,nvec]
// (tf.linalg.triangular_solve(Q, X, lower=False, adjoint=True), psgd.py:174).
// Q has leading dimension ldq (a diagonal block of a larger factor can be passed); X may alias Y.
__global__ __launch_bounds__(kThreads) void k_trsm_ut(const float* __restrict__ Q, int n, int ldq, const float* X,
                                                      float* Y, int nvec, long si, long sj) {
  __shared__ float red[4][64][33];
  __shared__ float Qd[32][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int v0 = blockIdx.x * 64;
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
        const float x = (vok && j < jw) ? X[v * si + (long)(j0 + j) * sj] : 0.0f;
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

// rho = sqrt(max diag Ql / max diag Qr); QlS = Ql / rho; QrS = rho Qr      (psgd.py:166-170)
__global__ __launch_bounds__(kThreads) void k_kron_balance(const float* __restrict__ Ql, const float* __restrict__ Qr,
                                                           int M, int N, float* QlS, float* QrS) {
  __shared__ float red[2][4];
  float ml = -INFINITY, mr = -INFINITY;
  for (int i = threadIdx.x; i < M; i += kThreads) ml = fmaxf(ml, Ql[(long)i * M + i]);
  for (int i = threadIdx.x; i < N; i += kThreads) mr = fmaxf(mr, Qr[(long)i * N + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ml = fmaxf(ml, __shfl_down(ml, off, 64));
    mr = fmaxf(mr, __shfl_down(mr, off, 64));
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = ml; red[1][w] = mr; }
  __syncthreads();
  ml = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  mr = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  const float rho = sqrtf(ml / mr);
  const long nl = (long)M * M, nr = (long)N * N;
  const long tid = (long)blockIdx.x * kThreads + threadIdx.x, nth = (long)gridDim.x * kThreads;
  for (long i = tid; i < nl; i += nth) QlS[i] = Ql[i] / rho;
  for (long i = tid; i < nr; i += nth) QrS[i] = rho * Qr[i];
}

// ------------------------------------------------------------- host side ----
static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

struct KronWs {
  float *scal, *QlS, *QrS, *T, *A, *X1, *Bt, *g1, *g2;
  int64_t total;
};

static KronWs kron_layout(char* base, int M, int N) {
  KronWs k;
  const int64_t mm = (int64_t)M * M * 4, nn = (int64_t)N * N * 4, mn = (int64_t)M * N * 4;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { float* p = reinterpret_cast<float*>(base + off); off = align256(off + bytes); return p; };
  k.scal = take(256);
  k.QlS = take(mm); k.QrS = take(nn);
  k.T = take(mn); k.A = take(mn); k.X1 = take(mn); k.Bt = take(mn);
  k.g1 = take(mm); k.g2 = take(nn);
  k.total = off;
  return k;
}

static int g_force_gemm = 0;   // 0 auto, 1 always 64-tile kernel, 2 always 128-tile kernel (experiments)

static int launch_gemm(const GemmArgs& g, hipStream_t st) {
  // the 128-tile kernel needs enough tiles to fill the chip; small problems keep 64 x 64 tiles
  const long big_tiles = (long)((g.N + GN - 1) / GN) * ((g.M + GM - 1) / GM);
  const bool use_big = g_force_gemm == 2 || (g_force_gemm == 0 && big_tiles >= 64);
  if (use_big) {
    dim3 grid((g.N + GN - 1) / GN, (g.M + GM - 1) / GM);
    hipLaunchKernelGGL(k_gemm_f32_big, grid, dim3(kThreads), 0, st, g);
  } else {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(kThreads), 0, st, g);
  }
  return (int)hipGetLastError();
}

// C[M,N] = op(A) op(B); ta/tb: operand stored transposed (row-major [K,M] / [N,K])
static GemmArgs gemm_args(const float* A, int lda, bool ta, const float* B, int ldb, bool tb, float* C, int ldc, int M,
                          int N, int K) {
  GemmArgs g = {};
  g.A = A; g.a_rs = ta ? 1 : lda; g.a_cs = ta ? lda : 1;
  g.B = B; g.b_rs = tb ? 1 : ldb; g.b_cs = tb ? ldb : 1;
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.epi = EPI_STORE;
  return g;
}

// Blocked solve of  y[i,:] Q = x[i,:]  (see k_trsm_ut) for large n, right-looking so that the
// MFMA GEMMs are wide (N = all remaining columns, K = block width):
//   Y <- X
//   for each 256-wide column block jb:   Y[:, jb] <- Y[:, jb] Q[jb, jb]^-1      (substitution kernel, in place)
//                                        Y[:, >jb] -= Y[:, jb] Q[jb, >jb]      (one GEMM)
// X and Y are dense nvec*n arrays (either orientation), so the initial copy is one memcpy.
constexpr int kTrsmBlock = 256;

static int trsm_ut(const float* Q, int n, const float* X, float* Y, int nvec, long si, long sj, hipStream_t st) {
  const dim3 grid((nvec + 63) / 64);
  if (n <= 2 * kTrsmBlock) {
    hipLaunchKernelGGL(k_trsm_ut, grid, dim3(kThreads), 0, st, Q, n, n, X, Y, nvec, si, sj);
    return (int)hipGetLastError();
  }
  if (hipMemcpyAsync(Y, X, (size_t)nvec * n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
  for (int j0 = 0; j0 < n; j0 += kTrsmBlock) {
    const int jw = (n - j0 < kTrsmBlock) ? (n - j0) : kTrsmBlock;
    float* Yb = Y + (long)j0 * sj;
    hipLaunchKernelGGL(k_trsm_ut, grid, dim3(kThreads), 0, st, Q + (long)j0 * n + j0, jw, n, Yb, Yb, nvec, si, sj);
    int e = (int)hipGetLastError();
    if (e) return e;
    const int rest = n - j0 - jw;
    if (rest > 0) {
      GemmArgs g = {};
      g.A = Yb; g.a_rs = si; g.a_cs = sj;                              // A(i,k) = Y[i, j0+k]
      g.B = Q + (long)j0 * n + j0 + jw; g.b_rs = n; g.b_cs = 1;        // B(k,j) = Q[j0+k, j0+jw+j]
      float* Yr = Y + (long)(j0 + jw) * sj;
      g.C = Yr; g.ldc = si; g.c_cs = sj;
      g.D = Yr; g.ldd = si;                                            // in place: C = C - A B
      g.M = nvec; g.N = rest; g.K = jw;
      g.epi = EPI_D_MINUS;
      e = launch_gemm(g, st);
      if (e) return e;
    }
  }
  return 0;
}

}  // namespace psgdk

using namespace psgdk;

#define KRON_LAUNCH(expr)                     \
  do {                                        \
    if ((expr) != 0) return PSGD_ERR_LAUNCH;  \
  } while (0)

extern "C" {

int psgd_kron_set_tuning(int key, int value) {
  if (key == 0) { g_force_gemm = value; return PSGD_OK; }
  return PSGD_ERR_BAD_ARG;
}

int64_t psgd_kron_dd_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return PSGD_ERR_BAD_ARG;
  return kron_layout(nullptr, M, N).total;
}

int psgd_kron_dd_apply_f32(const float* Ql, const float* Qr, const float* G, float* out, int M, int N, void* ws,
                           int64_t ws_bytes, void* stream) {
  if (!Ql || !Qr || !G || !out) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0) return PSGD_ERR_SHAPE;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < kron_layout(nullptr, M, N).total)
    return PSGD_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KronWs k = kron_layout(static_cast<char*>(ws), M, N);
  if (M < N) {                                                               // psgd.py:189-190
    GemmArgs g1 = gemm_args(Ql, M, true, Ql, M, false, k.g1, M, M, M, M);                    // Ql'Ql
    g1.kmode = KHI_M | KHI_N;
    KRON_LAUNCH(launch_gemm(g1, st));
    KRON_LAUNCH(launch_gemm(gemm_args(k.g1, M, false, G, N, false, k.T, N, M, N, M), st));    // (.) G
    GemmArgs g3 = gemm_args(k.T, N, false, Qr, N, true, k.A, N, M, N, N);                     // (.) Qr'
    g3.kmode = KLO_N;
    KRON_LAUNCH(launch_gemm(g3, st));
    GemmArgs g4 = gemm_args(k.A, N, false, Qr, N, false, out, N, M, N, N);                    // (.) Qr
    g4.kmode = KHI_N;
    KRON_LAUNCH(launch_gemm(g4, st));
  } else {                                                                   // psgd.py:191-192
    GemmArgs g1 = gemm_args(Qr, N, true, Qr, N, false, k.g2, N, N, N, N);                    // Qr'Qr
    g1.kmode = KHI_M | KHI_N;
    KRON_LAUNCH(launch_gemm(g1, st));
    KRON_LAUNCH(launch_gemm(gemm_args(G, N, false, k.g2, N, false, k.T, N, M, N, N), st));    // G (.)
    GemmArgs g3 = gemm_args(Ql, M, false, k.T, N, false, k.A, N, M, N, M);                    // Ql (.)
    g3.kmode = KLO_M;
    KRON_LAUNCH(launch_gemm(g3, st));
    GemmArgs g4 = gemm_args(Ql, M, true, k.A, N, false, out, N, M, N, M);                     // Ql' (.)
    g4.kmode = KHI_M;
    KRON_LAUNCH(launch_gemm(g4, st));
  }
  return PSGD_OK;
}

int psgd_kron_dd_update_f32(const float* Ql, const float* Qr, const float* dX, const float* dG, float* QlOut,
                            float* QrOut, int M, int N, float step, float tiny, void* ws, int64_t ws_bytes,
                            void* stream) {
  if (!Ql || !Qr || !dX || !dG || !QlOut || !QrOut) return PSGD_ERR_BAD_ARG;
  if (M <= 0 || N <= 0) return PSGD_ERR_SHAPE;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) || ws_bytes < kron_layout(nullptr, M, N).total)
    return PSGD_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KronWs k = kron_layout(static_cast<char*>(ws), M, N);
  if (hipMemsetAsync(k.scal, 0, 256, st) != hipSuccess) return PSGD_ERR_LAUNCH;
  // K0: balance
  {
    const long tot = (long)M * M + (long)N * N;
    int grid = (int)((tot + kThreads - 1) / kThreads);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_kron_balance, dim3(grid), dim3(kThreads), 0, st, Ql, Qr, M, N, k.QlS, k.QrS);
    KRON_LAUNCH((int)hipGetLastError());
  }
  // K1: A = QlS (dG QrS')
  {
    GemmArgs g = gemm_args(dG, N, false, k.QrS, N, true, k.T, N, M, N, N);
    g.kmode = KLO_N;
    KRON_LAUNCH(launch_gemm(g, st));
  }
  {
    GemmArgs g = gemm_args(k.QlS, M, false, k.T, N, false, k.A, N, M, N, M);
    g.kmode = KLO_M;
    KRON_LAUNCH(launch_gemm(g, st));
  }
  // K2: X1 = dX QrS^-1 (rows independent), Bt = QlS^-T X1 (columns independent)
  KRON_LAUNCH(trsm_ut(k.QrS, N, dX, k.X1, M, (long)N, 1L, st));
  KRON_LAUNCH(trsm_ut(k.QlS, M, k.X1, k.Bt, N, 1L, (long)N, st));
  // K3/K5: grad1 = triu(A A' - Bt Bt'), max|grad1| -> scal[0]
  {
    GemmArgs g = gemm_args(k.A, N, false, k.A, N, true, k.g1, M, M, M, N);
    g.A2 = k.Bt; g.a2_rs = N; g.a2_cs = 1; g.B2 = k.Bt; g.b2_rs = 1; g.b2_cs = N; g.K2 = N;
    g.epi = EPI_TRIU_MAX; g.maxout = k.scal + 0;
    KRON_LAUNCH(launch_gemm(g, st));
  }
  // K4/K5: grad2 = triu(A'A - Bt'Bt), max|grad2| -> scal[1]
  {
    GemmArgs g = gemm_args(k.A, N, true, k.A, N, false, k.g2, N, N, N, M);
    g.A2 = k.Bt; g.a2_rs = 1; g.a2_cs = N; g.B2 = k.Bt; g.b2_rs = N; g.b2_cs = 1; g.K2 = M;
    g.epi = EPI_TRIU_MAX; g.maxout = k.scal + 1;
    KRON_LAUNCH(launch_gemm(g, st));
  }
  // K6: Q - (step_i grad_i) Q
  {
    GemmArgs g = gemm_args(k.g1, M, false, k.QlS, M, false, QlOut, M, M, M, M);
    g.epi = EPI_D_MINUS; g.D = k.QlS; g.ldd = M; g.scale_max = k.scal + 0; g.step = step; g.tiny = tiny;
    g.kmode = KLO_M | KHI_N;
    KRON_LAUNCH(launch_gemm(g, st));
  }
  {
    GemmArgs g = gemm_args(k.g2, N, false, k.QrS, N, false, QrOut, N, N, N, N);
    g.epi = EPI_D_MINUS; g.D = k.QrS; g.ldd = N; g.scale_max = k.scal + 1; g.step = step; g.tiny = tiny;
    g.kmode = KLO_M | KHI_N;
    KRON_LAUNCH(launch_gemm(g, st));
  }
  return PSGD_OK;
}

}  // extern "C"
