// psgd_splu.hip -- C ABI (include/psgd_hip.h) of the sparse-LU preconditioner path
// (update_precond_splu / precond_grad_splu, psgd.py:396-524): the r x r "corner" algebra in fp64
// (L1, U1, four triangular solves, corner gradients, step sizes, the rho balance of :411-417),
// partial reductions, and the host sequencing of the sweeps in splu_kernels.h.
#include "splu_kernels.h"
#include "psgd_hip.h"
#include <math.h>

namespace psgd {

const SpluOps* splu_ops_group0(int r);
const SpluOps* splu_ops_group1(int r);
const SpluOps* splu_ops_group2(int r);
const SpluOps* splu_ops_group3(int r);
const SpluOps* splu_wide_group0(int r);      // ranks 33 .. 64 (round 5): the same kernels on 64-row tiles, one workgroup per CU
const SpluOps* splu_wide_group1(int r);
const SpluOps* splu_wide_group2(int r);
const SpluOps* splu_wide_group3(int r);

const SpluOps* splu_ops_for_rank(int r) {
  if (r < 1 || r > PSGD_SPLU_MAX_RANK) return nullptr;
  switch ((r - 1) / 8) {
    case 0: return splu_ops_group0(r);
    case 1: return splu_ops_group1(r);
    case 2: return splu_ops_group2(r);
    case 3: return splu_ops_group3(r);
    case 4: return splu_wide_group0(r);
    case 5: return splu_wide_group1(r);
    case 6: return splu_wide_group2(r);
    default: return splu_wide_group3(r);
  }
}

constexpr int MR = PSGD_SPLU_MAX_RANK;       // capacity of the corner algebra and of the workspace regions

// ------------------------------------------------------------ workspace ----
// doubles: sums A [MR] | sums B [2 MR] | state vectors (7 x MR) | sums C [r] followed by the fp64 copies of the 4 maxima
// (one contiguous multi-GPU send region for update stage 3)
enum { kSumA = 0, kSumB = MR, kStUg1 = 3 * MR, kStQg1 = 4 * MR, kStIUtx1 = 5 * MR, kStIQtx1 = 6 * MR,
       kStLtQg1 = 7 * MR, kStPg1 = 8 * MR, kStILiQtx1 = 9 * MR, kSumC = 10 * MR, kDoubles = 11 * MR + 8 };
constexpr int64_t kCoefFloats = 9 * MR;
constexpr int64_t kMaxFloats = 8;

struct SpluWs {
  double* dbl; float* coef; float* maxbuf; float* part; float* pmax;
};

static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

static int64_t splu_layout(int64_t N, int r, char* base, SpluWs* w) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) { const int64_t o = off; off = align256(off + bytes); return o; };
  const int64_t o_dbl = take(kDoubles * 8), o_coef = take(kCoefFloats * 4), o_max = take(kMaxFloats * 4);
  const int64_t o_part = take((int64_t)kMaxGrid * 2 * MR * 4), o_pmax = take((int64_t)kMaxGrid * 4 * 4);
  (void)N; (void)r;
  if (w) {
    w->dbl = reinterpret_cast<double*>(base + o_dbl);
    w->coef = reinterpret_cast<float*>(base + o_coef);
    w->maxbuf = reinterpret_cast<float*>(base + o_max);
    w->part = reinterpret_cast<float*>(base + o_part);
    w->pmax = reinterpret_cast<float*>(base + o_pmax);
  }
  return off;
}

// --------------------------------------------------------- small kernels ---
// sums[id] = sum_b part[id][b]: one wave per element, fp64 tree, fixed order (cf. k_reduce_sum_t in psgd_uvd.hip)
__global__ __launch_bounds__(kThreads) void k_splu_reduce_sum(const float* __restrict__ part, int G, int L,
                                                              double* __restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (id >= L) return;
  const float* p = part + (long)id * G;
  float x[kMaxGrid / 64];
#pragma unroll
  for (int u = 0; u < kMaxGrid / 64; ++u) x[u] = p[min(lane + 64 * u, G - 1)];      // clamped, unconditional: a guarded load
  double s = 0.0;                                                                    // compiles to load-then-wait, one latency each
#pragma unroll
  for (int u = 0; u < kMaxGrid / 64; ++u) s += (lane + 64 * u < G) ? (double)x[u] : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) sums[id] = s;
}

// out[set] = max_b part[set*G + b]   (signed: the balance of psgd.py:411-412 takes max, not max|.|)
__global__ __launch_bounds__(kThreads) void k_splu_reduce_max(const float* __restrict__ part, int G,
                                                              float* __restrict__ out, double* __restrict__ outd) {
  __shared__ float red[kWavesPerBlock];
  const float* p = part + (long)blockIdx.x * G;
  float v = -INFINITY;
  for (int b = threadIdx.x; b < G; b += kThreads) v = nmaxf(v, p[b]);
  block_max_store<true>(v, red, out + blockIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) outd[blockIdx.x] = (double)out[blockIdx.x];
}

// multi-GPU exchange, second half: fold the all-gathered send regions ([world][count] doubles) in rank order, the first
// nsum entries by +, the rest by max (cf. k_fold_gathered in psgd_uvd.hip)
__global__ void k_splu_fold_gathered(const double* __restrict__ gathered, int world, int count, int nsum,
                                     double* __restrict__ dst, float* __restrict__ maxdst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double a = gathered[i];
  if (i < nsum) {
    for (int k = 1; k < world; ++k) a += gathered[(long)k * count + i];
  } else {
    for (int k = 1; k < world; ++k) a = nmax(a, gathered[(long)k * count + i]);
    maxdst[i - nsum] = (float)a;
  }
  dst[i] = a;
}

// ----------------------------------------------------- r x r corner (fp64) -
// One 256-thread block; L1 and U1 live in LDS as doubles.  Helpers are block-cooperative.
struct Corner {
  double L1[MR][MR + 1];
  double U1[MR][MR + 1];
  double v[8][MR];   // scratch vectors
};

__device__ __forceinline__ void corner_load(Corner& c, const float* L12, const float* U12, long ldu, int r) {
  for (int e = threadIdx.x; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e % r;
    c.L1[i][j] = (double)L12[(long)i * r + j];
    c.U1[i][j] = (double)U12[(long)i * ldu + j];
  }
  __syncthreads();
}

// y = M x (trans = false) or M' x (trans = true); y must not alias x
__device__ __forceinline__ void corner_matvec(const double (*M)[MR + 1], bool trans, const double* x, double* y, int r) {
  const int i = threadIdx.x;
  if (i < r) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s += (trans ? M[j][i] : M[i][j]) * x[j];
    y[i] = s;
  }
  __syncthreads();
}

// in place: b <- T^-1 b with T = M (trans = false) or M' (trans = true); `lower` says which triangle of M is used
// (tf.linalg.triangular_solve ignores the other one)
__device__ __forceinline__ void corner_trisolve(const double (*M)[MR + 1], bool lower, bool trans, double* b, int r) {
  const bool fwd = (lower != trans);
  const int j = threadIdx.x;
  for (int step = 0; step < r; ++step) {
    const int p = fwd ? step : r - 1 - step;
    if (j == p) b[p] = b[p] / M[p][p];
    __syncthreads();
    const bool rem = fwd ? (j > p && j < r) : (j < p);
    if (rem) b[j] -= (trans ? M[p][j] : M[j][p]) * b[p];
    __syncthreads();
  }
}

// after sweep 1 of either path: Ug1 = U1 g1 + U2 g2 (:430/:506), Qg1 = L1 Ug1 (:433/:509);
// update only: iUtx1 = U1^-T dx1 (:436).  coef <- [Ug1 | iUtx1]
__global__ __launch_bounds__(kThreads) void k_splu_corner1(const float* L12, const float* U12, long ldu, int r,
                                                           const float* g1, const float* x1 /* null for apply */,
                                                           double* dbl, float* coef) {
  __shared__ Corner c;
  corner_load(c, L12, U12, ldu, r);
  const int t = threadIdx.x;
  if (t < r) c.v[0][t] = (double)g1[t];
  __syncthreads();
  corner_matvec(c.U1, false, c.v[0], c.v[1], r);
  if (t < r) {
    c.v[1][t] += dbl[kSumA + t];
    dbl[kStUg1 + t] = c.v[1][t];
    coef[t] = (float)c.v[1][t];
  }
  __syncthreads();
  corner_matvec(c.L1, false, c.v[1], c.v[2], r);
  if (t < r) dbl[kStQg1 + t] = c.v[2][t];
  if (x1) {
    if (t < r) c.v[3][t] = (double)x1[t];
    __syncthreads();
    corner_trisolve(c.U1, /*lower=*/false, /*trans=*/true, c.v[3], r);
    if (t < r) {
      dbl[kStIUtx1 + t] = c.v[3][t];
      coef[r + t] = (float)c.v[3][t];
    }
  }
}

// apply, after sweep 2: LtQg1 = L1' Qg1 + L2' Qg2 (:512); out1 = U1' LtQg1 (:515).  coef <- LtQg1
__global__ __launch_bounds__(kThreads) void k_splu_corner_apply2(const float* L12, const float* U12, long ldu, int r,
                                                                 double* dbl, float* coef, float* out1) {
  __shared__ Corner c;
  corner_load(c, L12, U12, ldu, r);
  const int t = threadIdx.x;
  if (t < r) c.v[0][t] = dbl[kStQg1 + t];
  __syncthreads();
  corner_matvec(c.L1, true, c.v[0], c.v[1], r);
  if (t < r) {
    c.v[1][t] += dbl[kSumB + t];
    coef[t] = (float)c.v[1][t];
  }
  __syncthreads();
  corner_matvec(c.U1, true, c.v[1], c.v[2], r);
  if (t < r) out1[t] = (float)c.v[2][t];
}

// update, after sweep 2: iQtx1 (:440), LtQg1 (:442), Pg1 (:445), iLiQtx1 (:448).
// coef <- [LtQg1 | iLiQtx1 | Qg1 | iQtx1 | Pg1 | dx1 | Ug1 | iUtx1]
__global__ __launch_bounds__(kThreads) void k_splu_corner_upd2(const float* L12, const float* U12, long ldu, int r,
                                                               const float* x1, double* dbl, float* coef) {
  __shared__ Corner c;
  corner_load(c, L12, U12, ldu, r);
  const int t = threadIdx.x;
  if (t < r) {
    c.v[0][t] = dbl[kStIUtx1 + t] - dbl[kSumB + r + t];   // iUtx1 - L2' iQtx2
    c.v[1][t] = dbl[kStQg1 + t];
  }
  __syncthreads();
  corner_trisolve(c.L1, /*lower=*/true, /*trans=*/true, c.v[0], r);   // iQtx1
  corner_matvec(c.L1, true, c.v[1], c.v[2], r);                      // L1' Qg1
  if (t < r) c.v[2][t] += dbl[kSumB + t];                             // LtQg1
  __syncthreads();
  corner_matvec(c.U1, true, c.v[2], c.v[3], r);                      // Pg1
  if (t < r) c.v[4][t] = c.v[0][t];
  __syncthreads();
  corner_trisolve(c.L1, /*lower=*/true, /*trans=*/false, c.v[4], r);  // iLiQtx1
  if (t < r) {
    dbl[kStIQtx1 + t] = c.v[0][t];
    dbl[kStLtQg1 + t] = c.v[2][t];
    dbl[kStPg1 + t] = c.v[3][t];
    dbl[kStILiQtx1 + t] = c.v[4][t];
    coef[t] = (float)c.v[2][t];
    coef[r + t] = (float)c.v[4][t];
    coef[2 * r + t] = (float)c.v[1][t];
    coef[3 * r + t] = (float)c.v[0][t];
    coef[4 * r + t] = (float)c.v[3][t];
    coef[5 * r + t] = x1[t];
    coef[6 * r + t] = (float)dbl[kStUg1 + t];
    coef[7 * r + t] = (float)dbl[kStIUtx1 + t];
  }
}

__device__ __forceinline__ double block_max_f64(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v = nmax(v, __shfl_down(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = nmax(m, red[k]);
  return m;
}

// update, after sweep 3: iPx1 (:452); rho (:411-413); corner gradients, both step sizes, new L1 (:455-463) and new
// U1 (:468-476) written out; coef <- [LtQg1 | iLiQtx1 | a | b | c | e | Ug1 | iUtx1 | sL sU rho 1/rho] for sweep 4.
__global__ __launch_bounds__(kThreads) void k_splu_corner_upd3(const float* L12, const float* U12, long ldu, int r,
                                                               const float* x1, const float* g1, int has_tail,
                                                               float step, float tiny, double* dbl,
                                                               const float* maxbuf, float* coef, float* L12o,
                                                               float* U12o) {
  __shared__ Corner c;
  __shared__ double red[kWavesPerBlock];
  __shared__ double G[MR][MR + 1];
  corner_load(c, L12, U12, ldu, r);
  const int t = threadIdx.x;
  // v0 = iPx1, v1 = Qg1, v2 = iQtx1, v3 = Pg1, v4 = dx1, v5 = dg1
  if (t < r) {
    c.v[0][t] = dbl[kStILiQtx1 + t] - dbl[kSumC + t];
    c.v[1][t] = dbl[kStQg1 + t];
    c.v[2][t] = dbl[kStIQtx1 + t];
    c.v[3][t] = dbl[kStPg1 + t];
    c.v[4][t] = (double)x1[t];
    c.v[5][t] = (double)g1[t];
  }
  __syncthreads();
  corner_trisolve(c.U1, /*lower=*/false, /*trans=*/false, c.v[0], r);

  // dynamic-range balance (:411-417): signed maxima, as written
  double dl = -INFINITY, du = -INFINITY;
  if (t < r) { dl = c.L1[t][t]; du = c.U1[t][t]; }
  double max_l = block_max_f64(dl, red);
  double max_u = block_max_f64(du, red);
  if (has_tail) { max_l = nmax(max_l, (double)maxbuf[2]); max_u = nmax(max_u, (double)maxbuf[3]); }
  const float rho_f = sqrtf((float)max_l / (float)max_u);
  const double rho = (double)rho_f, irho = 1.0 / rho;

  // ---- L: grad1 = tril(Qg1 Qg1' - iQtx1 iQtx1'), step0, newL1 = L1s - step0 grad1 L1s, L1s = L1 / rho
  double m = 0.0;
  for (int e = t; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e % r;
    const double g = (j <= i) ? c.v[1][i] * c.v[1][j] - c.v[2][i] * c.v[2][j] : 0.0;
    G[i][j] = g;
    m = nmax(m, fabs(g));
  }
  m = block_max_f64(m, red);
  if (has_tail) m = nmax(m, (double)maxbuf[0]);
  const double sL = (double)(step / ((float)m + tiny));
  for (int e = t; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e % r;
    double s = 0.0;
    for (int k = 0; k < r; ++k) s += G[i][k] * (c.L1[k][j] * irho);
    L12o[(long)i * r + j] = (float)(c.L1[i][j] * irho - sL * s);
  }
  if (t < r) {   // a = L1s' Qg1, b = L1s' iQtx1
    double a = 0.0, b = 0.0;
    for (int k = 0; k < r; ++k) {
      a += c.L1[k][t] * irho * c.v[1][k];
      b += c.L1[k][t] * irho * c.v[2][k];
    }
    coef[2 * r + t] = (float)a;
    coef[3 * r + t] = (float)b;
  }
  __syncthreads();

  // ---- U: grad1 = triu(Pg1 dg1' - dx1 iPx1'), step0, newU1 = U1s - U1s (step0 grad1), U1s = rho U1
  m = 0.0;
  for (int e = t; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e % r;
    const double g = (j >= i) ? c.v[3][i] * c.v[5][j] - c.v[4][i] * c.v[0][j] : 0.0;
    G[i][j] = g;
    m = nmax(m, fabs(g));
  }
  m = block_max_f64(m, red);
  if (has_tail) m = nmax(m, (double)maxbuf[1]);
  const double sU = (double)(step / ((float)m + tiny));
  for (int e = t; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e % r;
    double s = 0.0;
    for (int k = 0; k < r; ++k) s += (c.U1[i][k] * rho) * G[k][j];
    U12o[(long)i * ldu + j] = (float)(c.U1[i][j] * rho - sU * s);
  }
  if (t < r) {   // c = U1s Pg1, e = U1s dx1
    double cc = 0.0, ee = 0.0;
    for (int k = 0; k < r; ++k) {
      cc += c.U1[t][k] * rho * c.v[3][k];
      ee += c.U1[t][k] * rho * c.v[4][k];
    }
    coef[4 * r + t] = (float)cc;
    coef[5 * r + t] = (float)ee;
    coef[t] = (float)dbl[kStLtQg1 + t];
    coef[r + t] = (float)dbl[kStILiQtx1 + t];
  }
  if (t == 0) {
    coef[8 * r + 0] = (float)sL;
    coef[8 * r + 1] = (float)sU;
    coef[8 * r + 2] = rho_f;
    coef[8 * r + 3] = (float)irho;
  }
}

// ------------------------------------------------------------- host side ---
static inline int col_tile_rows(int nvec) { return 64 * ((nvec <= 12) ? 4 : ((nvec <= 24) ? 2 : 1)); }

static int splu_grid(const SpluOps* ops, int r, int which, int64_t rows, bool tiled) {
  static int occ_cache[PSGD_SPLU_MAX_RANK + 1][6];
  int occ = occ_cache[r][which];
  if (occ == 0) {
    occ = ops->occupancy(which);
    if (occ <= 0) occ = 1;
    occ_cache[r][which] = occ;
  }
  occ = policy_grid_blocks(occ);
  int64_t grid;
  if (tiled) {
    const int64_t tiles = (rows + ops->tile_rows - 1) / ops->tile_rows;
    grid = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  } else {   // column sweeps (ColCfg in splu_kernels.h): which 0 = u2dot (r + 1 vectors), 2 = apply_s3 (r + 3)
    const int tr = col_tile_rows(which == 0 ? r + 1 : r + 3);
    const int64_t tiles = (rows + tr - 1) / tr;
    grid = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  }
  const int64_t cap = (int64_t)device_cus() * occ;
  if (grid > cap) grid = cap;
  if (grid > kMaxGrid) grid = kMaxGrid;
  if (grid < 1) grid = 1;
  return (int)grid;
}

struct SpluGeom {
  int64_t n2, n2s;
  int head;
  long ldu;
};

static SpluGeom splu_geom(int64_t N, int r) {
  SpluGeom g;
  g.n2 = N - r;
  // The streamed part starts `head` rows into the tail so that (r + head) is a multiple of 32 rows: the L2 tiles then
  // start on a 16-byte boundary for every r ((r + head) r floats), and the r column streams of U2 (row k of U12 starts
  // k N + r floats into the buffer) start on a 128-byte line whenever N is a multiple of 32 -- unaligned 256-byte
  // wave accesses straddle an extra line that the neighbouring tile fetches again (+13 % fetched bytes measured).
  int h = (32 - (r & 31)) & 31;
  if (h > g.n2) h = (int)g.n2;
  g.head = h;
  g.n2s = g.n2 - h;
  g.ldu = (long)N;
  return g;
}

}  // namespace psgd

using namespace psgd;

#define PSGD_CHECK_LAUNCH(expr)                 \
  do {                                          \
    const int _e = (expr);                      \
    if (_e != 0) return PSGD_ERR_LAUNCH;        \
  } while (0)
static inline int last_launch() { return (int)hipGetLastError(); }
static inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

static int splu_open(void* ws, int64_t ws_bytes, int64_t N, int r, SpluWs* w) {
  if (r < 1 || r > PSGD_SPLU_MAX_RANK) return PSGD_ERR_RANK;
  if (N < r) return PSGD_ERR_BAD_ARG;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return PSGD_ERR_WORKSPACE;
  if (ws_bytes < splu_layout(N, r, nullptr, nullptr)) return PSGD_ERR_WORKSPACE;
  splu_layout(N, r, static_cast<char*>(ws), w);
  return PSGD_OK;
}

extern "C" {

int64_t psgd_splu_workspace_bytes(int64_t N, int r) {
  if (r < 1 || r > PSGD_SPLU_MAX_RANK || N < r) return 0;
  return splu_layout(N, r, nullptr, nullptr);
}

/* ---- stage entry points.  One call = stage 1..3 (apply) or 1..4 (update) back to back.  With the tail rows sharded
 * over several GPUs (every rank holds [L1; its rows of L2], [U1, its columns of U2], its slices of l3, u3 and of the
 * tail of the flat vectors; the r x r corner and the first r vector entries are replicated) the caller all-reduces
 * the region psgd_splu_ws_region() reports between two stages; every rank then redoes the corner algebra on
 * identical inputs. ---- */

struct SpluCtx {
  SpluWs w; SpluGeom ge; const SpluOps* ops; hipStream_t st; int nt, rblocks, r2blocks;
};

static int splu_ctx(int64_t N, int r, void* ws, int64_t ws_bytes, void* stream, const void* L12, SpluCtx* c) {
  const int rc = splu_open(ws, ws_bytes, N, r, &c->w);
  if (rc) return rc;
  c->ge = splu_geom(N, r);
  if (L12 && misaligned16(L12)) return PSGD_ERR_ALIGN;
  c->ops = splu_ops_for_rank(r);
  if (!c->ops) return PSGD_ERR_RANK;
  c->st = static_cast<hipStream_t>(stream);
  c->nt = policy_nt((int64_t)N * r * 8);
  c->rblocks = (r + kWavesPerBlock - 1) / kWavesPerBlock;
  c->r2blocks = (2 * r + kWavesPerBlock - 1) / kWavesPerBlock;
  return PSGD_OK;
}

int psgd_splu_ws_region(int which, int stage, int64_t N, int r, int64_t* offset_bytes, int64_t* count) {
  if (!offset_bytes || !count || r < 1 || r > PSGD_SPLU_MAX_RANK || N < r) return PSGD_ERR_BAD_ARG;
  SpluWs w;
  char* const base = reinterpret_cast<char*>(static_cast<uintptr_t>(1) << 20);   // any aligned address: offsets only
  splu_layout(N, r, base, &w);
  const int64_t dbl = reinterpret_cast<char*>(w.dbl) - base;
  const int64_t mx = reinterpret_cast<char*>(w.maxbuf) - base;
  if (which == 0) {                    // fp64 sums (all-reduce SUM)
    if (stage == 1) { *offset_bytes = dbl + kSumA * 8; *count = r; return PSGD_OK; }       // U2 x2        (both paths)
    if (stage == 2) { *offset_bytes = dbl + kSumB * 8; *count = 2 * r; return PSGD_OK; }   // L2'Qg2 [, L2'iQtx2]
    if (stage == 3) { *offset_bytes = dbl + kSumC * 8; *count = r; return PSGD_OK; }       // U2 iPx2      (update)
  } else if (which == 1 && stage == 3) {   // fp32 maxima (all-reduce MAX): |grad L|, |grad U|, l3, u3
    *offset_bytes = mx; *count = 4; return PSGD_OK;
  } else if (which == PSGD_WS_SEND_F64) {  // fp64 send regions (all-gather, then psgd_splu_fold_gathered_f64)
    if (stage == 1 || stage == 2) return psgd_splu_ws_region(0, stage, N, r, offset_bytes, count);
    if (stage == 3) { *offset_bytes = dbl + kSumC * 8; *count = r + 4; return PSGD_OK; }   // [U2 iPx2 | 4 maxima]
  }
  return PSGD_ERR_BAD_ARG;
}

int psgd_splu_fold_gathered_f64(int stage, const double* gathered, int world, int64_t N, int r, void* ws,
                                int64_t ws_bytes, void* stream) {
  if (!gathered || world < 1) return PSGD_ERR_BAD_ARG;
  SpluWs w;
  const int rc = splu_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  int64_t off = 0, count = 0;
  const int rr = psgd_splu_ws_region(PSGD_WS_SEND_F64, stage, N, r, &off, &count);
  if (rr) return rr;
  double* dst = reinterpret_cast<double*>(static_cast<char*>(ws) + off);
  const int nsum = (stage == 3) ? r : (int)count;
  hipLaunchKernelGGL(k_splu_fold_gathered, dim3(((int)count + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     gathered, world, (int)count, nsum, dst, w.maxbuf);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

/* stage 1 of either path: sums A = U2 x2 (x = the flat gradient for the apply, dg for the update) */
int psgd_splu_stage1_f32(const float* U12, const float* x, int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!U12 || !x) return PSGD_ERR_BAD_ARG;
  SpluCtx c;
  const int rc = splu_ctx(N, r, ws, ws_bytes, stream, nullptr, &c);
  if (rc) return rc;
  const int h = c.ge.head;
  const int grid = splu_grid(c.ops, r, 0, c.ge.n2s, false);
  PSGD_CHECK_LAUNCH(c.ops->u2dot(c.nt, U12 + r + h, c.ge.ldu, x + r + h, c.ge.n2s, h, c.w.part, grid, c.st));
  hipLaunchKernelGGL(k_splu_reduce_sum, dim3(c.rblocks), dim3(kThreads), 0, c.st, c.w.part, grid, r, c.w.dbl + kSumA);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_splu_apply_stage2_f32(const float* L12, const float* l3, const float* U12, const float* u3, const float* g,
                               float* out, int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!L12 || !U12 || !g || !out) return PSGD_ERR_BAD_ARG;
  SpluCtx c;
  const int rc = splu_ctx(N, r, ws, ws_bytes, stream, L12, &c);
  if (rc) return rc;
  if (c.ge.n2 > 0 && (!l3 || !u3)) return PSGD_ERR_BAD_ARG;
  const int h = c.ge.head;
  hipLaunchKernelGGL(k_splu_corner1, dim3(1), dim3(kThreads), 0, c.st, L12, U12, c.ge.ldu, r, g, (const float*)nullptr,
                     c.w.dbl, c.w.coef);
  PSGD_CHECK_LAUNCH(last_launch());
  const int grid = splu_grid(c.ops, r, 1, c.ge.n2s, true);
  PSGD_CHECK_LAUNCH(c.ops->apply_s2(c.nt, L12 + (int64_t)(r + h) * r, l3 + h, u3 + h, g + r + h, out + r + h, c.ge.n2s, h,
                                    c.w.coef, c.w.part, grid, c.st));
  hipLaunchKernelGGL(k_splu_reduce_sum, dim3(c.rblocks), dim3(kThreads), 0, c.st, c.w.part, grid, r, c.w.dbl + kSumB);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_splu_apply_stage3_f32(const float* L12, const float* l3, const float* U12, const float* u3, float* out, int64_t N,
                               int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!L12 || !U12 || !out) return PSGD_ERR_BAD_ARG;
  SpluCtx c;
  const int rc = splu_ctx(N, r, ws, ws_bytes, stream, L12, &c);
  if (rc) return rc;
  const int h = c.ge.head;
  hipLaunchKernelGGL(k_splu_corner_apply2, dim3(1), dim3(kThreads), 0, c.st, L12, U12, c.ge.ldu, r, c.w.dbl, c.w.coef, out);
  PSGD_CHECK_LAUNCH(last_launch());
  if (c.ge.n2 > 0) {
    const int grid = splu_grid(c.ops, r, 2, c.ge.n2s, false);
    PSGD_CHECK_LAUNCH(c.ops->apply_s3(c.nt, U12 + r + h, c.ge.ldu, l3 + h, u3 + h, out + r + h, c.ge.n2s, h, c.w.coef, grid,
                                      c.st));
  }
  return PSGD_OK;
}

int psgd_splu_apply_f32(const float* L12, const float* l3, const float* U12, const float* u3, const float* g, float* out,
                        int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!L12 || !U12 || !g || !out) return PSGD_ERR_BAD_ARG;
  int rc = psgd_splu_stage1_f32(U12, g, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  rc = psgd_splu_apply_stage2_f32(L12, l3, U12, u3, g, out, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  return psgd_splu_apply_stage3_f32(L12, l3, U12, u3, out, N, r, ws, ws_bytes, stream);
}

int psgd_splu_update_stage2_f32(const float* L12, const float* l3, const float* U12, const float* u3, const float* dx,
                                const float* dg, int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!L12 || !U12 || !dx || !dg) return PSGD_ERR_BAD_ARG;
  SpluCtx c;
  const int rc = splu_ctx(N, r, ws, ws_bytes, stream, L12, &c);
  if (rc) return rc;
  if (c.ge.n2 > 0 && (!l3 || !u3)) return PSGD_ERR_BAD_ARG;
  const int h = c.ge.head;
  hipLaunchKernelGGL(k_splu_corner1, dim3(1), dim3(kThreads), 0, c.st, L12, U12, c.ge.ldu, r, dg, dx, c.w.dbl, c.w.coef);
  PSGD_CHECK_LAUNCH(last_launch());
  const int grid = splu_grid(c.ops, r, 3, c.ge.n2s, true);
  PSGD_CHECK_LAUNCH(c.ops->upd_s2(c.nt, L12 + (int64_t)(r + h) * r, U12 + r + h, c.ge.ldu, l3 + h, u3 + h, dx + r + h,
                                  dg + r + h, c.ge.n2s, h, c.w.coef, c.w.part, grid, c.st));
  // sums B are stored [L2'Qg2 (r) | L2'iQtx2 (r)] contiguously
  hipLaunchKernelGGL(k_splu_reduce_sum, dim3(c.r2blocks), dim3(kThreads), 0, c.st, c.w.part, grid, 2 * r, c.w.dbl + kSumB);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_splu_update_stage3_f32(const float* L12, const float* l3, const float* U12, const float* u3, const float* dx,
                                const float* dg, int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!L12 || !U12 || !dx || !dg) return PSGD_ERR_BAD_ARG;
  SpluCtx c;
  const int rc = splu_ctx(N, r, ws, ws_bytes, stream, L12, &c);
  if (rc) return rc;
  const int h = c.ge.head;
  hipLaunchKernelGGL(k_splu_corner_upd2, dim3(1), dim3(kThreads), 0, c.st, L12, U12, c.ge.ldu, r, dx, c.w.dbl, c.w.coef);
  PSGD_CHECK_LAUNCH(last_launch());
  const int grid = splu_grid(c.ops, r, 4, c.ge.n2s, true);
  PSGD_CHECK_LAUNCH(c.ops->upd_s3(c.nt, L12 + (int64_t)(r + h) * r, U12 + r + h, c.ge.ldu, l3 + h, u3 + h, dg + r + h,
                                  dx + r + h, c.ge.n2s, h, c.w.coef, c.w.part, c.w.pmax, grid, c.st));
  hipLaunchKernelGGL(k_splu_reduce_sum, dim3(c.rblocks), dim3(kThreads), 0, c.st, c.w.part, grid, r, c.w.dbl + kSumC);
  hipLaunchKernelGGL(k_splu_reduce_max, dim3(4), dim3(kThreads), 0, c.st, c.w.pmax, grid, c.w.maxbuf, c.w.dbl + kSumC + r);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

/* has_tail: the GLOBAL problem has tail rows (N_global > r); a rank of a sharded call may hold none of them itself */
int psgd_splu_update_stage4_f32(const float* L12, const float* l3, const float* U12, const float* u3, const float* dx,
                                const float* dg, float* L12_new, float* l3_new, float* U12_new, float* u3_new, int64_t N,
                                int r, float step, float tiny, int has_tail, void* ws, int64_t ws_bytes, void* stream) {
  if (!L12 || !U12 || !dx || !dg || !L12_new || !U12_new) return PSGD_ERR_BAD_ARG;
  SpluCtx c;
  const int rc = splu_ctx(N, r, ws, ws_bytes, stream, L12, &c);
  if (rc) return rc;
  if (misaligned16(L12_new)) return PSGD_ERR_ALIGN;
  if (c.ge.n2 > 0 && (!l3 || !u3 || !l3_new || !u3_new)) return PSGD_ERR_BAD_ARG;
  const int h = c.ge.head;
  hipLaunchKernelGGL(k_splu_corner_upd3, dim3(1), dim3(kThreads), 0, c.st, L12, U12, c.ge.ldu, r, dx, dg, has_tail ? 1 : 0,
                     step, tiny, c.w.dbl, c.w.maxbuf, c.w.coef, L12_new, U12_new);
  PSGD_CHECK_LAUNCH(last_launch());
  if (c.ge.n2 > 0) {
    const int grid = splu_grid(c.ops, r, 5, c.ge.n2s, true);
    PSGD_CHECK_LAUNCH(c.ops->upd_s4(c.nt, L12 + (int64_t)(r + h) * r, U12 + r + h, c.ge.ldu, l3 + h, u3 + h, dg + r + h,
                                    dx + r + h, L12_new + (int64_t)(r + h) * r, U12_new + r + h, l3_new + h, u3_new + h,
                                    c.ge.n2s, h, c.w.coef, grid, c.st));
  }
  return PSGD_OK;
}

int psgd_splu_update_f32(const float* L12, const float* l3, const float* U12, const float* u3, const float* dx,
                         const float* dg, float* L12_new, float* l3_new, float* U12_new, float* u3_new, int64_t N, int r,
                         float step, float tiny, void* ws, int64_t ws_bytes, void* stream) {
  if (!L12 || !U12 || !dx || !dg || !L12_new || !U12_new) return PSGD_ERR_BAD_ARG;
  int rc = psgd_splu_stage1_f32(U12, dg, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  rc = psgd_splu_update_stage2_f32(L12, l3, U12, u3, dx, dg, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  rc = psgd_splu_update_stage3_f32(L12, l3, U12, u3, dx, dg, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  return psgd_splu_update_stage4_f32(L12, l3, U12, u3, dx, dg, L12_new, l3_new, U12_new, u3_new, N, r, step, tiny,
                                     N > r ? 1 : 0, ws, ws_bytes, stream);
}

}  // extern "C"
