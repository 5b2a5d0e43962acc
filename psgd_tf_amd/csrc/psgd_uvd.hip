// psgd_uvd.hip -- C ABI (include/psgd_hip.h) of the UVd preconditioner path
// plus the rank-independent kernels: partial-row reductions, the r x r algebra
// of update_precond_UVd_math_ (psgd.py:574-615) in fp64, the d update
// (psgd.py:582-584) and the U/V balancing branch (psgd.py:562-567).
#include "uvd_kernels.h"
#include "psgd_hip.h"
#include <math.h>

namespace psgd {

const UvdOps* uvd_ops_group0(int r);
const UvdOps* uvd_ops_group1(int r);
const UvdOps* uvd_ops_group2(int r);
const UvdOps* uvd_ops_group3(int r);

const UvdWideOps* uvd_wide_group0(int r);
const UvdWideOps* uvd_wide_group1(int r);
const UvdWideOps* uvd_wide_group2(int r);
const UvdWideOps* uvd_wide_group3(int r);
const UvdWideOps* uvd_wide_ops_for_rank(int r) {
  if (r <= PSGD_UVD_MAX_RANK || r > 2 * PSGD_UVD_MAX_RANK) return nullptr;
  switch ((r - 33) / 8) {
    case 0: return uvd_wide_group0(r);
    case 1: return uvd_wide_group1(r);
    case 2: return uvd_wide_group2(r);
    default: return uvd_wide_group3(r);
  }
}

const UvdOps* uvd_ops_for_rank(int r) {
  if (r < 1 || r > PSGD_UVD_MAX_RANK) return nullptr;
  switch ((r - 1) / 8) {
    case 0: return uvd_ops_group0(r);
    case 1: return uvd_ops_group1(r);
    case 2: return uvd_ops_group2(r);
    default: return uvd_ops_group3(r);
  }
}

// ------------------------------------------------------------ workspace ----
constexpr int64_t kSumsCap = 4352;   // doubles (Gram of r = 32 needs 3840; the small regions below follow it)
constexpr int kPqSumsOff = 3840;      // the fused sums [pU | pV | qU | qV] (4r <= 128 doubles) live above the largest Gram; the
                                      // fp64 copy of max|nablaD| follows them at [kPqSumsOff + 4r] (one contiguous send region)
constexpr int kBalD64Off = 4000;      // fp64 copies of the two balance maxima (send region of stage 10)
constexpr int kCoef64Off = 4016;      // fp64 by-products of the r x r algebra that k_fused_post needs: Ua | Ub | c1 | c2 (4r) + mu, a'a, a'b, b'b
constexpr int kPostSumsOff = 4160;    // fp64 s1', s2' of the fused step (2r <= 64 doubles; diagnostics)
constexpr int64_t kCoefCap = 256;    // floats
constexpr int64_t kMaxCap = 64;      // floats

struct WsLayout {
  int64_t sums_off, coef_off, max_off, part_off, part_bytes, pmax_off, nabla_off, total;
};

static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

static WsLayout ws_layout(int64_t N, int r) {
  WsLayout L;
  const int nc = 2 * r + 2, nb = (nc + 15) / 16, np = nb * (nb + 1) / 2;
  int64_t off = 0;
  L.sums_off = off; off = align256(off + kSumsCap * 8);
  L.coef_off = off; off = align256(off + kCoefCap * 4);
  L.max_off = off;  off = align256(off + kMaxCap * 4);
  const int64_t part_f32 = (int64_t)kMaxGrid * 2 * PSGD_UVD_MAX_RANK * 4;
  const int64_t part_f64 = (int64_t)kGramMaxGrid * np * 256 * 8;
  L.part_bytes = part_f32 > part_f64 ? part_f32 : part_f64;
  L.part_off = off; off = align256(off + L.part_bytes);
  L.pmax_off = off; off = align256(off + 2 * (int64_t)kMaxGrid * 4);
  L.nabla_off = off; off = align256(off + N * 4);
  L.total = off;
  return L;
}

struct Ws {
  double* sums; float* coef; float* maxbuf; void* part; float* pmax; float* nabla;
};

static int ws_open(void* ws, int64_t ws_bytes, int64_t N, int r, Ws* out) {
  if (N <= 0) return PSGD_ERR_BAD_ARG;
  if (r < 1 || r > PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) return PSGD_ERR_WORKSPACE;
  const WsLayout L = ws_layout(N, r);
  if (ws_bytes < L.total) return PSGD_ERR_WORKSPACE;
  char* b = static_cast<char*>(ws);
  out->sums = reinterpret_cast<double*>(b + L.sums_off);
  out->coef = reinterpret_cast<float*>(b + L.coef_off);
  out->maxbuf = reinterpret_cast<float*>(b + L.max_off);
  out->part = b + L.part_off;
  out->pmax = reinterpret_cast<float*>(b + L.pmax_off);
  out->nabla = reinterpret_cast<float*>(b + L.nabla_off);
  return PSGD_OK;
}

static inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

// ------------------------------------------------------------ device info --
static int g_tune_staging = 0;
static int g_tune_blocks_per_cu = 0;
static int g_tune_tiles_per_wave = 8;

// Optional live kernel timing (bench.py): when enabled, every sweep launch is bracketed by a pair
// of HIP events recorded on the launch stream; psgd_prof_collect() resolves them afterwards.
// Nothing here synchronises inside a hot-path call.
constexpr int kProfSlots = 8;
constexpr int kProfMaxPairs = 4096;
struct ProfPair { hipEvent_t e0, e1; };
static int g_prof_on = 0;
static ProfPair g_prof_pairs[kProfSlots][kProfMaxPairs];
static int g_prof_created[kProfSlots];
static int g_prof_used[kProfSlots];

struct ProfScope {
  hipEvent_t e1 = nullptr;
  hipStream_t st;
  ProfScope(int slot, hipStream_t s) : st(s) {
    if (!g_prof_on || slot < 0 || slot >= kProfSlots || g_prof_used[slot] >= kProfMaxPairs) return;
    const int i = g_prof_used[slot];
    if (i >= g_prof_created[slot]) {
      if (hipEventCreate(&g_prof_pairs[slot][i].e0) != hipSuccess) return;
      if (hipEventCreate(&g_prof_pairs[slot][i].e1) != hipSuccess) return;
      g_prof_created[slot] = i + 1;
    }
    g_prof_used[slot] = i + 1;
    (void)hipEventRecord(g_prof_pairs[slot][i].e0, st);
    e1 = g_prof_pairs[slot][i].e1;
  }
  ~ProfScope() {
    if (e1) (void)hipEventRecord(e1, st);
  }
};

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}

// non-temporal streams only when U and V together exceed the Infinity Cache (256 MiB)
static inline int use_nt(int64_t N, int r) {
  if (g_tune_staging == 1) return 0;
  if (g_tune_staging == 2) return 1;
  return (int64_t)N * r * 8 > (int64_t)192 * 1024 * 1024;
}

int policy_nt(int64_t stream_bytes) {
  if (g_tune_staging == 1) return 0;
  if (g_tune_staging == 2) return 1;
  return stream_bytes > (int64_t)192 * 1024 * 1024;
}
int policy_grid_blocks(int occ) {
  if (occ <= 0) occ = 1;
  if (g_tune_blocks_per_cu > 0 && g_tune_blocks_per_cu < occ) occ = g_tune_blocks_per_cu;
  if (g_tune_blocks_per_cu < 0) occ = -g_tune_blocks_per_cu;
  return occ;
}
int device_cus() { return num_cus(); }

static int g_tune_grid[16] = {0};     // psgd_set_tuning key 10 + kind: workgroups of that sweep kind (0 = the rule below)

static int sweep_grid(const UvdOps* ops, int r, int which, int64_t N, int hard_cap) {
  static int occ_cache[PSGD_UVD_MAX_RANK + 1][16];
  if (which >= 0 && which < 16 && g_tune_grid[which] > 0) {
    const int64_t tiles = (N + ops->tile_rows - 1) / ops->tile_rows;
    int64_t grid = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    if (grid > g_tune_grid[which]) grid = g_tune_grid[which];
    if (grid > hard_cap) grid = hard_cap;
    return (int)(grid < 1 ? 1 : grid);
  }
  int occ = occ_cache[r][which];
  if (occ == 0) {
    occ = ops->occupancy(which);
    if (occ <= 0) occ = 1;
    occ_cache[r][which] = occ;
  }
  // Round 6 (profiles/r06_grid_scan.txt): the sweeps that WRITE a factor or several vectors run faster on FEWER waves -- update sweep 2
  // on one workgroup per CU instead of the three that fit: 4.87 -> 4.69 ms at N = 100M, r = 20 (placed state), -6 .. -17 % at
  // r = 10 .. 32 and N = 4M .. 50M; the last sweep of the fused step -1 .. -5 %.  (Fewer concurrent 5-KiB streams keep more DRAM rows
  // open between a tile's read and its write-back.)  Tiles below 4 KiB per operand (r <= 8) need two workgroups per CU to cover the
  // latency; the read-only sweeps (Gram, apply) keep every workgroup that fits.  Multiples of the CU count only: 320 or 384
  // workgroups leave a second, mostly idle round.
  if (which == kOccUpdS2U || which == kOccUpdS2V || which == kOccUpdS2F) {
    const int want = ((long)ops->tile_rows * r * 4 >= 4096) ? 1 : 2;
    if (want < occ) occ = want;
  } else if (which == kOccFinal) {
    occ = 1;
  }
  if (g_tune_blocks_per_cu > 0 && g_tune_blocks_per_cu < occ) occ = g_tune_blocks_per_cu;
  if (g_tune_blocks_per_cu < 0) occ = -g_tune_blocks_per_cu;   // experiments: force, even above the occupancy query
  const int64_t tiles = (N + ops->tile_rows - 1) / ops->tile_rows;
  int64_t grid = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  int64_t cap = (int64_t)num_cus() * occ;
  // short sweeps (cache-resident problems): a wave should stream at least ~8 tiles, or its fixed costs -- first-load
  // latency, block reduction, one row of partials per block for the reduce kernels to read -- dominate; never fewer
  // blocks than CUs (profiles/r02_c2_nscale.txt: one block per CU is as fast or faster up to N = 8M at r = 10)
  if (g_tune_blocks_per_cu == 0) {
    int64_t want = tiles / (kWavesPerBlock * (g_tune_tiles_per_wave > 0 ? g_tune_tiles_per_wave : 1));
    if (want < num_cus()) want = num_cus();
    if (cap > want) cap = want;
  }
  if (grid > cap) grid = cap;
  if (grid > hard_cap) grid = hard_cap;
  if (grid < 1) grid = 1;
  return (int)grid;
}

// ---------------------------------------------------------- small kernels --
// sums[id] = sum_b part[b][id] in fp64, fixed order, for the big fp64 Gram partials stored
// [G][L].  One lane per element id; the four waves of a block take interleaved slices of the
// G partial rows, eight independent loads in flight per lane.
template <class T>
__global__ __launch_bounds__(kThreads) void k_reduce_sum(const T* __restrict__ part, int G, int L,
                                                         double* __restrict__ sums, float* __restrict__ coef) {
  __shared__ double red[kWavesPerBlock][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int id = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (id < L) {
    int b = w;
    for (; b + 7 * kWavesPerBlock < G; b += 8 * kWavesPerBlock) {
      T x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = part[(long)(b + u * kWavesPerBlock) * L + id];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)x[u];
    }
    for (; b < G; b += kWavesPerBlock) s += (double)part[(long)b * L + id];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && id < L) {
    const double t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    sums[id] = t;
    if (coef) coef[id] = (float)t;
  }
}

// sums[id] = sum_b part[id][b] for the transposed fp32 partials of the column-reduction sweeps
// ([L][G], G <= kMaxGrid = 2048): one wave per element; every lane issues its (up to 32) loads of
// the element's G partials before summing, so one memory latency is paid, not G/64 of them.
// fp64 shuffle tree, fixed order.
__global__ __launch_bounds__(kThreads) void k_reduce_sum_t(const float* __restrict__ part, int G, int L,
                                                           double* __restrict__ sums, float* __restrict__ coef) {
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
  if (lane == 0) {
    sums[id] = s;
    if (coef) coef[id] = (float)s;
  }
}

// out[set] = max_b part[set*stride + b]; outd[set] = the same value as a double (multi-GPU send regions are fp64)
__global__ __launch_bounds__(kThreads) void k_reduce_max(const float* __restrict__ part, int G, int stride,
                                                         float* __restrict__ out, double* __restrict__ outd) {
  __shared__ float red[kWavesPerBlock];
  const float* p = part + (long)blockIdx.x * stride;
  float v = 0.0f;
  for (int b = threadIdx.x; b < G; b += kThreads) v = amaxf(v, p[b]);
  block_max_store(v, red, out + blockIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) outd[blockIdx.x] = (double)out[blockIdx.x];
}

// Multi-GPU exchange, second half (the first half is an all-gather of every rank's send region): `gathered` holds the
// regions of all ranks, [world][count] doubles in rank order.  Every rank folds them in that order -- entries [0, nsum)
// by +, the rest by max -- so all ranks end up with bit-identical reduced values whatever algorithm the collective
// library uses for the gather.  Maxima also go back to the fp32 max buffer the next stage reads.
__global__ void k_fold_gathered(const double* __restrict__ gathered, int world, int count, int nsum,
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

__global__ void k_publish(const double* __restrict__ sums, float* __restrict__ coef, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) coef[i] = (float)sums[i];
}

// Reduction behind the fused sweep 2: sums[e] = sum_b part[e * G + b] for the 4r column sums of ColSum2 (transposed fp64
// partials: one wave per element, every lane issues its <= 32 loads before summing; fp64 shuffle tree, fixed order),
// and max|nablaD| over the block maxima (wave 0 of block 0), as float and as the double that ends the send region.
__global__ __launch_bounds__(kThreads) void k_reduce_pq(const double* __restrict__ part, const float* __restrict__ pmax,
                                                        int G, int L, double* __restrict__ sums,
                                                        float* __restrict__ maxout) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (id == 0 && pmax) {
    float m = 0.0f;
    for (int b = lane; b < G; b += 64) m = amaxf(m, pmax[b]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = amaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) {
      *maxout = m;
      sums[L] = (double)m;
    }
  }
  if (id >= L) return;
  const double* p = part + (long)id * G;
  double x[kMaxGrid / 64];
#pragma unroll
  for (int u = 0; u < kMaxGrid / 64; ++u) {
    const int b = lane + 64 * u;
    x[u] = (b < G) ? p[b] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < kMaxGrid / 64; ++u) s += x[u];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) sums[id] = s;
}

// d <- d - (mu d) nablaD, mu = step / (max|nablaD| + tiny)      psgd.py:582-584
// pmax != nullptr (single-GPU call): max|nablaD| is still G block maxima; every block folds them itself (a maximum
// does not depend on the order) instead of a reduction kernel of its own in front of this one.
__global__ __launch_bounds__(kThreads) void k_update_d(float* d, const float* __restrict__ nabla, long N,
                                                       float* __restrict__ maxbuf, const float* __restrict__ pmax, int G,
                                                       float step, float tiny) {
  __shared__ float red[kWavesPerBlock];
  __shared__ float bmax;
  float m;
  if (pmax) {
    float v = 0.0f;
    for (int b = threadIdx.x; b < G; b += kThreads) v = amaxf(v, pmax[b]);
    block_max_store(v, red, &bmax);
    __syncthreads();
    m = bmax;
    if (blockIdx.x == 0 && threadIdx.x == 0) maxbuf[0] = m;
  } else {
    m = maxbuf[0];
  }
  const float mu = step / (m + tiny);
  const long n4 = N / 4;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nth = (long)gridDim.x * blockDim.x;
  float4* d4 = reinterpret_cast<float4*>(d);
  const float4* n4p = reinterpret_cast<const float4*>(nabla);
  for (long i = tid; i < n4; i += nth) {
    float4 a = d4[i];
    const float4 b = n4p[i];
    a.x = a.x - (mu * a.x) * b.x;
    a.y = a.y - (mu * a.y) * b.y;
    a.z = a.z - (mu * a.z) * b.z;
    a.w = a.w - (mu * a.w) * b.w;
    d4[i] = a;
  }
  for (long i = n4 * 4 + tid; i < N; i += nth) d[i] = d[i] - (mu * d[i]) * nabla[i];
}

// max|U|, max|V| over the flat [N*r] arrays                       psgd.py:563-564
__global__ __launch_bounds__(kThreads) void k_maxabs2(const float* __restrict__ U, const float* __restrict__ V,
                                                      long n, float* part, int G) {
  __shared__ float red[2][kWavesPerBlock];
  const long n4 = n / 4;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nth = (long)gridDim.x * blockDim.x;
  float mu = 0.0f, mv = 0.0f;
  const float4* U4 = reinterpret_cast<const float4*>(U);
  const float4* V4 = reinterpret_cast<const float4*>(V);
  for (long i = tid; i < n4; i += nth) {
    const float4 a = U4[i], b = V4[i];
    mu = amaxf(mu, amaxf(amaxf(fabsf(a.x), fabsf(a.y)), amaxf(fabsf(a.z), fabsf(a.w))));
    mv = amaxf(mv, amaxf(amaxf(fabsf(b.x), fabsf(b.y)), amaxf(fabsf(b.z), fabsf(b.w))));
  }
  for (long i = n4 * 4 + tid; i < n; i += nth) {
    mu = amaxf(mu, fabsf(U[i]));
    mv = amaxf(mv, fabsf(V[i]));
  }
  block_max_store(mu, red[0], part + blockIdx.x);
  __syncthreads();
  block_max_store(mv, red[1], part + G + blockIdx.x);
}

// U <- U / rho, V <- rho V, rho = sqrt(max|U| / max|V|)            psgd.py:565-567
__global__ __launch_bounds__(kThreads) void k_scale2(float* U, float* V, long n, const float* __restrict__ maxbuf) {
  const float rho = sqrtf(maxbuf[0] / maxbuf[1]);
  const long n4 = n / 4;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nth = (long)gridDim.x * blockDim.x;
  float4* U4 = reinterpret_cast<float4*>(U);
  float4* V4 = reinterpret_cast<float4*>(V);
  for (long i = tid; i < n4; i += nth) {
    float4 a = U4[i], b = V4[i];
    a.x /= rho; a.y /= rho; a.z /= rho; a.w /= rho;
    b.x *= rho; b.y *= rho; b.z *= rho; b.w *= rho;
    U4[i] = a;
    V4[i] = b;
  }
  for (long i = n4 * 4 + tid; i < n; i += nth) {
    U[i] = U[i] / rho;
    V[i] = rho * V[i];
  }
}

// ------------------------------------------------- r x r algebra (fp64) ----
constexpr int MR = PSGD_UVD_MAX_RANK;

// Solve M x = rhs (r <= 32) with Gaussian elimination and partial pivoting (first maximal |entry|, as
// LAPACK's getrf behind tf.linalg.solve), cooperatively by a 256-thread block: the augmented matrix
// Mx[i][0..r] (column r = rhs) lives in LDS; every elimination step updates its (r-k-1) x (r-k)
// trailing entries in parallel, one barrier pair per step.  x is left in xs[0..r).
// 1/d to full fp64 accuracy: hardware v_rcp_f64 estimate + two Newton steps (an IEEE fp64 division
// costs ~500 cycles on the vector ALU and sat on the critical path of every elimination step)
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = x * (2.0 - d * x);
  x = x * (2.0 - d * x);
  return x;
}

// max over lanes 0..31 of a wave with DPP row shifts (VALU speed; a __shfl_down tree on doubles costs
// ~900 cycles of ds_bpermute latency per search).  Result valid in every lane (readlane 31).
__device__ __forceinline__ unsigned wave32_umax(unsigned x) {
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));   // row_shr:1
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));   // row_shr:2
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));   // row_shr:4
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));   // row_shr:8
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1,3
  return (unsigned)__builtin_amdgcn_readlane((int)x, 31);
}

// 64-lane maximum of unsigned keys on DPP row shifts / broadcasts; result valid in every lane.
__device__ __forceinline__ unsigned wave64_umax(unsigned x) {
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));   // row_shr:1
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));   // row_shr:2
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));   // row_shr:4
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));   // row_shr:8
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1,3
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2,3
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

template <int CAP>
__device__ void lu_solve_block(double (*Mx)[CAP + 2], int r, double* xs) {
  const int tid = threadIdx.x;
  for (int k = 0; k < r; ++k) {
    // pivot = first row i >= k with the largest |Mx[i][k]| (compared as fp32 keys): every wave finds it
    // redundantly (lane l looks at row k + l), so no broadcast and no extra barrier is needed
    const int lane = tid & 63;
    const unsigned key = (lane < r - k) ? __float_as_uint(fabsf((float)Mx[k + lane][k])) : 0u;
    const unsigned kmax = (CAP <= 32) ? wave32_umax(key) : wave64_umax(key);
    const unsigned long long hit = __ballot(key == kmax && lane < r - k);
    const int piv = hit ? k + (__ffsll((long long)hit) - 1) : k;
    __syncthreads();                                   // all scans done before rows move
    if (piv != k) {                                    // (uniform: every wave found the same pivot)
      if (tid <= r) {
        const double t = Mx[k][tid];
        Mx[k][tid] = Mx[piv][tid];
        Mx[piv][tid] = t;
      }
      __syncthreads();
    }
    const double rpk = fast_rcp(Mx[k][k]);
    // rows k+1..r-1, columns k+1..r (incl. rhs; at most 64 of them): a thread keeps one column and every fourth row; all its LDS
    // reads are issued before the first use (the step is a few LDS latencies long, not one per row), no integer division per entry
    {
      static_assert(CAP <= 64, "one column per lane of a 64-column group");
      constexpr int NI = (CAP + 3) / 4;
      const int j = k + 1 + (tid & 63), i0 = k + 1 + (tid >> 6);
      if (j <= r && i0 < r) {
        const double pj = Mx[k][j];
        double mik[NI], mij[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
          const int i = i0 + 4 * u, ic = i < r ? i : r - 1;
          mik[u] = Mx[ic][k];
          mij[u] = Mx[ic][j];
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
          const int i = i0 + 4 * u;
          if (i < r) Mx[i][j] = mij[u] - (mik[u] * rpk) * pj;
        }
      }
    }
    __syncthreads();
  }
  for (int k = r - 1; k >= 0; --k) {
    const double xk = Mx[k][r] * fast_rcp(Mx[k][k]);      // row k is final: rows > k were folded in already
    if (tid == 0) xs[k] = xk;
    if (tid < k) Mx[tid][r] -= Mx[tid][k] * xk;
    __syncthreads();
  }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
}

template <int RG>
__device__ __forceinline__ double lu_solve_rows(double (&a)[RG + 1], int r, int lane);      // (defined with the register-row kernels below)

// LDS of the block-cooperative r x r algebra, r <= CAP.
template <int CAP>
struct CoefBlockLds {
  double A[CAP][CAP + 1];    // U'U
  double B[CAP][CAP + 1];    // V'V
  double Cm[CAP][CAP + 1];   // V'U  (psgd.py:574)
  double Mx[CAP][CAP + 2];   // augmented elimination matrix
  double ut[CAP], uw[CAP], vt[CAP], vw[CAP], s1[CAP], s2[CAP], x1[CAP], x2[CAP], p2[CAP], cs1[CAP];
  double e1[CAP], e2[CAP];
  double sc[3];
};

// One 256-thread block.  Reads the reduced Gram of W = [U | V | t | w] -- DENSE = false: the MFMA block layout of
// k_update_gram; DENSE = true: row-major [2r + 2][2r + 2] (k_gram_wide_finish, ranks 33 .. 64) -- and produces the
// coefficient block of UpdCoef.  fp64 throughout.
// ROWLU: the two solves by ONE wave with a row of the matrix per lane in registers (lu_solve_rows: no barrier, no LDS traffic in the
// elimination) instead of the block-cooperative elimination in LDS -- at r = 64 the latter is 77 us per solve, two barriers and a chain
// of LDS latencies per step.
template <int CAP, bool DENSE, bool ROWLU = false, int RG = CAP>       // RG: columns a lane's register row holds (r <= RG <= CAP)
__device__ void coef_block_ref(const double* __restrict__ gram, int r, float step, float tiny, int update_U,
                               float* __restrict__ coef, double* __restrict__ c64, CoefBlockLds<CAP>& L) {
  auto& A = L.A; auto& B = L.B; auto& Cm = L.Cm; auto& Mx = L.Mx;
  auto& ut = L.ut; auto& uw = L.uw; auto& vt = L.vt; auto& vw = L.vw; auto& s1 = L.s1; auto& s2 = L.s2;
  auto& x1 = L.x1; auto& x2 = L.x2; auto& p2 = L.p2; auto& cs1 = L.cs1; auto& e1 = L.e1; auto& e2 = L.e2; auto& sc = L.sc;
  const int tid = threadIdx.x;
  const int nb = (2 * r + 2 + 15) / 16;
  const int ncol = 2 * r + 2;
  auto G = [&](int a, int b) -> double {
    if constexpr (DENSE) return gram[a * ncol + b];
    if (a > b) { const int t = a; a = b; b = t; }
    const int bi = a >> 4, bj = b >> 4, i = a & 15, j = b & 15;
    const int p = bi * nb - (bi * (bi - 1)) / 2 + (bj - bi);
    return gram[p * 256 + (i & 3) * 64 + (((i >> 2) << 4) | j)];
  };
  for (int idx = tid; idx < r * r; idx += kThreads) {
    const int i = idx / r, j = idx % r;
    A[i][j] = G(i, j);
    B[i][j] = G(r + i, r + j);
    Cm[i][j] = DENSE ? G(r + i, j) : G(j, r + i);      // (the dense Gram is symmetric to the bit: the coalesced of the two reads)
  }
  if (tid < r) {
    ut[tid] = G(tid, 2 * r);
    uw[tid] = G(tid, 2 * r + 1);
    vt[tid] = G(r + tid, 2 * r);
    vw[tid] = G(r + tid, 2 * r + 1);
    s1[tid] = vt[tid];                                   // s1 = V't
  }
  if (tid == 0) { sc[0] = G(2 * r, 2 * r); sc[1] = G(2 * r, 2 * r + 1); sc[2] = G(2 * r + 1, 2 * r + 1); }
  __syncthreads();

  // s2 = U'Qh = U't + (U'U) s1 ; cs1 = (V'U) s1
  if (tid < r) {
    double a = ut[tid], c = 0.0;
    for (int k = 0; k < r; ++k) { a += A[tid][k] * s1[k]; c += Cm[tid][k] * s1[k]; }
    s2[tid] = a;
    cs1[tid] = c;
  }
  if constexpr (ROWLU) {
    __syncthreads();
    if (tid >= 64) return;
    const bool actl = tid < r;
    double a[RG + 1];
    // x1 = solve(K', U'w): lane i holds row i of K' = column i of K = I + V'U
#pragma unroll
    for (int j = 0; j < RG; ++j) a[j] = (actl && j < r) ? Cm[j][tid] + ((j == tid) ? 1.0 : 0.0) : 0.0;
    a[RG] = actl ? uw[tid] : 0.0;
    const double x1v = lu_solve_rows<RG>(a, r, tid);
    if (actl) x1[tid] = x1v;
    __builtin_amdgcn_wave_barrier();
    // p2 = V'w - (V'V) x1 ; x2 = solve(K, p2)
    double pv = actl ? vw[tid] : 0.0;
    if (actl)
      for (int k = 0; k < r; ++k) pv -= B[tid][k] * x1[k];
    if (actl) p2[tid] = pv;
#pragma unroll
    for (int j = 0; j < RG; ++j) a[j] = (actl && j < r) ? Cm[tid][j] + ((j == tid) ? 1.0 : 0.0) : 0.0;
    a[RG] = pv;
    const double x2v = lu_solve_rows<RG>(a, r, tid);
    if (actl) x2[tid] = x2v;
    __builtin_amdgcn_wave_barrier();
  } else {
  // x1 = solve(K', U'w), K = I + V'U            (psgd.py:575-577, adjoint=True)
  for (int idx = tid; idx < r * r; idx += kThreads) {
    const int i = idx / r, j = idx % r;
    Mx[i][j] = Cm[j][i] + (i == j ? 1.0 : 0.0);
  }
  if (tid < r) Mx[tid][r] = uw[tid];
  __syncthreads();
  lu_solve_block<CAP>(Mx, r, x1);
  __syncthreads();
  // p2 = V' invQtv = V'w - (V'V) x1 ; x2 = solve(K, p2)          (psgd.py:578)
  if (tid < r) {
    double a = vw[tid];
    for (int k = 0; k < r; ++k) a -= B[tid][k] * x1[k];
    p2[tid] = a;
    Mx[tid][r] = a;
  }
  for (int idx = tid; idx < r * r; idx += kThreads) {
    const int i = idx / r, j = idx % r;
    Mx[i][j] = Cm[i][j] + (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  lu_solve_block<CAP>(Mx, r, x2);
  __syncthreads();
  }
  if (tid >= 64) return;                                  // the rest is one wave of r-vector algebra

  const int lane = tid;
  const bool act = lane < r;
  const double tt = sc[0], tw = sc[1], ww = sc[2];
  // a = Qh = t + U s1, b = invQtv = w - V x1                      (psgd.py:587)
  const double s1ut = wave_sum(act ? s1[lane] * ut[lane] : 0.0);
  const double s1s2 = wave_sum(act ? s1[lane] * s2[lane] : 0.0);
  const double x1vw = wave_sum(act ? x1[lane] * vw[lane] : 0.0);
  const double x1p2 = wave_sum(act ? x1[lane] * p2[lane] : 0.0);
  const double x1vt = wave_sum(act ? x1[lane] * vt[lane] : 0.0);
  const double s1uw = wave_sum(act ? s1[lane] * uw[lane] : 0.0);
  const double x1cs1 = wave_sum(act ? x1[lane] * cs1[lane] : 0.0);
  const double aa = tt + s1ut + s1s2;          // a'a = t't + 2 s1'U't + s1'(U'U)s1
  const double bb = ww - x1vw - x1p2;          // b'b = w'w - 2 x1'V'w + x1'(V'V)x1
  const double ab = tw - x1vt + s1uw - x1cs1;  // a'b

  // e1 = a'M, e2 = b'M with M = V (update U) or U (update V); the norm needs
  // ||M e1'||^2 = e1 (M'M) e1' etc.                       (psgd.py:589-596 / :603-610)
  double my_e1 = 0.0, my_e2 = 0.0, ub = 0.0;
  if (act) {
    ub = uw[lane];
    for (int k = 0; k < r; ++k) ub -= Cm[k][lane] * x1[k];      // U'b = U'w - (U'V) x1
    if (update_U) {
      my_e1 = vt[lane] + cs1[lane];      // atV = V't + (V'U) s1
      my_e2 = p2[lane];                  // btV = V'w - (V'V) x1
    } else {
      my_e1 = s2[lane];                  // atU = U't + (U'U) s1
      my_e2 = ub;                        // btU
    }
    e1[lane] = my_e1;
    e2[lane] = my_e2;
  }
  __builtin_amdgcn_wave_barrier();
  double g1 = 0.0, g2 = 0.0;
  if (act) {
    for (int k = 0; k < r; ++k) {
      const double m = update_U ? B[lane][k] : A[lane][k];
      g1 += m * e1[k];
      g2 += m * e2[k];
    }
  }
  const double pp = wave_sum(act ? my_e1 * g1 : 0.0);
  const double qq = wave_sum(act ? my_e2 * g2 : 0.0);
  const double pq = wave_sum(act ? my_e1 * g2 : 0.0);
  const double nrm = sqrt(fabs(aa * pp + bb * qq - 2.0 * ab * pq));
  const double mu = (double)step / (nrm + (double)tiny);
  // c1, c2: update U -> (atV K), (btV K) (psgd.py:600-601); update V -> atU, btU (:614-615)
  if (act) {
    double c1 = my_e1, c2 = my_e2;
    if (update_U) {                        // + e K's off-identity part
      for (int i = 0; i < r; ++i) { c1 += e1[i] * Cm[i][lane]; c2 += e2[i] * Cm[i][lane]; }
    }
    coef[0 * r + lane] = (float)s1[lane];
    coef[1 * r + lane] = (float)s2[lane];
    coef[2 * r + lane] = (float)x1[lane];
    coef[3 * r + lane] = (float)x2[lane];
    coef[4 * r + lane] = (float)c1;
    coef[5 * r + lane] = (float)c2;
    if (c64) {
      c64[0 * r + lane] = s2[lane];
      c64[1 * r + lane] = ub;
      c64[2 * r + lane] = (double)(float)c1;
      c64[3 * r + lane] = (double)(float)c2;
    }
  }
  if (lane == 0) {
    coef[6 * r] = (float)mu;
    coef[6 * r + 1] = (float)nrm;
    if (c64) {
      c64[4 * r + 0] = (double)(float)mu;
      c64[4 * r + 1] = aa;
      c64[4 * r + 2] = ab;
      c64[4 * r + 3] = bb;
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_update_coef(const double* __restrict__ gram, int r, float step, float tiny,
                                                          int update_U, float* __restrict__ coef,
                                                          double* __restrict__ c64) {
  __shared__ CoefBlockLds<MR> L;
  coef_block_ref<MR, false>(gram, r, step, tiny, update_U, coef, c64, L);
}

// ranks 33 .. 64 (psgd_uvd_wide_update_f32): 137 KiB of LDS, requested at the launch
template <int RG>
__global__ __launch_bounds__(kThreads) void k_wide_coef(const double* __restrict__ gram, int r, float step, float tiny,
                                                        int update_U, float* __restrict__ coef, double* __restrict__ c64) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_coef_lds[];
  coef_block_ref<2 * MR, true, true, RG>(gram, r, step, tiny, update_U, coef, c64,
                                         *reinterpret_cast<CoefBlockLds<2 * MR>*>(wide_coef_lds));
}

template <int RG>
static int wide_coef_launch_rg(hipStream_t st, const double* G, int r, float step, float tiny, int update_U, float* coef, double* c64) {
  static bool attr_done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 1;
  const int lds = (int)sizeof(CoefBlockLds<2 * MR>);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide_coef<RG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return 1;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL(k_wide_coef<RG>, dim3(1), dim3(kThreads), lds, st, G, r, step, tiny, update_U, coef, c64);
  return (int)hipGetLastError();
}
static int wide_coef_launch(hipStream_t st, const double* G, int r, float step, float tiny, int update_U, float* coef, double* c64) {
  if (r <= 40) return wide_coef_launch_rg<40>(st, G, r, step, tiny, update_U, coef, c64);      // (the register rows of the two solves:
  if (r <= 48) return wide_coef_launch_rg<48>(st, G, r, step, tiny, update_U, coef, c64);      //  work grows with RG^2)
  if (r <= 56) return wide_coef_launch_rg<56>(st, G, r, step, tiny, update_U, coef, c64);
  return wide_coef_launch_rg<64>(st, G, r, step, tiny, update_U, coef, c64);
}

// ---- the same r x r algebra, latency-oriented (what the entry points launch; k_update_coef above stays as the
// reference form, psgd_set_tuning key 2).  The block-cooperative elimination above pays three workgroup barriers per
// pivot and a shuffle tree per inner product: 16 us at r = 10, 30 us at r = 20 -- on the critical path between the
// two update sweeps, and a fifth of a whole step at N = 1M.  Here the augmented matrix lives ONE ROW PER LANE in
// registers (r <= 32 rows, the loops unrolled to RG = r rounded up to 8 with identity padding): a pivot is a DPP
// maximum + ballot, the row exchange and the pivot-row broadcast are v_readlane, an elimination step is one fp64 fma
// per column per lane, and nothing synchronises.  Element by element the arithmetic is the same as lu_solve_block's.
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Solves M x = rhs with partial pivoting; lane i < r holds row i of M in a[0..RG) (columns >= r: 0) and rhs_i in
// a[RG]; returns x_i in lane i.  Rows are never exchanged: the pivot row of step k stays in its lane (its step number
// is remembered), rows that have not been a pivot yet are eliminated against it, and the back substitution walks the
// pivots in reverse.  Value by value this is lu_solve_block's arithmetic (same pivots: first row with the largest
// |entry| among the rows still in play -- in ROW order, which is the order LAPACK's partial pivoting scans when no
// exchange has happened; after an exchange the tie order may differ, the pivot VALUE does not).
template <int RG>
__device__ __forceinline__ double lu_solve_rows(double (&a)[RG + 1], int r, int lane) {
  int mystep = (lane < r) ? RG : -1;                // step at which this lane's row was the pivot (RG: not yet)
  int pivlane[RG];
#pragma unroll
  for (int k = 0; k < RG; ++k) {
    pivlane[k] = 0;
    if (k < r) {
      const bool cand = mystep == RG;
      const unsigned key = cand ? (__float_as_uint(fabsf((float)a[k])) | 1u) : 0u;    // |1: a candidate beats "none"
      const unsigned kmax = wave64_umax(key);
      const unsigned long long hit = __ballot(cand && key == kmax);
      const int piv = __builtin_amdgcn_readfirstlane((int)(__ffsll((long long)hit) - 1));
      pivlane[k] = piv;
      const double pk = readlane_f64(a[k], piv);
      const double f = a[k] * fast_rcp(pk);          // multiplier of this lane's row
      if (lane == piv) mystep = k;
      const bool elim = mystep == RG;                // rows still in play
#pragma unroll
      for (int j = k + 1; j <= RG; ++j) {
        const double pj = readlane_f64(a[j], piv);   // pivot row, column j
        if (elim) a[j] -= f * pj;
      }
    }
  }
  double x = 0.0;
#pragma unroll
  for (int k = RG - 1; k >= 0; --k) {
    if (k < r) {
      const int piv = pivlane[k];
      const double xk = readlane_f64(a[RG], piv) * fast_rcp(readlane_f64(a[k], piv));
      if (mystep < k && mystep >= 0) a[RG] -= a[k] * xk;      // rows that were pivots of earlier steps
      if (lane == k) x = xk;
    }
  }
  return x;
}

struct CoefLds {
  double q[10][MR + 1];     // per-lane products whose sums over the lanes are needed
  double qs[10];
};

// sums over lanes 0..r-1 of NQ per-lane values: every value goes to LDS, lane q adds up row q (NQ lanes work side by
// side, the loads of a row issued together), the totals come back to all lanes.
template <int NQ, int RG>
__device__ __forceinline__ void lane_sums(double (&val)[NQ], int r, int tid, CoefLds& L) {
  if (tid < RG) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) L.q[q][tid] = (tid < r) ? val[q] : 0.0;
  }
  __syncthreads();
  if (tid < NQ) {
    double x[RG];
#pragma unroll
    for (int k = 0; k < RG; ++k) x[k] = L.q[tid][k];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < RG; ++k) s += x[k];
    L.qs[tid] = s;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) val[q] = L.qs[q];
  __syncthreads();
}

// y_i = sum_k row[k] * x_k with x_k living in lane k: the broadcast is a v_readlane pair, no LDS
template <int RG>
__device__ __forceinline__ double row_dot_lanes(const double (&row)[RG], double x) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < RG; ++k) s += row[k] * readlane_f64(x, k);
  return s;
}

// One wave (lanes tid < r carry rows; a 64-thread block).  Everything r x r stays in registers: lane i holds row i of
// U'U, V'V, V'U and U'V (columns >= r: 0), vectors live one entry per lane, a matrix-vector product is RG fma's on
// v_readlane broadcasts.
template <int RG>
__device__ void coef_block(const double* __restrict__ gram, int r, float step, float tiny, int update_U,
                           float* __restrict__ coef, double* __restrict__ c64, CoefLds& L) {
  const int tid = threadIdx.x, lane = tid & 63;
  const bool act = tid < r;
  const int nb = (2 * r + 2 + 15) / 16;
  // All 4 RG + 7 loads of a lane are issued before the first is used (branch-free, clamped indices): the kernel is one
  // memory latency long here, not one per element.
  auto Gi = [&](int a, int b) -> int {          // index of Gram entry (a, b) in the MFMA block layout of k_update_gram
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const int bi = lo >> 4, bj = hi >> 4, i = lo & 15, j = hi & 15;
    const int p = bi * nb - (bi * (bi - 1)) / 2 + (bj - bi);
    return p * 256 + (i & 3) * 64 + (((i >> 2) << 4) | j);
  };
  const int row = act ? tid : 0;
  double A[RG], B[RG], C[RG], Ct[RG];           // rows of U'U, V'V, V'U (psgd.py:574) and U'V = (V'U)'
#pragma unroll
  for (int j = 0; j < RG; ++j) {
    const int jj = j < r ? j : 0;
    A[j] = gram[Gi(row, jj)];
    B[j] = gram[Gi(r + row, r + jj)];
    C[j] = gram[Gi(jj, r + row)];
    Ct[j] = gram[Gi(row, r + jj)];
  }
  double ut = gram[Gi(row, 2 * r)], uw = gram[Gi(row, 2 * r + 1)], vt = gram[Gi(r + row, 2 * r)],
         vw = gram[Gi(r + row, 2 * r + 1)];
  const double tt = gram[Gi(2 * r, 2 * r)], tw = gram[Gi(2 * r, 2 * r + 1)], ww = gram[Gi(2 * r + 1, 2 * r + 1)];
#pragma unroll
  for (int j = 0; j < RG; ++j)
    if (!act || j >= r) A[j] = B[j] = C[j] = Ct[j] = 0.0;
  if (!act) ut = uw = vt = vw = 0.0;
  // s2 = U'Qh = U't + (U'U) s1 ; cs1 = (V'U) s1
  const double s1 = vt;                                                   // s1 = V't
  const double s2 = ut + row_dot_lanes<RG>(A, s1);
  const double cs1 = row_dot_lanes<RG>(C, s1);
  double a[RG + 1];
  // x1 = solve(K', U'w), K = I + V'U            (psgd.py:575-577, adjoint=True)
#pragma unroll
  for (int j = 0; j < RG; ++j) a[j] = Ct[j] + ((j == lane) ? 1.0 : 0.0);
  a[RG] = uw;
  const double x1 = lu_solve_rows<RG>(a, r, lane);
  // p2 = V' invQtv = V'w - (V'V) x1 ; x2 = solve(K, p2)          (psgd.py:578)
  const double p2 = vw - row_dot_lanes<RG>(B, x1);
#pragma unroll
  for (int j = 0; j < RG; ++j) a[j] = C[j] + ((j == lane) ? 1.0 : 0.0);
  a[RG] = p2;
  const double x2 = lu_solve_rows<RG>(a, r, lane);
  // a = Qh = t + U s1, b = invQtv = w - V x1                      (psgd.py:587)
  double d7[7] = {s1 * ut, s1 * s2, x1 * vw, x1 * p2, x1 * vt, s1 * uw, x1 * cs1};
  lane_sums<7, RG>(d7, r, tid, L);
  const double aa = tt + d7[0] + d7[1];          // a'a = t't + 2 s1'U't + s1'(U'U)s1
  const double bb = ww - d7[2] - d7[3];          // b'b = w'w - 2 x1'V'w + x1'(V'V)x1
  const double ab = tw - d7[4] + d7[5] - d7[6];  // a'b
  // e1 = a'M, e2 = b'M with M = V (update U) or U (update V); the norm needs ||M e1'||^2 = e1 (M'M) e1' etc.
  const double ub = uw - row_dot_lanes<RG>(Ct, x1);                      // U'b = U'w - (U'V) x1
  double e1, e2;                                   //                      (psgd.py:589-596 / :603-610)
  if (update_U) {
    e1 = vt + cs1;                                // atV = V't + (V'U) s1
    e2 = p2;                                      // btV = V'w - (V'V) x1
  } else {
    e1 = s2;                                      // atU = U't + (U'U) s1
    e2 = ub;                                      // btU
  }
  double g1, g2;
  if (update_U) { g1 = row_dot_lanes<RG>(B, e1); g2 = row_dot_lanes<RG>(B, e2); }
  else { g1 = row_dot_lanes<RG>(A, e1); g2 = row_dot_lanes<RG>(A, e2); }
  double d3[3] = {e1 * g1, e2 * g2, e1 * g2};
  lane_sums<3, RG>(d3, r, tid, L);
  const double nrm = sqrt(fabs(aa * d3[0] + bb * d3[1] - 2.0 * ab * d3[2]));
  const double mu = (double)step / (nrm + (double)tiny);
  // c1, c2: update U -> (atV K), (btV K) (psgd.py:600-601); update V -> atU, btU (:614-615)
  double c1 = e1, c2 = e2;
  if (update_U) {                        // + e K's off-identity part: (e (V'U))_i = sum_k e_k (V'U)[k][i] = (U'V)[i][.] e
    c1 += row_dot_lanes<RG>(Ct, e1);
    c2 += row_dot_lanes<RG>(Ct, e2);
  }
  if (act) {
    coef[0 * r + tid] = (float)s1;
    coef[1 * r + tid] = (float)s2;
    coef[2 * r + tid] = (float)x1;
    coef[3 * r + tid] = (float)x2;
    coef[4 * r + tid] = (float)c1;
    coef[5 * r + tid] = (float)c2;
    // what k_fused_post needs to correct U'U for the update of U: U'a (= s2), U'b, and c1, c2, mu as the floats the
    // sweep multiplies with
    c64[0 * r + tid] = s2;
    c64[1 * r + tid] = ub;
    c64[2 * r + tid] = (double)(float)c1;
    c64[3 * r + tid] = (double)(float)c2;
  }
  if (tid == 0) {
    coef[6 * r] = (float)mu;
    coef[6 * r + 1] = (float)nrm;
    c64[4 * r + 0] = (double)(float)mu;
    c64[4 * r + 1] = aa;
    c64[4 * r + 2] = ab;
    c64[4 * r + 3] = bb;
  }
}

template <int RG>
__global__ __launch_bounds__(64) void k_coef_fast(const double* __restrict__ gram, int r, float step, float tiny,
                                                  int update_U, float* __restrict__ coef, double* __restrict__ c64) {
  __shared__ CoefLds L;
  coef_block<RG>(gram, r, step, tiny, update_U, coef, c64, L);
}

static int g_tune_coef = 0;      // psgd_set_tuning key 2: 1 = the block-cooperative reference kernel

static int launch_coef(hipStream_t st, const double* gram, int r, float step, float tiny, int update_U, float* coef,
                       double* c64) {
#define PSGD_COEF_CASE(RG)                                                                                         \
  case RG / 4:                                                                                                     \
    hipLaunchKernelGGL(k_coef_fast<RG>, dim3(1), dim3(64), 0, st, gram, r, step, tiny, update_U, coef, c64);       \
    break;
  if (g_tune_coef == 1) {
    hipLaunchKernelGGL(k_update_coef, dim3(1), dim3(kThreads), 0, st, gram, r, step, tiny, update_U, coef, c64);
  } else {
    switch ((r + 3) / 4) {
      PSGD_COEF_CASE(4) PSGD_COEF_CASE(8) PSGD_COEF_CASE(12) PSGD_COEF_CASE(16)
      PSGD_COEF_CASE(20) PSGD_COEF_CASE(24) PSGD_COEF_CASE(28) PSGD_COEF_CASE(32)
      default: return (int)hipErrorInvalidValue;
    }
  }
#undef PSGD_COEF_CASE
  return (int)hipGetLastError();
}

// The r x r algebra between the fused sweep 2 and the last sweep (one block, fp64).  Inputs: the U'U block of the reduced
// Gram of sweep 1, the by-products c64 of the coefficient kernel (U'a, U'b, c1, c2, mu, a'a, a'b, b'b), the sums
// pq = [pU | pV | qU | qV] = [Unew | Vnew]' [d.*g, d.*g.*nablaD] and max|nablaD|.  Output: the two r-vectors of the
// apply on the UPDATED state (psgd.py:625-626 after :584, :600 / :614):
//   s1' = Vnew'(dnew.*g) = pV - mu_d qV
//   s2' = Unew'(dnew.*g + Unew s1') = (pU - mu_d qU) + (Unew'Unew) s1'
// Unew = U - mu (a c1' - b c2') when U was updated (psgd.py:600-601), so with Ua = U'a and Ub = U'b
//   Unew'Unew = U'U - mu (Ua c1' + c1 Ua' - Ub c2' - c2 Ub') + mu^2 (a'a c1 c1' - a'b (c1 c2' + c2 c1') + b'b c2 c2')
// and Unew'Unew = U'U when V was updated.  Writes coef[0, r) = s1', coef[r, 2r) = s2' (+ the fp64 values to s_out).
// REDUCE (single GPU, grids of <= 512 blocks): the block first reduces the sweep's partials itself (what k_reduce_pq
// does in the staged path, same order per element): one launch less between the two sweeps.
template <bool REDUCE, int CAP = MR, bool DENSE = false>       // (CAP = 64, DENSE: ranks 33 .. 64, the row-major Gram of k_gram_wide_finish)
__global__ __launch_bounds__(1024) void k_fused_post(const double* __restrict__ gram, const double* __restrict__ c64,
                                                     const double* __restrict__ part, const float* __restrict__ pmax,
                                                     int G, double* pq, float* maxbuf, int r, float step, float tiny,
                                                     int update_U, float* coef, double* s_out) {
  __shared__ double A[CAP][CAP + 1];
  __shared__ double s1n[CAP];
  __shared__ double pqs[4 * CAP];
  __shared__ float mx;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  if constexpr (REDUCE) {
    if (w == nw - 1) {
      float m = 0.0f;
      for (int b = lane; b < G; b += 64) m = amaxf(m, pmax[b]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = amaxf(m, __shfl_xor(m, off, 64));
      if (lane == 0) { mx = m; maxbuf[0] = m; pq[4 * r] = (double)m; }
    }
    for (int e = w; e < 4 * r; e += nw) {                 // G <= 512: eight loads per lane, issued together
      const double* p = part + (long)e * G;
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = (lane + 64 * u < G) ? p[lane + 64 * u] : 0.0;
      double sum = 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += x[u];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
      if (lane == 0) { pqs[e] = sum; pq[e] = sum; }
    }
  } else {
    for (int e = tid; e < 4 * r; e += blockDim.x) pqs[e] = pq[e];
    if (tid == 0) mx = maxbuf[0];
  }
  const int nb = (2 * r + 2 + 15) / 16;
  auto Gm = [&](int a, int b) -> double {
    if constexpr (DENSE) return gram[a * (2 * r + 2) + b];
    if (a > b) { const int t = a; a = b; b = t; }
    const int bi = a >> 4, bj = b >> 4, i = a & 15, j = b & 15;
    const int p = bi * nb - (bi * (bi - 1)) / 2 + (bj - bi);
    return gram[p * 256 + (i & 3) * 64 + (((i >> 2) << 4) | j)];
  };
  const bool act = tid < r;
  if (act) {
    for (int j0 = 0; j0 < r; j0 += 8) {                  // eight independent loads in flight
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = Gm(tid, (j0 + u < r) ? j0 + u : 0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + u < r) A[tid][j0 + u] = t[u];
    }
    if (update_U) {
      const double mu = c64[4 * r], aa = c64[4 * r + 1], ab = c64[4 * r + 2], bb = c64[4 * r + 3];
      const double Ua = c64[tid], Ub = c64[r + tid], c1 = c64[2 * r + tid], c2 = c64[3 * r + tid];
      for (int j = 0; j < r; ++j) {
        const double Uaj = c64[j], Ubj = c64[r + j], c1j = c64[2 * r + j], c2j = c64[3 * r + j];
        A[tid][j] += -mu * (Ua * c1j + c1 * Uaj - Ub * c2j - c2 * Ubj) +
                     mu * mu * (aa * c1 * c1j - ab * (c1 * c2j + c2 * c1j) + bb * c2 * c2j);
      }
    }
  }
  __syncthreads();
  const double mud = (double)(step / (mx + tiny));          // the float expression of k_update_d / k_uvd_final
  if (act) s1n[tid] = pqs[r + tid] - mud * pqs[3 * r + tid];          // pV - mu_d qV
  __syncthreads();
  if (act) {
    double s2 = pqs[tid] - mud * pqs[2 * r + tid];                     // pU - mu_d qU
    for (int k = 0; k < r; ++k) s2 += A[tid][k] * s1n[k];
    coef[tid] = (float)s1n[tid];
    coef[r + tid] = (float)s2;
    s_out[tid] = s1n[tid];
    s_out[r + tid] = s2;
  }
}

}  // namespace psgd

// ================================================================== C ABI ===
using namespace psgd;

#define PSGD_CHECK_LAUNCH(expr)                 \
  do {                                          \
    const int _e = (expr);                      \
    if (_e != 0) return PSGD_ERR_LAUNCH;        \
  } while (0)

static inline int last_launch() { return (int)hipGetLastError(); }

extern "C" {

int psgd_abi_version(void) { return PSGD_ABI_VERSION; }

const char* psgd_error_string(int code) {
  switch (code) {
    case PSGD_OK: return "ok";
    case PSGD_ERR_BAD_ARG: return "bad argument (null pointer or non-positive size)";
    case PSGD_ERR_RANK: return "rank of modification outside [1, 32]";
    case PSGD_ERR_WORKSPACE: return "workspace missing, too small or not 256-byte aligned";
    case PSGD_ERR_ALIGN: return "matrix pointer not 16-byte aligned";
    case PSGD_ERR_LAUNCH: return "HIP kernel launch failed";
    case PSGD_ERR_SHAPE: return "inconsistent or unsupported matrix shapes";
    default: return "unknown error";
  }
}

int psgd_set_tuning(int key, int value) {
  if (key == 0) { g_tune_staging = value; return PSGD_OK; }
  if (key == 1) { g_tune_blocks_per_cu = value; return PSGD_OK; }
  if (key == 2) { g_tune_coef = value; return PSGD_OK; }
  if (key == 3) { g_tune_tiles_per_wave = value; return PSGD_OK; }
  if (key >= 10 && key < 26) { g_tune_grid[key - 10] = value; return PSGD_OK; }
  return PSGD_ERR_BAD_ARG;
}

int psgd_prof_enable(int on) {
  g_prof_on = on ? 1 : 0;
  for (int s = 0; s < kProfSlots; ++s) g_prof_used[s] = 0;
  return PSGD_OK;
}

int psgd_prof_collect(int slot, double* total_ms, int* count) {
  if (slot < 0 || slot >= kProfSlots || !total_ms || !count) return PSGD_ERR_BAD_ARG;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i < g_prof_used[slot]; ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(g_prof_pairs[slot][i].e1) != hipSuccess) continue;
    if (hipEventElapsedTime(&ms, g_prof_pairs[slot][i].e0, g_prof_pairs[slot][i].e1) != hipSuccess) continue;
    tot += ms;
    ++n;
  }
  g_prof_used[slot] = 0;
  *total_ms = tot;
  *count = n;
  return PSGD_OK;
}

int64_t psgd_uvd_workspace_bytes(int64_t N, int r) {
  if (N <= 0) return PSGD_ERR_BAD_ARG;
  if (r < 1 || r > PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  return ws_layout(N, r).total;
}

int psgd_uvd_ws_region(int which, int stage, int64_t N, int r, int64_t* offset_bytes, int64_t* count) {
  if (!offset_bytes || !count || N <= 0) return PSGD_ERR_BAD_ARG;
  if (r < 1 || r > PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  const WsLayout L = ws_layout(N, r);
  const int nc = 2 * r + 2, nb = (nc + 15) / 16, np = nb * (nb + 1) / 2;
  if (which == PSGD_WS_SUMS_F64) {
    if (stage == 1) { *offset_bytes = L.sums_off; *count = r; return PSGD_OK; }
    if (stage == 2) { *offset_bytes = L.sums_off + (int64_t)r * 8; *count = r; return PSGD_OK; }
    if (stage == 11) { *offset_bytes = L.sums_off; *count = (int64_t)np * 256; return PSGD_OK; }
    if (stage == 13) { *offset_bytes = L.sums_off + (int64_t)kPqSumsOff * 8; *count = 4 * r; return PSGD_OK; }
  } else if (which == PSGD_WS_MAX_F32) {
    if (stage == 10) { *offset_bytes = L.max_off; *count = 2; return PSGD_OK; }
    if (stage == 12) { *offset_bytes = L.max_off + 8; *count = 1; return PSGD_OK; }
  } else if (which == PSGD_WS_SEND_F64) {
    if (stage == 1 || stage == 2 || stage == 11) return psgd_uvd_ws_region(PSGD_WS_SUMS_F64, stage, N, r, offset_bytes, count);
    if (stage == 10) { *offset_bytes = L.sums_off + (int64_t)kBalD64Off * 8; *count = 2; return PSGD_OK; }
    if (stage == 12) { *offset_bytes = L.sums_off + (int64_t)(kPqSumsOff + 4 * r) * 8; *count = 1; return PSGD_OK; }
    if (stage == 13) { *offset_bytes = L.sums_off + (int64_t)kPqSumsOff * 8; *count = 4 * r + 1; return PSGD_OK; }
  }
  return PSGD_ERR_BAD_ARG;
}

int psgd_uvd_fold_gathered_f64(int stage, const double* gathered, int world, int64_t N, int r, void* ws,
                               int64_t ws_bytes, void* stream) {
  if (!gathered || world < 1) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  int64_t off = 0, count = 0;
  const int rr = psgd_uvd_ws_region(PSGD_WS_SEND_F64, stage, N, r, &off, &count);
  if (rr) return rr;
  double* dst = reinterpret_cast<double*>(static_cast<char*>(ws) + off);
  int nsum = (int)count;
  float* maxdst = nullptr;
  if (stage == 10) { nsum = 0; maxdst = w.maxbuf; }
  else if (stage == 12) { nsum = 0; maxdst = w.maxbuf + 2; }
  else if (stage == 13) { nsum = 4 * r; maxdst = w.maxbuf + 2; }
  hipLaunchKernelGGL(k_fold_gathered, dim3(((int)count + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     gathered, world, (int)count, nsum, dst, maxdst);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

// ---------------------------------------------------------------- apply ----
int psgd_uvd_apply_sweep1_f32(const float* V, const float* d, const float* g, int64_t N, int r, void* ws,
                              int64_t ws_bytes, void* stream) {
  if (!V || !d || !g) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(V)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = sweep_grid(ops, r, kOccColreduce, N, kMaxGrid);
  float* part = static_cast<float*>(w.part);
  {
    ProfScope ps(PSGD_PROF_APPLY_S1, st);
    PSGD_CHECK_LAUNCH(ops->colreduce(use_nt(N, r), 2, V, d, g, N, part, grid, st));
  }
  hipLaunchKernelGGL(k_reduce_sum_t, dim3((r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, part, grid, r, w.sums, w.coef);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_uvd_apply_sweep2_f32(const float* U, const float* d, const float* g, float* out, int64_t N, int r,
                              int sums_reduced, void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !d || !g || !out) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (sums_reduced) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, w.sums, w.coef, r);
    PSGD_CHECK_LAUNCH(last_launch());
  }
  const int grid = sweep_grid(ops, r, kOccApplyS2, N, kMaxGrid);
  float* part = static_cast<float*>(w.part);
  {
    ProfScope ps(PSGD_PROF_APPLY_S2, st);
    PSGD_CHECK_LAUNCH(ops->apply_s2(use_nt(N, r), U, d, g, out, N, w.coef, part, grid, st));
  }
  hipLaunchKernelGGL(k_reduce_sum_t, dim3((r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, part, grid, r, w.sums + r, w.coef + r);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_uvd_apply_sweep3_f32(const float* V, const float* d, float* out, int64_t N, int r, int sums_reduced,
                              void* ws, int64_t ws_bytes, void* stream) {
  if (!V || !d || !out) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(V)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (sums_reduced) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, w.sums + r, w.coef + r, r);
    PSGD_CHECK_LAUNCH(last_launch());
  }
  const int grid = sweep_grid(ops, r, kOccApplyS3, N, kMaxGrid);
  {
    ProfScope ps(PSGD_PROF_APPLY_S3, st);
    PSGD_CHECK_LAUNCH(ops->apply_s3(use_nt(N, r), V, d, out, N, w.coef, grid, st));
  }
  return PSGD_OK;
}

int psgd_uvd_apply_f32(const float* U, const float* V, const float* d, const float* g, float* out, int64_t N, int r,
                       void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V || !d || !g || !out) return PSGD_ERR_BAD_ARG;
  int rc = psgd_uvd_apply_sweep1_f32(V, d, g, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  rc = psgd_uvd_apply_sweep2_f32(U, d, g, out, N, r, 0, ws, ws_bytes, stream);
  if (rc) return rc;
  return psgd_uvd_apply_sweep3_f32(V, d, out, N, r, 0, ws, ws_bytes, stream);
}

int psgd_uvd_ipuvt_matvec_f32(const float* U, const float* V, const float* x, float* out, int64_t N, int r,
                              void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V || !x || !out) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U) || misaligned16(V)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(w.part);
  int grid = sweep_grid(ops, r, kOccColreduce, N, kMaxGrid);
  PSGD_CHECK_LAUNCH(ops->colreduce(use_nt(N, r), 1, V, x, nullptr, N, part, grid, st));
  hipLaunchKernelGGL(k_reduce_sum_t, dim3((r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, part, grid, r, w.sums, w.coef);
  PSGD_CHECK_LAUNCH(last_launch());
  grid = sweep_grid(ops, r, kOccRowdot, N, kMaxGrid);
  PSGD_CHECK_LAUNCH(ops->rowdot_axpy(use_nt(N, r), U, x, out, N, w.coef, grid, st));
  return PSGD_OK;
}

/* IpUVtmatvec on k columns (psgd.py:540-544, "matrices or column vectors"): xs / outs are HOST arrays of k device pointers
 * to contiguous [N] columns.  U and V are swept once per group of four columns. */
int psgd_uvd_ipuvt_matvec_cols_f32(const float* U, const float* V, const float* const* xs, float* const* outs, int k,
                                   int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V || !xs || !outs || k < 1) return PSGD_ERR_BAD_ARG;
  for (int j = 0; j < k; ++j)
    if (!xs[j] || !outs[j]) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U) || misaligned16(V)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(w.part);
  for (int j0 = 0; j0 < k; j0 += 4) {
    const int nc = k - j0 < 4 ? k - j0 : 4;
    const float* x4[4];
    float* o4[4];
    for (int j = 0; j < 4; ++j) { x4[j] = xs[j0 + (j < nc ? j : 0)]; o4[j] = outs[j0 + (j < nc ? j : 0)]; }
    int grid = sweep_grid(ops, r, kOccColreduce, N, kMaxGrid);
    PSGD_CHECK_LAUNCH(ops->colreduce4(use_nt(N, r), V, x4, N, part, grid, st));
    hipLaunchKernelGGL(k_reduce_pq, dim3((4 * r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, part,
                       static_cast<const float*>(nullptr), grid, 4 * r, w.sums + kPqSumsOff, static_cast<float*>(nullptr));
    PSGD_CHECK_LAUNCH(last_launch());
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(128), 0, st, w.sums + kPqSumsOff, w.coef, 4 * r);
    PSGD_CHECK_LAUNCH(last_launch());
    grid = sweep_grid(ops, r, kOccRowdot, N, kMaxGrid);
    PSGD_CHECK_LAUNCH(ops->rowdot_axpy4(use_nt(N, r), U, x4, o4, nc, N, w.coef, grid, st));
  }
  return PSGD_OK;
}

/* precond_grad_UVd_math on the k columns of a matrix g (psgd.py:619-627; docstring :623 "either matrices or column vectors"):
 * gs / outs are HOST arrays of k device pointers to contiguous [N] columns (outs[j] may be gs[j]).  Per group of four columns the
 * three sweeps of the single-column apply (V, U, V), so U and V are read 1.5 times per FOUR columns. */
int psgd_uvd_apply_cols_f32(const float* U, const float* V, const float* d, const float* const* gs, float* const* outs, int k,
                            int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V || !d || !gs || !outs || k < 1) return PSGD_ERR_BAD_ARG;
  for (int j = 0; j < k; ++j)
    if (!gs[j] || !outs[j]) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U) || misaligned16(V)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(w.part);
  const int nt = use_nt(N, r);
  auto reduce_publish = [&](int grid) -> int {
    hipLaunchKernelGGL(k_reduce_pq, dim3((4 * r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, part,
                       static_cast<const float*>(nullptr), grid, 4 * r, w.sums + kPqSumsOff, static_cast<float*>(nullptr));
    PSGD_CHECK_LAUNCH(last_launch());
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(128), 0, st, w.sums + kPqSumsOff, w.coef, 4 * r);
    PSGD_CHECK_LAUNCH(last_launch());
    return PSGD_OK;
  };
  for (int j0 = 0; j0 < k; j0 += 4) {
    const int nc = k - j0 < 4 ? k - j0 : 4;
    const float* x4[4];
    float* o4[4];
    for (int j = 0; j < 4; ++j) { x4[j] = gs[j0 + (j < nc ? j : 0)]; o4[j] = outs[j0 + (j < nc ? j : 0)]; }
    int grid = sweep_grid(ops, r, kOccColreduce, N, kMaxGrid);
    PSGD_CHECK_LAUNCH(ops->apply4_s1(nt, V, d, x4, N, part, grid, st));
    int rr = reduce_publish(grid);
    if (rr) return rr;
    grid = sweep_grid(ops, r, kOccRowdot, N, kMaxGrid);
    PSGD_CHECK_LAUNCH(ops->apply4_s2(nt, U, d, x4, o4, nc, N, w.coef, part, grid, st));
    rr = reduce_publish(grid);
    if (rr) return rr;
    PSGD_CHECK_LAUNCH(ops->apply4_s3(nt, V, d, o4, nc, N, w.coef, grid, st));
  }
  return PSGD_OK;
}

/* ---- ranks 33 .. 64 on the whole [N, r] matrices (round 5; uvd_wide_group.hip) -------------------------------------------------
 * scratch (caller-owned, 256-byte aligned, psgd_uvd_wide_scratch_bytes): [part: 4 r kMaxGrid doubles | sums: 4 r doubles | coef: 4 r floats] */
struct WideWs { double* part; double* sums; float* coef; };
static int64_t wide_scratch_layout(int r, char* base, WideWs* w) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off = align256(off + bytes); return p; };
  double* part = reinterpret_cast<double*>(take((int64_t)4 * r * kMaxGrid * 8));
  double* sums = reinterpret_cast<double*>(take((int64_t)4 * r * 8));
  float* coef = reinterpret_cast<float*>(take((int64_t)4 * r * 4));
  if (w) { w->part = part; w->sums = sums; w->coef = coef; }
  return off;
}
static int wide_grid(const UvdWideOps* ops, int64_t N) {
  const int64_t tiles = (N + ops->tile_rows - 1) / ops->tile_rows;
  int64_t grid = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t cap = (int64_t)num_cus() * 2;        // 60-70 KiB of LDS per workgroup: two per CU
  if (grid > cap) grid = cap;
  if (grid > kMaxGrid) grid = kMaxGrid;
  return grid < 1 ? 1 : (int)grid;
}
static int wide_open(const float* A, const float* B, int64_t N, int r, void* scratch, int64_t scratch_bytes, const UvdWideOps** ops,
                     WideWs* w) {
  if (N <= 0) return PSGD_ERR_BAD_ARG;
  *ops = uvd_wide_ops_for_rank(r);
  if (!*ops) return PSGD_ERR_RANK;
  if (w) {
    if (!scratch || (reinterpret_cast<uintptr_t>(scratch) & 255) || scratch_bytes < wide_scratch_layout(r, nullptr, nullptr)) return PSGD_ERR_WORKSPACE;
    wide_scratch_layout(r, static_cast<char*>(scratch), w);
  }
  if (misaligned16(A) || (B && misaligned16(B))) return PSGD_ERR_ALIGN;      // (the tile loads are 16 / 8 / 4 bytes wide by rank)
  return PSGD_OK;
}

int64_t psgd_uvd_wide_scratch_bytes(int64_t N, int r) {
  if (N <= 0 || r <= PSGD_UVD_MAX_RANK || r > 2 * PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  return wide_scratch_layout(r, nullptr, nullptr);
}

/* precond_grad_UVd_math for 32 < r <= 64 on k columns (k = 1: the column-vector call): the three sweeps of psgd_uvd_apply_cols_f32. */
int psgd_uvd_wide_apply_cols_f32(const float* U, const float* V, const float* d, const float* const* gs, float* const* outs, int k,
                                 int64_t N, int r, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!U || !V || !d || !gs || !outs || k < 1) return PSGD_ERR_BAD_ARG;
  for (int j = 0; j < k; ++j)
    if (!gs[j] || !outs[j]) return PSGD_ERR_BAD_ARG;
  const UvdWideOps* ops;
  WideWs w;
  const int rc = wide_open(U, V, N, r, scratch, scratch_bytes, &ops, &w);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nt = use_nt(N, r), grid = wide_grid(ops, N);
  auto reduce_publish = [&]() -> int {
    hipLaunchKernelGGL(k_reduce_pq, dim3((4 * r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, w.part,
                       static_cast<const float*>(nullptr), grid, 4 * r, w.sums, static_cast<float*>(nullptr));
    PSGD_CHECK_LAUNCH(last_launch());
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, st, w.sums, w.coef, 4 * r);
    PSGD_CHECK_LAUNCH(last_launch());
    return PSGD_OK;
  };
  for (int j0 = 0; j0 < k; j0 += 4) {
    const int nc = k - j0 < 4 ? k - j0 : 4;
    const float* x4[4];
    float* o4[4];
    for (int j = 0; j < 4; ++j) { x4[j] = gs[j0 + (j < nc ? j : 0)]; o4[j] = outs[j0 + (j < nc ? j : 0)]; }
    PSGD_CHECK_LAUNCH(ops->apply4_s1(nt, V, d, x4, N, w.part, grid, st));
    int rr = reduce_publish();
    if (rr) return rr;
    PSGD_CHECK_LAUNCH(ops->apply4_s2(nt, U, d, x4, o4, nc, N, w.coef, w.part, grid, st));
    rr = reduce_publish();
    if (rr) return rr;
    PSGD_CHECK_LAUNCH(ops->apply4_s3(nt, V, d, o4, nc, N, w.coef, grid, st));
  }
  return PSGD_OK;
}

/* S[j][:] = M' x_j (fp64, device [k][r]);  out_j = x_j + M S_j (S fp32 [k][r]);  M <- M - (a c1' - b c2') (c = [c1 | c2]): the building
 * blocks of psgd_uvd_colsums_f32 / _axpy_cols_f32 / _rank2_update_f32 for 32 < r <= 64, on the whole contiguous [N, r] matrix. */
int psgd_uvd_wide_colsums_f32(const float* M, const float* const* xs, int k, double* S, int64_t N, int r, void* scratch,
                              int64_t scratch_bytes, void* stream) {
  if (!M || !xs || !S || k < 1) return PSGD_ERR_BAD_ARG;
  const UvdWideOps* ops;
  WideWs w;
  const int rc = wide_open(M, nullptr, N, r, scratch, scratch_bytes, &ops, &w);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = wide_grid(ops, N);
  for (int j0 = 0; j0 < k; j0 += 4) {
    const int nc = k - j0 < 4 ? k - j0 : 4;
    const float* x4[4];
    for (int j = 0; j < 4; ++j) {
      x4[j] = xs[j0 + (j < nc ? j : 0)];
      if (!x4[j]) return PSGD_ERR_BAD_ARG;
    }
    PSGD_CHECK_LAUNCH(ops->colreduce4(use_nt(N, r), M, x4, N, w.part, grid, st));
    hipLaunchKernelGGL(k_reduce_pq, dim3((nc * r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, w.part,
                       static_cast<const float*>(nullptr), grid, nc * r, S + (int64_t)j0 * r, static_cast<float*>(nullptr));
    PSGD_CHECK_LAUNCH(last_launch());
  }
  return PSGD_OK;
}

int psgd_uvd_wide_axpy_cols_f32(const float* M, const float* const* xs, float* const* outs, int k, const float* S, int64_t N, int r,
                                void* stream) {
  if (!M || !xs || !outs || !S || k < 1) return PSGD_ERR_BAD_ARG;
  const UvdWideOps* ops;
  const int rc = wide_open(M, nullptr, N, r, nullptr, 0, &ops, nullptr);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = wide_grid(ops, N);
  for (int j0 = 0; j0 < k; j0 += 4) {
    const int nc = k - j0 < 4 ? k - j0 : 4;
    const float* x4[4];
    float* o4[4];
    for (int j = 0; j < 4; ++j) {
      x4[j] = xs[j0 + (j < nc ? j : 0)];
      o4[j] = outs[j0 + (j < nc ? j : 0)];
      if (!x4[j] || !o4[j]) return PSGD_ERR_BAD_ARG;
    }
    PSGD_CHECK_LAUNCH(ops->rowdot_axpy4(use_nt(N, r), M, x4, o4, nc, N, S + (int64_t)j0 * r, grid, st));
  }
  return PSGD_OK;
}

int psgd_uvd_wide_rank2_update_f32(float* M, const float* a, const float* b, const float* c, int64_t N, int r, void* stream) {
  if (!M || !a || !b || !c) return PSGD_ERR_BAD_ARG;
  const UvdWideOps* ops;
  const int rc = wide_open(M, nullptr, N, r, nullptr, 0, &ops, nullptr);
  if (rc) return rc;
  PSGD_CHECK_LAUNCH(ops->rank2_update(use_nt(N, r), M, a, b, N, c, wide_grid(ops, N), static_cast<hipStream_t>(stream)));
  return PSGD_OK;
}

static int flat_grid(int64_t n);

// ---- the whole update for 32 < r <= 64 on one GPU: Gram (one sweep over U and V, uvd_wide_gram.hip) -> r x r algebra (one block,
// fp64) -> sweep 2 (reads U and V, writes the updated factor and nablaD) -> d.  5r + 10 floats per row, as for the specialised ranks.
struct WideUpd { char* gram_scr; int64_t gram_bytes; double* G; float* coef; float* nabla; float* pmax; float* maxbuf; };
static int64_t wide_update_layout(int64_t N, int r, char* base, WideUpd* w) {
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off = align256(off + bytes); return p; };
  const int64_t gb = psgd_uvd_gram_wide_scratch_bytes(N, r);
  const int ncol = 2 * r + 2;
  char* gs = take(gb);
  double* G = reinterpret_cast<double*>(take((int64_t)ncol * ncol * 8));
  float* coef = reinterpret_cast<float*>(take((int64_t)(6 * r + 2) * 4));
  float* nabla = reinterpret_cast<float*>(take(N * 4));
  float* pmax = reinterpret_cast<float*>(take((int64_t)kMaxGrid * 4));
  float* maxbuf = reinterpret_cast<float*>(take(16));
  if (w) { w->gram_scr = gs; w->gram_bytes = gb; w->G = G; w->coef = coef; w->nabla = nabla; w->pmax = pmax; w->maxbuf = maxbuf; }
  return off;
}

// ---- update followed by the apply on the updated state (the UVd.step pattern, psgd.py:732 -> :748) for 32 < r <= 64: the fused
// sequence of the specialised ranks -- sweep 2 also reduces [Unew | Vnew]' [d.*g, d.*g.*nablaD], k_fused_post turns those sums and the
// Gram into the two r-vectors of the apply, one last sweep updates d and forms the output: U and V are read three times, not four.
struct WideStep { WideUpd u; double* part_pq; double* c64; double* pq; double* post; };
static int64_t wide_step_layout(int64_t N, int r, char* base, WideStep* w) {
  int64_t off = wide_update_layout(N, r, base, w ? &w->u : nullptr);
  auto take = [&](int64_t bytes) { char* p = base ? base + off : nullptr; off = align256(off + bytes); return p; };
  double* part_pq = reinterpret_cast<double*>(take((int64_t)4 * r * 512 * 8));     // (grids of at most 512 workgroups: k_fused_post<true>)
  double* c64 = reinterpret_cast<double*>(take((int64_t)(4 * r + 4) * 8));
  double* pq = reinterpret_cast<double*>(take((int64_t)(4 * r + 1) * 8));
  double* post = reinterpret_cast<double*>(take((int64_t)2 * r * 8));
  if (w) { w->part_pq = part_pq; w->c64 = c64; w->pq = pq; w->post = post; }
  return off;
}

int64_t psgd_uvd_wide_update_apply_scratch_bytes(int64_t N, int r) {
  if (N <= 0 || r <= PSGD_UVD_MAX_RANK || r > 2 * PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  return wide_step_layout(N, r, nullptr, nullptr);
}

static int wide_s2_grid(const UvdWideOps* ops, int64_t N) {       // a tile of U and one of V per wave: one workgroup per CU
  const int64_t tiles = (N + ops->tile_rows - 1) / ops->tile_rows;
  int64_t grid = (tiles + kWavesPerBlock - 1) / kWavesPerBlock;
  if (grid > num_cus()) grid = num_cus();
  if (grid > 512) grid = 512;
  return grid < 1 ? 1 : (int)grid;
}

int psgd_uvd_wide_update_apply_f32(float* U, float* V, float* d, const float* v, const float* h, const float* g, float* out, int64_t N,
                                   int r, float step, float tiny, int update_U, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!U || !V || !d || !v || !h || !g || !out) return PSGD_ERR_BAD_ARG;
  const UvdWideOps* ops;
  int rc = wide_open(U, V, N, r, nullptr, 0, &ops, nullptr);
  if (rc) return rc;
  if (misaligned16(d)) return PSGD_ERR_ALIGN;
  if (!scratch || (reinterpret_cast<uintptr_t>(scratch) & 255) || scratch_bytes < wide_step_layout(N, r, nullptr, nullptr))
    return PSGD_ERR_WORKSPACE;
  WideStep w;
  wide_step_layout(N, r, static_cast<char*>(scratch), &w);
  hipStream_t st = static_cast<hipStream_t>(stream);
  rc = psgd_uvd_gram_wide_f32(U, V, d, v, h, N, r, w.u.G, w.u.gram_scr, w.u.gram_bytes, stream);
  if (rc) return rc;
  PSGD_CHECK_LAUNCH(wide_coef_launch(st, w.u.G, r, step, tiny, update_U, w.u.coef, w.c64));
  const int grid = wide_s2_grid(ops, N);
  PSGD_CHECK_LAUNCH(ops->update_s2(use_nt(N, r), update_U, U, V, d, v, h, g, N, w.u.coef, w.u.nabla, w.u.pmax, w.part_pq, grid, st));
  hipLaunchKernelGGL((k_fused_post<true, 2 * MR, true>), dim3(1), dim3(1024), 0, st, w.u.G, w.c64, w.part_pq, w.u.pmax, grid, w.pq,
                     w.u.maxbuf, r, step, tiny, update_U, w.u.coef, w.post);
  PSGD_CHECK_LAUNCH(last_launch());
  PSGD_CHECK_LAUNCH(ops->final_sweep(use_nt(N, r), U, V, d, w.u.nabla, g, out, N, w.u.coef, w.u.maxbuf, step, tiny, grid, st));
  return PSGD_OK;
}

int64_t psgd_uvd_wide_update_scratch_bytes(int64_t N, int r) {
  if (N <= 0 || r <= PSGD_UVD_MAX_RANK || r > 2 * PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  return wide_update_layout(N, r, nullptr, nullptr);
}

int psgd_uvd_wide_update_f32(float* U, float* V, float* d, const float* v, const float* h, int64_t N, int r, float step, float tiny,
                             int update_U, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!U || !V || !d || !v || !h) return PSGD_ERR_BAD_ARG;
  const UvdWideOps* ops;
  int rc = wide_open(U, V, N, r, nullptr, 0, &ops, nullptr);
  if (rc) return rc;
  if (misaligned16(d)) return PSGD_ERR_ALIGN;
  if (!scratch || (reinterpret_cast<uintptr_t>(scratch) & 255) || scratch_bytes < wide_update_layout(N, r, nullptr, nullptr))
    return PSGD_ERR_WORKSPACE;
  WideUpd w;
  wide_update_layout(N, r, static_cast<char*>(scratch), &w);
  hipStream_t st = static_cast<hipStream_t>(stream);
  rc = psgd_uvd_gram_wide_f32(U, V, d, v, h, N, r, w.G, w.gram_scr, w.gram_bytes, stream);
  if (rc) return rc;
  PSGD_CHECK_LAUNCH(wide_coef_launch(st, w.G, r, step, tiny, update_U, w.coef, nullptr));
  const int grid = wide_s2_grid(ops, N);
  PSGD_CHECK_LAUNCH(ops->update_s2(use_nt(N, r), update_U, U, V, d, v, h, nullptr, N, w.coef, w.nabla, w.pmax, nullptr, grid, st));
  hipLaunchKernelGGL(k_update_d, dim3(flat_grid(N)), dim3(kThreads), 0, st, d, w.nabla, (long)N, w.maxbuf, w.pmax, grid, step, tiny);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

/* Building blocks of the wide-rank path (r > PSGD_UVD_MAX_RANK; psgd_tf_amd/uvd_wide.py works on column chunks of U and V,
 * each a contiguous [N, rc] matrix with rc <= 32).  xs / outs: HOST arrays of k device pointers to contiguous [N] vectors.
 *   colsums:   S[j][:] = M' x_j                 (fp64, device [k][r])
 *   axpy_cols: out_j   = x_j + M S_j            (S fp32, device [k][r]; out_j may be x_j)
 *   rank2:     M      <- M - (a c1' - b c2')    (c = [c1 | c2] fp32, device [2r])                                  */
// a strided view [N, r] with row stride ld (floats): usable by the *_ld kernels when ld and the base are multiples of the rank's
// access width (4, 2 or 1 floats)
static inline bool view_misaligned(const void* p, int64_t ld, int load_vec) {
  return ld < 1 || (ld % load_vec) != 0 || (reinterpret_cast<uintptr_t>(p) % (4u * (unsigned)load_vec)) != 0;
}

int psgd_uvd_colsums_ld_f32(const float* M, int64_t ld, const float* const* xs, int k, double* S, int64_t N, int r, void* ws,
                            int64_t ws_bytes, void* stream) {
  if (!M || !xs || !S || k < 1 || ld < r) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  const bool strided = ld != r;
  if (strided ? view_misaligned(M, ld, ops->load_vec) : misaligned16(M)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(w.part);
  for (int j0 = 0; j0 < k; j0 += 4) {
    const int nc = k - j0 < 4 ? k - j0 : 4;
    const float* x4[4];
    for (int j = 0; j < 4; ++j) {
      x4[j] = xs[j0 + (j < nc ? j : 0)];
      if (!x4[j]) return PSGD_ERR_BAD_ARG;
    }
    const int grid = sweep_grid(ops, r, kOccColreduce, N, kMaxGrid);
    if (strided) PSGD_CHECK_LAUNCH(ops->colreduce4_ld(M, ld, x4, N, part, grid, st));
    else PSGD_CHECK_LAUNCH(ops->colreduce4(use_nt(N, r), M, x4, N, part, grid, st));
    hipLaunchKernelGGL(k_reduce_pq, dim3((nc * r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, part,
                       static_cast<const float*>(nullptr), grid, nc * r, S + (int64_t)j0 * r, static_cast<float*>(nullptr));
    PSGD_CHECK_LAUNCH(last_launch());
  }
  return PSGD_OK;
}
int psgd_uvd_colsums_f32(const float* M, const float* const* xs, int k, double* S, int64_t N, int r, void* ws,
                         int64_t ws_bytes, void* stream) {
  return psgd_uvd_colsums_ld_f32(M, r, xs, k, S, N, r, ws, ws_bytes, stream);
}

int psgd_uvd_axpy_cols_ld_f32(const float* M, int64_t ld, const float* const* xs, float* const* outs, int k, const float* S,
                              int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!M || !xs || !outs || !S || k < 1 || ld < r) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  const bool strided = ld != r;
  if (strided ? view_misaligned(M, ld, ops->load_vec) : misaligned16(M)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int j0 = 0; j0 < k; j0 += 4) {
    const int nc = k - j0 < 4 ? k - j0 : 4;
    const float* x4[4];
    float* o4[4];
    for (int j = 0; j < 4; ++j) {
      x4[j] = xs[j0 + (j < nc ? j : 0)];
      o4[j] = outs[j0 + (j < nc ? j : 0)];
      if (!x4[j] || !o4[j]) return PSGD_ERR_BAD_ARG;
    }
    const int grid = sweep_grid(ops, r, kOccRowdot, N, kMaxGrid);
    if (strided) PSGD_CHECK_LAUNCH(ops->rowdot_axpy4_ld(M, ld, x4, o4, nc, N, S + (int64_t)j0 * r, grid, st));
    else PSGD_CHECK_LAUNCH(ops->rowdot_axpy4(use_nt(N, r), M, x4, o4, nc, N, S + (int64_t)j0 * r, grid, st));
  }
  return PSGD_OK;
}
int psgd_uvd_axpy_cols_f32(const float* M, const float* const* xs, float* const* outs, int k, const float* S, int64_t N,
                           int r, void* ws, int64_t ws_bytes, void* stream) {
  return psgd_uvd_axpy_cols_ld_f32(M, r, xs, outs, k, S, N, r, ws, ws_bytes, stream);
}

int psgd_uvd_rank2_update_ld_f32(float* M, int64_t ld, const float* a, const float* b, const float* c, int64_t N, int r,
                                 void* ws, int64_t ws_bytes, void* stream) {
  if (!M || !a || !b || !c || ld < r) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  const bool strided = ld != r;
  if (strided ? view_misaligned(M, ld, ops->load_vec) : misaligned16(M)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = sweep_grid(ops, r, kOccRowdot, N, kMaxGrid);
  if (strided) PSGD_CHECK_LAUNCH(ops->rank2_update_ld(M, ld, a, b, N, c, grid, st));
  else PSGD_CHECK_LAUNCH(ops->rank2_update(use_nt(N, r), M, a, b, N, c, grid, st));
  return PSGD_OK;
}
int psgd_uvd_rank2_update_f32(float* M, const float* a, const float* b, const float* c, int64_t N, int r, void* ws,
                              int64_t ws_bytes, void* stream) {
  return psgd_uvd_rank2_update_ld_f32(M, r, a, b, c, N, r, ws, ws_bytes, stream);
}

// --------------------------------------------------------------- update ----
static int flat_grid(int64_t n) {
  int64_t g = (n / 4 + kThreads - 1) / kThreads;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (g > cap) g = cap;
  if (g > kMaxGrid) g = kMaxGrid;
  if (g < 1) g = 1;
  return (int)g;
}

int psgd_uvd_balance_max_f32(const float* U, const float* V, int64_t N, int r, void* ws, int64_t ws_bytes,
                             void* stream) {
  if (!U || !V) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U) || misaligned16(V)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n = N * r;
  const int grid = flat_grid(n);
  hipLaunchKernelGGL(k_maxabs2, dim3(grid), dim3(kThreads), 0, st, U, V, (long)n, w.pmax, grid);
  PSGD_CHECK_LAUNCH(last_launch());
  hipLaunchKernelGGL(k_reduce_max, dim3(2), dim3(kThreads), 0, st, w.pmax, grid, grid, w.maxbuf, w.sums + kBalD64Off);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_uvd_balance_scale_f32(float* U, float* V, int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U) || misaligned16(V)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n = N * r;
  hipLaunchKernelGGL(k_scale2, dim3(flat_grid(n)), dim3(kThreads), 0, st, U, V, (long)n, w.maxbuf);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_uvd_update_sweep1_ld_f32(const float* U, int64_t ldU, const float* V, int64_t ldV, const float* d, const float* v,
                                  const float* h, int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V || !d || !v || !h || ldU < r || ldV < r) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  const bool strided = ldU != r || ldV != r;
  if (strided ? (view_misaligned(U, ldU, ops->load_vec) || view_misaligned(V, ldV, ops->load_vec))
              : (misaligned16(U) || misaligned16(V))) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = sweep_grid(ops, r, kOccGram, N, kGramMaxGrid);
  double* part = static_cast<double*>(w.part);
  {
    ProfScope ps(PSGD_PROF_UPDATE_S1, st);
    if (strided) PSGD_CHECK_LAUNCH(ops->update_gram_ld(U, ldU, V, ldV, d, v, h, N, part, grid, st));
    else PSGD_CHECK_LAUNCH(ops->update_gram(use_nt(N, r), U, V, d, v, h, N, part, grid, st));
  }
  const int L = ops->gram_len;
  hipLaunchKernelGGL((k_reduce_sum<double>), dim3((L + 63) / 64), dim3(kThreads), 0, st, part, grid, L, w.sums,
                     static_cast<float*>(nullptr));
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_uvd_update_sweep1_f32(const float* U, const float* V, const float* d, const float* v, const float* h,
                               int64_t N, int r, void* ws, int64_t ws_bytes, void* stream) {
  return psgd_uvd_update_sweep1_ld_f32(U, r, V, r, d, v, h, N, r, ws, ws_bytes, stream);
}

static int update_sweep2_impl(float* U, float* V, const float* d, const float* v, const float* h, const float* g,
                              int64_t N, int r, float step, float tiny, int update_U, void* ws, int64_t ws_bytes,
                              void* stream, bool single_gpu_tail = false) {
  if (!U || !V || !d || !v || !h) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U) || misaligned16(V)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PSGD_CHECK_LAUNCH(launch_coef(st, w.sums, r, step, tiny, update_U, w.coef, w.sums + kCoef64Off));
  const int grid = sweep_grid(ops, r, g ? kOccUpdS2F : (update_U ? kOccUpdS2U : kOccUpdS2V), N, kMaxGrid);
  double* part = static_cast<double*>(w.part);
  {
    ProfScope ps(PSGD_PROF_UPDATE_S2, st);
    PSGD_CHECK_LAUNCH(ops->update_s2(use_nt(N, r), update_U, U, V, d, v, h, g, N, w.coef, w.nabla, w.pmax, part, grid, st));
  }
  if (g && single_gpu_tail && grid <= 512) {     // small grids: the post kernel reduces the partials itself
    hipLaunchKernelGGL(k_fused_post<true>, dim3(1), dim3(1024), 0, st, w.sums, w.sums + kCoef64Off, part, w.pmax, grid,
                       w.sums + kPqSumsOff, w.maxbuf + 2, r, step, tiny, update_U, w.coef, w.sums + kPostSumsOff);
    PSGD_CHECK_LAUNCH(last_launch());
    return PSGD_OK;
  }
  if (g) {      // block maxima -> max|nablaD|, column-sum partials -> [pU | pV | qU | qV]: one launch, one send region
    hipLaunchKernelGGL(k_reduce_pq, dim3((4 * r + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kThreads), 0, st, part,
                       w.pmax, grid, 4 * r, w.sums + kPqSumsOff, w.maxbuf + 2);
    PSGD_CHECK_LAUNCH(last_launch());
    if (single_gpu_tail) {
      hipLaunchKernelGGL(k_fused_post<false>, dim3(1), dim3(kThreads), 0, st, w.sums, w.sums + kCoef64Off,
                         static_cast<const double*>(nullptr), static_cast<const float*>(nullptr), 0, w.sums + kPqSumsOff,
                         w.maxbuf + 2, r, step, tiny, update_U, w.coef, w.sums + kPostSumsOff);
      PSGD_CHECK_LAUNCH(last_launch());
    }
    return PSGD_OK;
  }
  if (single_gpu_tail) {      // the d update folds the block maxima itself
    {
      ProfScope ps(PSGD_PROF_UPDATE_S3, st);
      hipLaunchKernelGGL(k_update_d, dim3(flat_grid(N)), dim3(kThreads), 0, st, const_cast<float*>(d), w.nabla, (long)N,
                         w.maxbuf + 2, w.pmax, grid, step, tiny);
    }
    PSGD_CHECK_LAUNCH(last_launch());
    return PSGD_OK;
  }
  hipLaunchKernelGGL(k_reduce_max, dim3(1), dim3(kThreads), 0, st, w.pmax, grid, grid, w.maxbuf + 2,
                     w.sums + kPqSumsOff + 4 * r);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_uvd_update_sweep2_f32(float* U, float* V, const float* d, const float* v, const float* h, int64_t N, int r,
                               float step, float tiny, int update_U, void* ws, int64_t ws_bytes, void* stream) {
  return update_sweep2_impl(U, V, d, v, h, nullptr, N, r, step, tiny, update_U, ws, ws_bytes, stream);
}

int psgd_uvd_update_sweep2_fused_f32(float* U, float* V, const float* d, const float* v, const float* h,
                                     const float* g, int64_t N, int r, float step, float tiny, int update_U, void* ws,
                                     int64_t ws_bytes, void* stream) {
  if (!g) return PSGD_ERR_BAD_ARG;
  return update_sweep2_impl(U, V, d, v, h, g, N, r, step, tiny, update_U, ws, ws_bytes, stream);
}

int psgd_uvd_fused_post_f32(int64_t N, int r, float step, float tiny, int update_U, void* ws, int64_t ws_bytes,
                            void* stream) {
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_fused_post<false>, dim3(1), dim3(kThreads), 0, st, w.sums, w.sums + kCoef64Off,
                     static_cast<const double*>(nullptr), static_cast<const float*>(nullptr), 0, w.sums + kPqSumsOff,
                     w.maxbuf + 2, r, step, tiny, update_U, w.coef, w.sums + kPostSumsOff);
  PSGD_CHECK_LAUNCH(last_launch());
  return PSGD_OK;
}

int psgd_uvd_fused_final_f32(const float* U, const float* V, float* d, const float* g, float* out, int64_t N, int r,
                             float step, float tiny, void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V || !d || !g || !out) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(U) || misaligned16(V)) return PSGD_ERR_ALIGN;
  const UvdOps* ops = uvd_ops_for_rank(r);
  if (!ops) return PSGD_ERR_RANK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int grid = sweep_grid(ops, r, kOccFinal, N, kMaxGrid);
  {
    ProfScope ps(PSGD_PROF_APPLY_S3, st);
    PSGD_CHECK_LAUNCH(ops->final_sweep(use_nt(N, r), U, V, d, w.nabla, g, out, N, w.coef, w.maxbuf + 2, step, tiny, grid, st));
  }
  return PSGD_OK;
}

int psgd_uvd_update_sweep3_f32(float* d, int64_t N, int r, float step, float tiny, void* ws, int64_t ws_bytes,
                               void* stream) {
  if (!d) return PSGD_ERR_BAD_ARG;
  Ws w;
  const int rc = ws_open(ws, ws_bytes, N, r, &w);
  if (rc) return rc;
  if (misaligned16(d)) return PSGD_ERR_ALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  {
    ProfScope ps(PSGD_PROF_UPDATE_S3, st);
    hipLaunchKernelGGL(k_update_d, dim3(flat_grid(N)), dim3(kThreads), 0, st, d, w.nabla, (long)N, w.maxbuf + 2,
                       static_cast<const float*>(nullptr), 0, step, tiny);
    PSGD_CHECK_LAUNCH(last_launch());
  }
  return PSGD_OK;
}

int psgd_uvd_update_f32(float* U, float* V, float* d, const float* v, const float* h, int64_t N, int r, float step,
                        float tiny, int balance, int update_U, void* ws, int64_t ws_bytes, void* stream) {
  if (!U || !V || !d || !v || !h) return PSGD_ERR_BAD_ARG;
  int rc;
  if (balance) {
    rc = psgd_uvd_balance_max_f32(U, V, N, r, ws, ws_bytes, stream);
    if (rc) return rc;
    rc = psgd_uvd_balance_scale_f32(U, V, N, r, ws, ws_bytes, stream);
    if (rc) return rc;
  }
  rc = psgd_uvd_update_sweep1_f32(U, V, d, v, h, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  // sweep 2 and, behind it, the d update (which folds the block maxima of |nablaD| itself)
  if (misaligned16(d)) return PSGD_ERR_ALIGN;
  return update_sweep2_impl(U, V, d, v, h, nullptr, N, r, step, tiny, update_U, ws, ws_bytes, stream, /*single_gpu_tail=*/true);
}

/* update_precond_UVd_math_ followed by precond_grad_UVd_math on the updated state (the UVd.step
 * pattern, psgd.py:732 -> :748): three sweeps.  Sweep 2 also reduces [Unew | Vnew]' [d.*g, d.*g.*nablaD]; both
 * reductions of the apply follow from those sums and the Gram of sweep 1 (k_fused_post), so the d update and the
 * whole apply are ONE last sweep. */
int psgd_uvd_update_apply_f32(float* U, float* V, float* d, const float* v, const float* h, const float* g, float* out,
                              int64_t N, int r, float step, float tiny, int balance, int update_U, void* ws,
                              int64_t ws_bytes, void* stream) {
  if (!U || !V || !d || !v || !h || !g || !out) return PSGD_ERR_BAD_ARG;
  int rc;
  if (balance) {
    rc = psgd_uvd_balance_max_f32(U, V, N, r, ws, ws_bytes, stream);
    if (rc) return rc;
    rc = psgd_uvd_balance_scale_f32(U, V, N, r, ws, ws_bytes, stream);
    if (rc) return rc;
  }
  rc = psgd_uvd_update_sweep1_f32(U, V, d, v, h, N, r, ws, ws_bytes, stream);
  if (rc) return rc;
  rc = update_sweep2_impl(U, V, d, v, h, g, N, r, step, tiny, update_U, ws, ws_bytes, stream, /*single_gpu_tail=*/true);
  if (rc) return rc;
  return psgd_uvd_fused_final_f32(U, V, d, g, out, N, r, step, tiny, ws, ws_bytes, stream);
}

}  // extern "C"
