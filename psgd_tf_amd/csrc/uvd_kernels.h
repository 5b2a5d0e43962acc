// uvd_kernels.h -- CDNA4 (gfx950) streaming kernels for the UVd preconditioner
//   Q = (I + U V') diag(d)      reference: psgd.py:527-627
//
// Data layout in HBM is the reference's own: U, V are [N, r] row-major fp32
// (80-byte rows at r = 20), d/g/v/h/out are [N] fp32.
//
// Execution model (one scheme for every sweep):
//   * A 64-lane wavefront owns a tile of kTileRows consecutive rows.  The tile
//     of each [N, r] operand is one contiguous span of HBM, so it is fetched
//     with fully coalesced 16-byte loads (1 KiB per wave instruction), staged
//     in a wave-private LDS buffer, and then read back one row per lane
//     (ds_read_b128 / b64 at a row stride of r dwords is bank-conflict free for
//     every r because gcd(r, 64) distinct lanes map to distinct bank groups).
//   * Waves never synchronise with each other inside the sweep loop (no
//     s_barrier): each wave prefetches its next tile into registers while it
//     computes on the current one, so ~10 KiB per wave is always in flight.
//   * Column reductions (V't etc.) are accumulated per lane, reduced across
//     the wave with shuffles and across the block through LDS in a fixed order;
//     one partial row per block goes to the workspace and a second tiny kernel
//     sums the partial rows in fp64.  No float atomics: results are bitwise
//     reproducible for a given grid.
//   * The small r-vectors a sweep consumes (s1, s2, ...) come from a uniform
//     `const float*`, which hipcc turns into scalar loads (SGPR operands).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "nanmax.h"

namespace psgd {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWave * kWavesPerBlock;
constexpr int kMaxGrid = 2048;      // sweeps never launch more blocks than this
constexpr int kGramMaxGrid = 1024;  // Gram sweep (fp64 partials are large)
#ifndef PSGD_TILE_CHUNK
#define PSGD_TILE_CHUNK 1
#endif
constexpr int kTileChunk = PSGD_TILE_CHUNK;   // consecutive tiles a wave takes before striding (sweep_rows)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Streaming-access policy.  NT = true marks the big once-per-sweep streams non-temporal
// (global_load/store ... nt): measured +6 % on apply and +2 % on update at N = 100M, r = 20
// (profiles/r01_nt_variants.txt).  It is only chosen when the operands exceed the 256 MiB
// Infinity Cache; smaller problems keep the default policy so that the three sweeps of one call
// can hit in L2 / Infinity Cache.
// The wave's index in its workgroup as a SCALAR (v_readfirstlane), and the lane index recomputed from the execution mask.  The sweeps
// of ranks 33 .. 64 run with 256 VGPRs + AGPRs; values derived from threadIdx.x that stay live across the tile loop get parked in
// AGPRs there, and in k_splu_upd_s4<41> / <47> (hipcc 7.0/7.2, -O2 and -O3 alike) the copy was written inside the divergent
// `if (tile < nfull)` region: a wave that owned the partial last tile but no whole tile came back from it with lane = 0 in every lane
// and the tail rows were never written (found by the randomised sweep, round 5).  A scalar cannot be lost that way, and the tail
// takes its lane index from the hardware again.
#ifndef PSGD_WAVE_ID_VGPR
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#else       // (A/B builds only: the form before the fix)
__device__ __forceinline__ int wave_in_block() { return (int)(threadIdx.x >> 6); }
#endif
__device__ __forceinline__ int lane_from_exec() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <bool NT, class T>
__device__ __forceinline__ void stream_store(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT, class T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}

// Tile geometry for rank R.  A tile is kTileRows = 64 * kRowsPerLane rows (one
// contiguous span of an [N, R] matrix) and is fetched with kLoadVec-float vector
// loads, 64 lanes wide.  Preference: 16-byte loads; fall back to 8- or 4-byte
// loads when a 16-byte-loadable tile would exceed 2048 floats (8 KiB) of LDS.
constexpr int kMaxTileFloats = 2048;
// (ranks 33..64 -- the building blocks of the wide-rank path, round 5 -- take 64-row tiles of up to 16 KiB)
constexpr int max_tile_floats(int R) { return R <= 32 ? kMaxTileFloats : 2 * kMaxTileFloats; }
constexpr int cfg_rows_per_lane(int R, int lv) {
  // smallest rpl in {1,2,4} with 64*rpl*R divisible by 64*lv
  return (R % lv == 0) ? 1 : ((2 * R) % lv == 0 ? 2 : 4);
}
constexpr int cfg_load_vec(int R) {
  return (64 * cfg_rows_per_lane(R, 4) * R <= max_tile_floats(R)) ? 4
       : (64 * cfg_rows_per_lane(R, 2) * R <= max_tile_floats(R)) ? 2 : 1;
}
// native clang vector types (HIP's float4 is a struct-with-union that blocks SROA of
// register arrays: the prefetch buffers ended up in scratch memory with it)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int LV> struct VecT;
template <> struct VecT<4> { typedef f32x4 type; };
template <> struct VecT<2> { typedef f32x2 type; };
template <> struct VecT<1> { typedef float type; };

template <int R>
struct Cfg {
  static constexpr int kVec = (R % 4 == 0) ? 4 : ((R % 2 == 0) ? 2 : 1);   // LDS row read width
  static constexpr int kLoadVec = cfg_load_vec(R);                          // global load width (floats)
  static constexpr int kRowsPerLane = cfg_rows_per_lane(R, kLoadVec);       // rows one lane owns per tile
  static constexpr int kTileRows = 64 * kRowsPerLane;
  static constexpr int kTileFloats = kTileRows * R;
  static constexpr int kLoadsPerLane = kTileFloats / (64 * kLoadVec);       // vector loads per lane per tile
  typedef typename VecT<kLoadVec>::type LoadT;
  static_assert(kTileFloats % (64 * kLoadVec) == 0, "tile must be whole wave-wide vector loads");
  static_assert(kTileFloats <= max_tile_floats(R), "tile exceeds LDS budget");
};

template <int R>
struct GramCfg {
  static constexpr int kCols = 2 * R + 2;                // [U | V | t | w]
  static constexpr int kBlocks = (kCols + 15) / 16;      // 16-column MFMA blocks
  static constexpr int kPairs = kBlocks * (kBlocks + 1) / 2;
  static constexpr int kLen = kPairs * 256;              // fp64 sums per Gram
};

// ---------------------------------------------------------------- staging ---
template <int R, int NMAT, int NVEC>
struct Prefetch {
  typename Cfg<R>::LoadT m[NMAT][Cfg<R>::kLoadsPerLane];
  float s[NVEC > 0 ? NVEC : 1][Cfg<R>::kRowsPerLane];
};

template <int R, int NMAT, int NVEC, bool NT>
__device__ __forceinline__ void issue_tile(Prefetch<R, NMAT, NVEC>& pf,
                                           const float* const (&mats)[NMAT],
                                           const float* const (&vecs)[NVEC > 0 ? NVEC : 1],
                                           long tile, int lane) {
  using C = Cfg<R>;
  const long row0 = tile * C::kTileRows;
#pragma unroll
  for (int m = 0; m < NMAT; ++m) {
    const typename C::LoadT* src = reinterpret_cast<const typename C::LoadT*>(mats[m] + row0 * R);
#pragma unroll
    for (int j = 0; j < C::kLoadsPerLane; ++j) pf.m[m][j] = stream_load<NT>(src + lane + 64 * j);
  }
#pragma unroll
  for (int k = 0; k < NVEC; ++k) {
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) pf.s[k][i] = stream_load<NT>(vecs[k] + row0 + lane + 64 * i);
  }
}

template <int R, int NMAT, int NVEC>
__device__ __forceinline__ void commit_tile(const Prefetch<R, NMAT, NVEC>& pf, float* lds, int lane) {
  using C = Cfg<R>;
#pragma unroll
  for (int m = 0; m < NMAT; ++m) {
    typename C::LoadT* dst = reinterpret_cast<typename C::LoadT*>(lds + m * C::kTileFloats);
#pragma unroll
    for (int j = 0; j < C::kLoadsPerLane; ++j) dst[lane + 64 * j] = pf.m[m][j];
  }
  // the per-row vectors go through LDS as well (each lane reads back its own values): every
  // register a global load targets is then consumed here, at the top of the loop, so nothing
  // loop-carried forces an early s_waitcnt on the next tile's loads.
  float* sv = lds + NMAT * C::kTileFloats;
#pragma unroll
  for (int k = 0; k < NVEC; ++k)
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) sv[k * C::kTileRows + lane + 64 * i] = pf.s[k][i];
}

template <int R, int NMAT, int NVEC>
constexpr int sweep_lds_floats() { return NMAT * Cfg<R>::kTileFloats + NVEC * Cfg<R>::kTileRows; }

// ---- strided operands (round 4): an [N, R] matrix given as a column VIEW of a wider one -- element (row, c) at M[row * ld + c] --
// for the wide-rank paths (uvd_wide.py / splu_wide.py work on column chunks of U, V, L2, U2' without copying them).  Same tiles,
// same LDS image; only the global side differs: a lane's vectors are Cfg<R>::kVec floats wide (they never straddle a row), row and
// column of a vector come from its flat index by a division by the compile-time R.  The host checks that ld and the view's base
// are multiples of kVec floats.
template <int R, int NMAT, int NVEC>
struct PrefetchS {
  static constexpr int kV = Cfg<R>::kVec;
  static constexpr int kItems = Cfg<R>::kTileFloats / (64 * kV);
  typename VecT<kV>::type m[NMAT][kItems];
  float s[NVEC > 0 ? NVEC : 1][Cfg<R>::kRowsPerLane];
};
struct RowStrides { long ld[2]; };       // per operand (sweeps with strided operands have at most two)

template <int R, int NMAT, int NVEC>
__device__ __forceinline__ void issue_tile_s(PrefetchS<R, NMAT, NVEC>& pf, const float* const (&mats)[NMAT], const RowStrides& rs,
                                             const float* const (&vecs)[NVEC > 0 ? NVEC : 1], long tile, int lane) {
  using C = Cfg<R>;
  using P = PrefetchS<R, NMAT, NVEC>;
  typedef typename VecT<P::kV>::type VT;
  const long row0 = tile * C::kTileRows;
#pragma unroll
  for (int m = 0; m < NMAT; ++m) {
#pragma unroll
    for (int j = 0; j < P::kItems; ++j) {
      const int idx = (lane + 64 * j) * P::kV, row = idx / R, col = idx - row * R;
      pf.m[m][j] = *reinterpret_cast<const VT*>(mats[m] + (row0 + row) * rs.ld[m] + col);
    }
  }
#pragma unroll
  for (int k = 0; k < NVEC; ++k) {
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) pf.s[k][i] = vecs[k][row0 + lane + 64 * i];
  }
}
template <int R, int NMAT, int NVEC>
__device__ __forceinline__ void commit_tile_s(const PrefetchS<R, NMAT, NVEC>& pf, float* lds, int lane) {
  using C = Cfg<R>;
  using P = PrefetchS<R, NMAT, NVEC>;
  typedef typename VecT<P::kV>::type VT;
#pragma unroll
  for (int m = 0; m < NMAT; ++m) {
    VT* dst = reinterpret_cast<VT*>(lds + m * C::kTileFloats);
#pragma unroll
    for (int j = 0; j < P::kItems; ++j) dst[lane + 64 * j] = pf.m[m][j];
  }
  float* sv = lds + NMAT * C::kTileFloats;
#pragma unroll
  for (int k = 0; k < NVEC; ++k)
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) sv[k * C::kTileRows + lane + 64 * i] = pf.s[k][i];
}
// (tail tile / write-back helpers of the strided form: flat tile index -> address)
template <int R>
__device__ __forceinline__ long strided_off(int idx, long ld) { const int row = idx / R; return (long)row * ld + (idx - row * R); }

template <int R>
__device__ __forceinline__ void read_row(const float* tile, int row, float (&x)[R]) {
  const float* p = tile + row * R;
  if constexpr (R % 4 == 0) {
#pragma unroll
    for (int c = 0; c < R / 4; ++c) {
      const f32x4 q = reinterpret_cast<const f32x4*>(p)[c];
      x[4 * c + 0] = q[0]; x[4 * c + 1] = q[1]; x[4 * c + 2] = q[2]; x[4 * c + 3] = q[3];
    }
  } else if constexpr (R % 2 == 0) {
#pragma unroll
    for (int c = 0; c < R / 2; ++c) {
      const f32x2 q = reinterpret_cast<const f32x2*>(p)[c];
      x[2 * c + 0] = q[0]; x[2 * c + 1] = q[1];
    }
  } else {
#pragma unroll
    for (int c = 0; c < R; ++c) x[c] = p[c];
  }
}

template <int R>
__device__ __forceinline__ void write_row(float* tile, int row, const float (&x)[R]) {
  float* p = tile + row * R;
  if constexpr (R % 4 == 0) {
#pragma unroll
    for (int c = 0; c < R / 4; ++c)
      reinterpret_cast<f32x4*>(p)[c] = f32x4{x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]};
  } else if constexpr (R % 2 == 0) {
#pragma unroll
    for (int c = 0; c < R / 2; ++c) reinterpret_cast<f32x2*>(p)[c] = f32x2{x[2 * c], x[2 * c + 1]};
  } else {
#pragma unroll
    for (int c = 0; c < R; ++c) p[c] = x[c];
  }
}

// No per-tile work (default hook of sweep_rows).
struct NoTileHook {
  static constexpr unsigned kStoreVecMask = 0;    // which per-row scalars s[k] the body's values are written back to LDS
  static constexpr bool kActive = false;
  __device__ __forceinline__ void operator()(const float*) const {}
};

// Generic row sweep.  mats: NMAT [N,R] inputs; vecs: NVEC [N] inputs;
// WB >= 0: operand WB is modified by the body and streamed back to `mat_out`.
// body(row, valid, x[NMAT][R], s[NVEC]).  Rows past N (tail tile only) arrive
// zero-filled with valid == false.
// hook (optional): called once per tile after every row of the tile has been processed, with the wave's LDS tile
// (operands at m * kTileFloats -- operand WB holds the NEW rows --, per-row scalars at NMAT * kTileFloats +
// k * kTileRows; the slots in Hook::kStoreVecMask hold what the body left in s[k]).  Column reductions that would
// cost one accumulator register per column per lane run there on the matrix core instead (ColSum2).
template <int R, int NMAT, int NVEC, int WB, bool NT, bool STR = false, class Body, class Hook = NoTileHook>
__device__ __forceinline__ void sweep_rows(const float* const (&mats)[NMAT],
                                           const float* const (&vecs)[NVEC > 0 ? NVEC : 1],
                                           float* mat_out, long N, float* lds, Body&& body, Hook&& hook = Hook(),
                                           const RowStrides rs = RowStrides{{R, R}}) {
  using C = Cfg<R>;
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * kWavesPerBlock + wave_in_block();
  const long nw = (long)gridDim.x * kWavesPerBlock;
  const long nfull = N / C::kTileRows;

  std::conditional_t<STR, PrefetchS<R, NMAT, NVEC>, Prefetch<R, NMAT, NVEC>> pf;
  auto issue = [&](long t) {
    if constexpr (STR) issue_tile_s<R, NMAT, NVEC>(pf, mats, rs, vecs, t, lane);
    else issue_tile<R, NMAT, NVEC, NT>(pf, mats, vecs, t, lane);
  };
  // tile order of one wave: chunks of kTileChunk consecutive tiles, chunks dealt round-robin over the waves
  constexpr int CH = kTileChunk;
  long it = 0;
  auto tile_at = [&](long j) { return (CH == 1) ? gw + j * nw : ((j / CH) * nw + gw) * CH + (j % CH); };
  long tile = tile_at(0);
  // The prefetch is unconditional (the last iteration re-reads its own tile) so that the
  // buffers stay in registers and the loads stay in flight across the compute phase.
  if (tile < nfull) issue(tile);
  while (tile < nfull) {
    if constexpr (STR) commit_tile_s<R, NMAT, NVEC>(pf, lds, lane);
    else commit_tile<R, NMAT, NVEC>(pf, lds, lane);
    const long next = tile_at(++it);
    issue((next < nfull) ? next : tile);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) {
      const int rit = lane + 64 * i;
      float x[NMAT][R];
#pragma unroll
      for (int m = 0; m < NMAT; ++m) read_row<R>(lds + m * C::kTileFloats, rit, x[m]);
      float s[NVEC > 0 ? NVEC : 1];
#pragma unroll
      for (int k = 0; k < NVEC; ++k) s[k] = lds[NMAT * C::kTileFloats + k * C::kTileRows + rit];
      body(tile * C::kTileRows + rit, true, x, s);
      if constexpr (WB >= 0) write_row<R>(lds + WB * C::kTileFloats, rit, x[WB]);
#pragma unroll
      for (int k = 0; k < NVEC; ++k)
        if ((std::remove_reference_t<Hook>::kStoreVecMask >> k) & 1u)      // compile-time after unrolling
          lds[NMAT * C::kTileFloats + k * C::kTileRows + rit] = s[k];
    }
    if constexpr (std::remove_reference_t<Hook>::kActive) {
      __builtin_amdgcn_wave_barrier();
      hook(lds);
    }
    if constexpr (WB >= 0 && STR) {
      __builtin_amdgcn_wave_barrier();
      using P = PrefetchS<R, NMAT, NVEC>;
      typedef typename VecT<P::kV>::type VT;
      const VT* src = reinterpret_cast<const VT*>(lds + WB * C::kTileFloats);
      float* dst = mat_out + tile * C::kTileRows * rs.ld[WB];
#pragma unroll
      for (int j = 0; j < P::kItems; ++j)
        *reinterpret_cast<VT*>(dst + strided_off<R>((lane + 64 * j) * P::kV, rs.ld[WB])) = src[lane + 64 * j];
      __builtin_amdgcn_wave_barrier();
    } else if constexpr (WB >= 0) {
      __builtin_amdgcn_wave_barrier();
      const typename C::LoadT* src = reinterpret_cast<const typename C::LoadT*>(lds + WB * C::kTileFloats);
      typename C::LoadT* dst = reinterpret_cast<typename C::LoadT*>(mat_out + tile * C::kTileRows * R);
#pragma unroll
      for (int j = 0; j < C::kLoadsPerLane; ++j) stream_store<NT>(dst + lane + 64 * j, src[lane + 64 * j]);
      __builtin_amdgcn_wave_barrier();
    }
    tile = next;
  }

  // tail tile (N % kTileRows rows): guarded scalar staging, owned by one wave
  const long tail_rows = N - nfull * C::kTileRows;
  if (tail_rows > 0 && ((nfull / CH) % nw) == gw) {
    const int lane = lane_from_exec();               // (all 64 lanes are here: the condition is wave-uniform)
    const long row0 = nfull * C::kTileRows;

    const long tail_floats = tail_rows * R;
#pragma unroll
    for (int m = 0; m < NMAT; ++m) {
      const float* src = mats[m] + row0 * (STR ? rs.ld[m] : (long)R);
      float* dst = lds + m * C::kTileFloats;
      for (int idx = lane; idx < C::kTileFloats; idx += 64)
        dst[idx] = (idx < tail_floats) ? src[STR ? strided_off<R>(idx, rs.ld[m]) : (long)idx] : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) {
      const int rit = lane + 64 * i;
      const bool valid = rit < tail_rows;
      float x[NMAT][R];
#pragma unroll
      for (int m = 0; m < NMAT; ++m) read_row<R>(lds + m * C::kTileFloats, rit, x[m]);
      float s[NVEC > 0 ? NVEC : 1];
#pragma unroll
      for (int k = 0; k < NVEC; ++k) s[k] = valid ? vecs[k][row0 + rit] : 0.0f;
      body(row0 + rit, valid, x, s);
      if constexpr (WB >= 0) write_row<R>(lds + WB * C::kTileFloats, rit, x[WB]);
#pragma unroll
      for (int k = 0; k < NVEC; ++k)
        if ((std::remove_reference_t<Hook>::kStoreVecMask >> k) & 1u)      // compile-time after unrolling
          lds[NMAT * C::kTileFloats + k * C::kTileRows + rit] = s[k];
    }
    if constexpr (std::remove_reference_t<Hook>::kActive) {
      __builtin_amdgcn_wave_barrier();
      hook(lds);
    }
    if constexpr (WB >= 0) {
      __builtin_amdgcn_wave_barrier();
      const float* src = lds + WB * C::kTileFloats;
      float* dst = mat_out + row0 * (STR ? rs.ld[WB] : (long)R);
      for (int idx = lane; idx < tail_floats; idx += 64) dst[STR ? strided_off<R>(idx, rs.ld[WB]) : (long)idx] = src[idx];
    }
  }
}

// ------------------------------------------------------ block reductions ---
// Sum acc[L] over the block (fixed order) and store the block's partials TRANSPOSED:
// part[c * G + block], so that the reduce kernel reads each element's G partials coalesced.
template <int L>
__device__ __forceinline__ void block_sum_store(float (&acc)[L], float* red /* [waves][L] */, float* part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < L; ++c) {
    float v = acc[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[w * L + c] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < L; c += kThreads) {
    float s = red[c];
#pragma unroll
    for (int k = 1; k < kWavesPerBlock; ++k) s += red[k * L + c];
    part[(long)c * gridDim.x + blockIdx.x] = s;
  }
}

// NaN-propagating (nanmax.h).  SIGNED = false: maxima of |x| (one integer max per step); true: any values.
template <bool SIGNED = false>
__device__ __forceinline__ void block_max_store(float v, float* red /* [waves] */, float* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = SIGNED ? nmaxf(v, __shfl_down(v, off, 64)) : amaxf(v, __shfl_down(v, off, 64));
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = red[0];
#pragma unroll
    for (int k = 1; k < kWavesPerBlock; ++k) s = SIGNED ? nmaxf(s, red[k]) : amaxf(s, red[k]);
    *out = s;
  }
}

template <int R>
__device__ __forceinline__ float dot_row(const float (&x)[R], const float* __restrict__ c) {
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < R; ++k) s = fmaf(x[k], c[k], s);
  return s;
}

// ------------------------------------------------------------- kernels -----
// s = M' (a .* b)  (NVEC == 2) or M' a (NVEC == 1): apply sweep 1 (psgd.py:544
// inner matmul with x = d*g, :625) and the first half of IpUVtmatvec.
template <int R, int NVEC, bool NT>
__global__ __launch_bounds__(kThreads) void k_colreduce(const float* M, const float* a, const float* b,
                                                        long N, float* part) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 1, NVEC>()];
  __shared__ float red[kWavesPerBlock * R];
  float acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0f;
  const float* const mats[1] = {M};
  auto body = [&](long, bool, float (&x)[1][R], float (&s)[NVEC]) {
    float t = s[0];
    if constexpr (NVEC > 1) t *= s[1];
#pragma unroll
    for (int c = 0; c < R; ++c) acc[c] = fmaf(x[0][c], t, acc[c]);
  };
  if constexpr (NVEC == 2) {
    const float* const vecs[2] = {a, b};
    sweep_rows<R, 1, 2, -1, NT>(mats, vecs, nullptr, N, lds[wave_in_block()], body);
  } else {
    const float* const vecs[1] = {a};
    sweep_rows<R, 1, 1, -1, NT>(mats, vecs, nullptr, N, lds[wave_in_block()], body);
  }
  block_sum_store<R>(acc, red, part);
}

// apply sweep 2: g1 = t + U s1 (stored), s2 = U' g1, t = d .* g   (psgd.py:625 -> :626 -> :544)
// g1 is row-local, so the pass that reduces U'g1 also writes it out (4 B/row); sweep 3 then needs
// V only: U is read once per apply, not twice.
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_apply_s2(const float* U, const float* d, const float* g, float* g1out,
                                                       long N, const float* __restrict__ coef, float* part) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 1, 2>()];
  __shared__ float red[kWavesPerBlock * R];
  float acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0f;
  const float* const mats[1] = {U};
  const float* const vecs[2] = {d, g};
  sweep_rows<R, 1, 2, -1, NT>(mats, vecs, nullptr, N, lds[wave_in_block()],
                          [&](long row, bool valid, float (&x)[1][R], float (&s)[2]) {
                            const float t = s[0] * s[1];
                            const float g1 = t + dot_row<R>(x[0], coef);
                            if (valid) stream_store<NT>(g1out + row, g1);
#pragma unroll
                            for (int c = 0; c < R; ++c) acc[c] = fmaf(x[0][c], g1, acc[c]);
                          });
  block_sum_store<R>(acc, red, part);
}

// apply sweep 3: out = d .* (g1 + V s2), in place on the buffer that holds g1   (psgd.py:626)
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_apply_s3(const float* V, const float* d, float* out, long N,
                                                       const float* __restrict__ coef) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 1, 2>()];
  const float* const mats[1] = {V};
  const float* const vecs[2] = {d, out};
  sweep_rows<R, 1, 2, -1, NT>(mats, vecs, nullptr, N, lds[wave_in_block()],
                          [&](long row, bool valid, float (&x)[1][R], float (&s)[2]) {
                            const float o = s[0] * (s[1] + dot_row<R>(x[0], coef + R));
                            if (valid) stream_store<NT>(out + row, o);
                          });
}

// out = x + M s   (second half of IpUVtmatvec, psgd.py:544)
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_rowdot_axpy(const float* M, const float* xin, float* out, long N,
                                                          const float* __restrict__ coef) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 1, 1>()];
  const float* const mats[1] = {M};
  const float* const vecs[1] = {xin};
  sweep_rows<R, 1, 1, -1, NT>(mats, vecs, nullptr, N, lds[wave_in_block()],
                          [&](long row, bool valid, float (&x)[1][R], float (&s)[1]) {
                            const float o = s[0] + dot_row<R>(x[0], coef);
                            if (valid) stream_store<NT>(out + row, o);
                          });
}

// 3-way bf16 split of 8 fp32 values: x = h + m + l (bf16 keeps fp32's exponent range), so that
// x*y ~ h*h' + h*m' + m*h' + h*l' + l*h' + m*m' can run on the bf16 matrix cores (16x the fp32 MFMA rate).
#ifndef PSGD_GRAM_BF16X3
#define PSGD_GRAM_BF16X3 1
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// The split truncates (h = top 16 bits of x, r1 = x - h exactly, m = top 16 bits of r1, l = r1 - m, a bf16 exactly), so
// x = h + m + l holds exactly and the work is 4 full-rate VALU ops per element plus a v_perm_b32 per pair and plane
// (three quarter-rate v_cvt_pk_bf16_f32 per element in the first version).
__device__ __forceinline__ void split3_bf16(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
  const unsigned top = 0xFFFF0000u;
  u32x4s ph, pm, pl;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    // (pairs as two-element vectors: the two subtractions of a pair are one v_pk_add_f32 each)
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
    const f32x2s xx = {x[j], x[j + 1]};
    const f32x2s hh = __builtin_bit_cast(f32x2s, __builtin_bit_cast(u32x2s, xx) & top);
    const f32x2s rr = xx - hh;
    const f32x2s mm = __builtin_bit_cast(f32x2s, __builtin_bit_cast(u32x2s, rr) & top);
    const f32x2s ss = rr - mm;
    const float x0 = xx[0], x1 = xx[1], r0 = rr[0], r1 = rr[1], s0 = ss[0], s1 = ss[1];
    ph[j >> 1] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);   // (x0 >> 16) | (x1 & top)
    pm[j >> 1] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
    pl[j >> 1] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  h = __builtin_bit_cast(bf16x8, ph);
  m = __builtin_bit_cast(bf16x8, pm);
  l = __builtin_bit_cast(bf16x8, pl);
}

// update sweep 1: Gram of W = [U | V | t | w], t = d.*h, w = v./d, on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chains).  The Gram
// holds every inner product psgd.py:569-615 needs: V'U (:574), U'U, V'V, V't,
// U't, U'w, V'w, t't, w'w, t'w.  Only block pairs bi <= bj are computed.
// Chains are 1 tile long; tile results are accumulated in fp64.
template <int R, bool NT, bool STR = false>
__global__ __launch_bounds__(kThreads) void k_update_gram(const float* U, const float* V, const float* d,
                                                          const float* v, const float* h, long N, double* part,
                                                          RowStrides rs = RowStrides{{R, R}}) {
  using C = Cfg<R>;
  using GC = GramCfg<R>;
  constexpr int kSv = 2 * C::kTileFloats;             // staged d, v, h (3 x kTileRows) from commit_tile
  constexpr int kTw = kSv + 3 * C::kTileRows;         // t,w interleaved [kTileRows][2]
  constexpr int kZero = kTw + 2 * C::kTileRows;       // 4 zero floats
  constexpr int kWaveFloats = kZero + 4;
  constexpr int kTileBytes = kWavesPerBlock * kWaveFloats * 4;
  constexpr int kScratchBytes = 2 * GC::kLen * 8;
  constexpr int kLdsBytes = kTileBytes > kScratchBytes ? kTileBytes : kScratchBytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* lds = reinterpret_cast<float*>(smem) + w * kWaveFloats;
  if (lane < 4) lds[kZero + lane] = 0.0f;

  int obase[GC::kBlocks], obase8[GC::kBlocks], ostride[GC::kBlocks];
#pragma unroll
  for (int b = 0; b < GC::kBlocks; ++b) {
    const int col = 16 * b + (lane & 15);
    if (col < R) { obase[b] = col; ostride[b] = R; }
    else if (col < 2 * R) { obase[b] = C::kTileFloats + (col - R); ostride[b] = R; }
    else if (col < 2 * R + 2) { obase[b] = kTw + (col - 2 * R); ostride[b] = 2; }
    else { obase[b] = kZero; ostride[b] = 0; }
    obase8[b] = obase[b] + 8 * (lane >> 4) * ostride[b];    // bf16 fragments: rows 8(l>>4) + j
    obase[b] += (lane >> 4) * ostride[b];                    // fp32 MFMA: row (l>>4) of each group of 4
  }

  double acc64[GC::kPairs][4];
#pragma unroll
  for (int p = 0; p < GC::kPairs; ++p)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc64[p][e] = 0.0;

  const long gw = (long)blockIdx.x * kWavesPerBlock + w;
  const long nw = (long)gridDim.x * kWavesPerBlock;
  const long nfull = N / C::kTileRows;
  const long tail_rows = N - nfull * C::kTileRows;
  const float* const mats[2] = {U, V};
  const float* const vecs[3] = {d, v, h};

  // fp32 MFMA accumulators persist across kFlushTiles tiles (chains of <= 256 rows), then fold into fp64
  constexpr int kFlushTiles = (256 / C::kTileRows) > 0 ? (256 / C::kTileRows) : 1;
  f32x4 acc[GC::kPairs];
#pragma unroll
  for (int p = 0; p < GC::kPairs; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  int since_flush = 0;
  auto flush = [&]() {
#pragma unroll
    for (int p = 0; p < GC::kPairs; ++p) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc64[p][e] += (double)acc[p][e];
      acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    since_flush = 0;
  };
  auto gram_tile = [&]() {
#if PSGD_GRAM_BF16X3
    // 32-row chunks on v_mfma_f32_16x16x32_bf16: lane l holds, for each 16-column block, the 8 rows
    // 32c + 8(l>>4) + j of column (l & 15) -- the same fragment serves as A (W' block bi) and as B (block bj).
    // The fp32 formulation below needs 96 fp32 MFMAs (3072 SIMD cycles) per 64-row tile and pinned this
    // sweep at the MFMA rate; this one needs 72 bf16 MFMAs (1152 cycles) and is HBM-bound.
#pragma unroll 1
    for (int c = 0; c < C::kTileRows / 32; ++c) {
      bf16x8 fh[GC::kBlocks], fm[GC::kBlocks], fl[GC::kBlocks];
#pragma unroll
      for (int b = 0; b < GC::kBlocks; ++b) {
        float x[8];
        const int base = obase8[b] + 32 * c * ostride[b];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = lds[base + j * ostride[b]];
        split3_bf16(x, fh[b], fm[b], fl[b]);
      }
      int p = 0;
#pragma unroll
      for (int bi = 0; bi < GC::kBlocks; ++bi)
#pragma unroll
        for (int bj = bi; bj < GC::kBlocks; ++bj) {
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[bi], fm[bj], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[bi], fl[bj], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[bi], fh[bj], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[bi], fm[bj], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[bi], fh[bj], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[bi], fh[bj], acc[p], 0, 0, 0);
          ++p;
        }
    }
#else
#pragma unroll 4
    for (int m = 0; m < C::kTileRows / 4; ++m) {
      float a[GC::kBlocks];
#pragma unroll
      for (int b = 0; b < GC::kBlocks; ++b) a[b] = lds[obase[b] + 4 * m * ostride[b]];
      int p = 0;
#pragma unroll
      for (int bi = 0; bi < GC::kBlocks; ++bi)
#pragma unroll
        for (int bj = bi; bj < GC::kBlocks; ++bj) {
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[bi], a[bj], acc[p], 0, 0, 0);
          ++p;
        }
    }
#endif
    if (++since_flush == kFlushTiles) flush();
  };

  std::conditional_t<STR, PrefetchS<R, 2, 3>, Prefetch<R, 2, 3>> pf;
  auto issue = [&](long t) {
    if constexpr (STR) issue_tile_s<R, 2, 3>(pf, mats, rs, vecs, t, lane);
    else issue_tile<R, 2, 3, NT>(pf, mats, vecs, t, lane);
  };
  long tile = gw;
  if (tile < nfull) issue(tile);
  while (tile < nfull) {
    if constexpr (STR) commit_tile_s<R, 2, 3>(pf, lds, lane);
    else commit_tile<R, 2, 3>(pf, lds, lane);
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) {
      const float dd = lds[kSv + 0 * C::kTileRows + lane + 64 * i];
      const float vv = lds[kSv + 1 * C::kTileRows + lane + 64 * i];
      const float hh = lds[kSv + 2 * C::kTileRows + lane + 64 * i];
      reinterpret_cast<f32x2*>(lds + kTw)[lane + 64 * i] = f32x2{dd * hh, vv / dd};
    }
    const long next = tile + nw;
    issue((next < nfull) ? next : tile);
    __builtin_amdgcn_wave_barrier();
    gram_tile();
    __builtin_amdgcn_wave_barrier();
    tile = next;
  }
  if (tail_rows > 0 && (nfull % nw) == gw) {
    const long row0 = nfull * C::kTileRows;
    const long tail_floats = tail_rows * R;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float* src = mats[m] + row0 * (STR ? rs.ld[m] : (long)R);
      float* dst = lds + m * C::kTileFloats;
      for (int idx = lane; idx < C::kTileFloats; idx += 64)
        dst[idx] = (idx < tail_floats) ? src[STR ? strided_off<R>(idx, rs.ld[m]) : (long)idx] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) {
      const int rit = lane + 64 * i;
      float t = 0.0f, ww = 0.0f;
      if (rit < tail_rows) {
        const float dd = d[row0 + rit];
        t = dd * h[row0 + rit];
        ww = v[row0 + rit] / dd;
      }
      reinterpret_cast<f32x2*>(lds + kTw)[rit] = f32x2{t, ww};
    }
    __builtin_amdgcn_wave_barrier();
    gram_tile();
  }

  flush();
  // block reduction in fp64 through LDS (tile buffers are dead now), fixed order:
  // (w2,w3) -> scratch ; w0 += s0, w1 += s1 ; w1 -> scratch ; w0 += s0 ; w0 stores
  __syncthreads();
  double* scratch = reinterpret_cast<double*>(smem);
  if (w >= 2) {
#pragma unroll
    for (int p = 0; p < GC::kPairs; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) scratch[(w - 2) * GC::kLen + p * 256 + e * 64 + lane] = acc64[p][e];
  }
  __syncthreads();
  if (w < 2) {
#pragma unroll
    for (int p = 0; p < GC::kPairs; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc64[p][e] += scratch[w * GC::kLen + p * 256 + e * 64 + lane];
  }
  __syncthreads();
  if (w == 1) {
#pragma unroll
    for (int p = 0; p < GC::kPairs; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) scratch[p * 256 + e * 64 + lane] = acc64[p][e];
  }
  __syncthreads();
  if (w == 0) {
    double* dst = part + (long)blockIdx.x * GC::kLen;
#pragma unroll
    for (int p = 0; p < GC::kPairs; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[p * 256 + e * 64 + lane] = acc64[p][e] + scratch[p * 256 + e * 64 + lane];
  }
}

// coefficient block consumed by update sweep 2 (written by k_update_coef)
//   [0,R) s1 = V't      [R,2R) s2 = U'Qh      [2R,3R) x1      [3R,4R) x2
//   [4R,5R) c1          [5R,6R) c2            [6R] mu  [6R+1] norm (debug)
template <int R>
struct UpdCoef {
  static constexpr int kS1 = 0, kS2 = R, kX1 = 2 * R, kX2 = 3 * R, kC1 = 4 * R, kC2 = 5 * R, kMu = 6 * R;
};

// Column sums W' [x0 .. x_{NX-1}] over all the rows a wave sweeps, W = the first NW staged operands side by side
// ([U | V] or one matrix), on the matrix core: one v_mfma_f32_4x4x1_16b_f32 per row.  That instruction is 16 independent
// 4x4 outer products; lane l supplies A = a[l] and B = b[l], and D[e][lane 4q + j] += a[lane 4q + e] * b[lane 4q + j]
// (layout found with tools/micro/mfma4x4_probe.hip).  With a[l] = W[row][l] (column l of W, l < NW * R <= 64: one
// conflict-free LDS read per lane) and b[l] = x_{l & 3}[row], lane 4q + j ends up with the sums of columns
// 4q .. 4q + 3 against x_j in its four accumulator registers: 4 VGPRs instead of NX * NW * R per-lane accumulators, on
// a pipe the sweep does not otherwise use.  The x_j are the per-row scalars in LDS slots SLOT0 .. SLOT0 + NX - 1 (as
// the body left them: kStoreVecMask).  fp32 chains are <= 256 rows long, then folded into fp64.
template <int R, int NMAT, int NW, int NX, int SLOT0, int OP = 0>       // OP: the first of the NW staged operands that form W
struct ColSum {
  using C = Cfg<R>;
  static_assert(NW * R <= 64 && NX <= 4 && OP + NW <= NMAT, "one lane per column of W, one 4x4 block column per vector");
  static constexpr unsigned kStoreVecMask = ((1u << NX) - 1u) << SLOT0;
  static constexpr bool kActive = true;
  static constexpr int kFlushTiles = (256 / C::kTileRows) > 0 ? (256 / C::kTileRows) : 1;
  f32x4 acc0, acc1;
  double acc64[4];
  int aoff, boff, since;
  float amask, bmask;
  __device__ __forceinline__ ColSum() {
    const int l = threadIdx.x & 63;
    acc0 = acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) acc64[e] = 0.0;
    since = 0;
    aoff = (l < NW * R) ? (OP + l / R) * C::kTileFloats + (l % R) : 0;
    amask = (l < NW * R) ? 1.0f : 0.0f;
    const int j = l & 3;
    boff = NMAT * C::kTileFloats + (SLOT0 + (j < NX ? j : 0)) * C::kTileRows;
    bmask = (j < NX) ? 1.0f : 0.0f;
  }
  __device__ __forceinline__ void flush() {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc64[e] += (double)acc0[e] + (double)acc1[e];
    acc0 = acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    since = 0;
  }
  __device__ __forceinline__ void operator()(const float* lds) {
#pragma unroll 8
    for (int row = 0; row < C::kTileRows; row += 2) {
      const float a0 = lds[aoff + row * R] * amask, b0 = lds[boff + row] * bmask;
      const float a1 = lds[aoff + (row + 1) * R] * amask, b1 = lds[boff + row + 1] * bmask;
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, acc1, 0, 0, 0);
    }
    if (++since == kFlushTiles) flush();
  }
  // Block total (fixed order over the 4 waves) -> part[(j * NW * R + c) * G + block], j = vector, c = column of W:
  // transposed fp64 partials, so that a reduction reads an element's G partials contiguously.
  // (col0, ncols: this W as columns col0 .. of a wider one with ncols columns -- ColSumPair below)
  __device__ __forceinline__ void block_store(double* red /* [waves][4][64] */, double* part, int col0 = 0, int ncols = NW * R) {
    flush();
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(w * 4 + e) * 64 + l] = acc64[e];
    __syncthreads();
    if (w == 0 && (l & 3) < NX) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * (l >> 2) + e;
        if (c < NW * R) {
          const double t = ((red[(0 * 4 + e) * 64 + l] + red[(1 * 4 + e) * 64 + l]) + red[(2 * 4 + e) * 64 + l]) +
                           red[(3 * 4 + e) * 64 + l];
          part[(long)((l & 3) * ncols + col0 + c) * gridDim.x + blockIdx.x] = t;
        }
      }
    }
  }
};
template <int R, int NMAT, int SLOT0, int SLOT1>
using ColSum2 = ColSum<R, NMAT, 2, 2, SLOT0>;      // [U | V]' [x0 x1], x0, x1 in consecutive slots (sweep 2 of the fused step)
// The same sums for ranks 33 .. 64, where [U | V] has more than 64 columns: one ColSum per operand (U' [x0 x1], then V' [x0 x1]),
// stored as the columns [0, R) and [R, 2R) of the same partial layout.
template <int R, int SLOT0>
struct ColSumPair {
  ColSum<R, 2, 1, 2, SLOT0, 0> u;
  ColSum<R, 2, 1, 2, SLOT0, 1> v;
  static constexpr unsigned kStoreVecMask = ColSum<R, 2, 1, 2, SLOT0, 0>::kStoreVecMask;
  static constexpr bool kActive = true;
  // one pass over the tile's rows for both operands: the two per-row scalars are read once, four accumulator chains are in flight
  __device__ __forceinline__ void operator()(const float* lds) {
    using C = Cfg<R>;
#pragma unroll 8
    for (int row = 0; row < C::kTileRows; row += 2) {
      const float b0 = lds[u.boff + row] * u.bmask, b1 = lds[u.boff + row + 1] * u.bmask;
      const float a0 = lds[u.aoff + row * R] * u.amask, a1 = lds[u.aoff + (row + 1) * R] * u.amask;
      const float c0 = lds[v.aoff + row * R] * v.amask, c1 = lds[v.aoff + (row + 1) * R] * v.amask;
      u.acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, u.acc0, 0, 0, 0);
      v.acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(c0, b0, v.acc0, 0, 0, 0);
      u.acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, u.acc1, 0, 0, 0);
      v.acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(c1, b1, v.acc1, 0, 0, 0);
    }
    if (++u.since == ColSum<R, 2, 1, 2, SLOT0, 0>::kFlushTiles) { u.flush(); v.flush(); }
  }
  __device__ __forceinline__ void block_store(double* red, double* part) {
    u.block_store(red, part, 0, 2 * R);
    v.block_store(red, part, R, 2 * R);
  }
};

// IpUVtmatvec on up to four columns at once (psgd.py:540-544 with a matrix x): S = V' [x0 .. x3] in ONE sweep of V
// (the sums on the matrix core) ...
template <int R, bool NT, bool STR = false>
__global__ __launch_bounds__(kThreads) void k_colreduce4(const float* M, const float* x0, const float* x1, const float* x2,
                                                         const float* x3, long N, double* part, long ld = R) {
  constexpr int kLdsSweep = kWavesPerBlock * sweep_lds_floats<R, 1, 4>() * 4;
  constexpr int kLdsRed = kWavesPerBlock * 4 * 64 * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsSweep > kLdsRed ? kLdsSweep : kLdsRed];
  float* lds = reinterpret_cast<float*>(smem) + wave_in_block() * sweep_lds_floats<R, 1, 4>();
  const float* const mats[1] = {M};
  const float* const vecs[4] = {x0, x1, x2, x3};
  ColSum<R, 1, 1, 4, 0> cs;
  sweep_rows<R, 1, 4, -1, NT, STR>(mats, vecs, nullptr, N, lds, [&](long, bool, float (&)[1][R], float (&)[4]) {}, cs,
                                   RowStrides{{ld, ld}});
  cs.block_store(reinterpret_cast<double*>(smem), part);
}

// ... and out_j = x_j + M S_j for the four columns in ONE sweep of M (coef = S, [4][R]).
template <int R, bool NT, bool STR = false>
__global__ __launch_bounds__(kThreads) void k_rowdot_axpy4(const float* M, const float* x0, const float* x1,
                                                           const float* x2, const float* x3, float* o0, float* o1,
                                                           float* o2, float* o3, int ncols, long N,
                                                           const float* __restrict__ coef, long ld = R) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 1, 4>()];
  const float* const mats[1] = {M};
  const float* const vecs[4] = {x0, x1, x2, x3};
  sweep_rows<R, 1, 4, -1, NT, STR>(mats, vecs, nullptr, N, lds[wave_in_block()],
                                   [&](long row, bool valid, float (&x)[1][R], float (&s)[4]) {
                                     if (!valid) return;
                                     stream_store<NT>(o0 + row, s[0] + dot_row<R>(x[0], coef));
                                     if (ncols > 1) stream_store<NT>(o1 + row, s[1] + dot_row<R>(x[0], coef + R));
                                     if (ncols > 2) stream_store<NT>(o2 + row, s[2] + dot_row<R>(x[0], coef + 2 * R));
                                     if (ncols > 3) stream_store<NT>(o3 + row, s[3] + dot_row<R>(x[0], coef + 3 * R));
                                   }, NoTileHook(), RowStrides{{ld, ld}});
}

// precond_grad_UVd_math on up to four columns of a MATRIX g at once (psgd.py:619-627, docstring :623 "either matrices or
// column vectors": d*g broadcasts d over the columns and both IpUVtmatvec calls take [N, k]).  The three sweeps of the
// single-column apply, each over four columns, so U and V are read 1.5 times per group of four columns instead of per column:
//   sweep 1  S1 = V'(d .* G)                          (matrix core; the body leaves d .* g_j in the LDS slots ColSum reads)
//   sweep 2  G1 = d .* G + U S1 (stored to out_j),  S2 = U'G1
//   sweep 3  out_j = d .* (G1_j + V S2_j)             (in place on out_j)
// coef = [4][R].  Columns past ncols are duplicates of column 0 on the way in (the launcher's x[j], o[j]) and never stored.
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_apply4_s1(const float* V, const float* d, const float* x0, const float* x1,
                                                        const float* x2, const float* x3, long N, double* part) {
  constexpr int kLdsSweep = kWavesPerBlock * sweep_lds_floats<R, 1, 5>() * 4;
  constexpr int kLdsRed = kWavesPerBlock * 4 * 64 * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsSweep > kLdsRed ? kLdsSweep : kLdsRed];
  float* lds = reinterpret_cast<float*>(smem) + wave_in_block() * sweep_lds_floats<R, 1, 5>();
  const float* const mats[1] = {V};
  const float* const vecs[5] = {x0, x1, x2, x3, d};
  ColSum<R, 1, 1, 4, 0> cs;
  sweep_rows<R, 1, 5, -1, NT>(mats, vecs, nullptr, N, lds, [&](long, bool, float (&)[1][R], float (&s)[5]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] *= s[4];
  }, cs);
  cs.block_store(reinterpret_cast<double*>(smem), part);
}

template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_apply4_s2(const float* U, const float* d, const float* x0, const float* x1,
                                                        const float* x2, const float* x3, float* o0, float* o1, float* o2,
                                                        float* o3, int ncols, long N, const float* __restrict__ coef,
                                                        double* part) {
  constexpr int kLdsSweep = kWavesPerBlock * sweep_lds_floats<R, 1, 5>() * 4;
  constexpr int kLdsRed = kWavesPerBlock * 4 * 64 * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsSweep > kLdsRed ? kLdsSweep : kLdsRed];
  float* lds = reinterpret_cast<float*>(smem) + wave_in_block() * sweep_lds_floats<R, 1, 5>();
  const float* const mats[1] = {U};
  const float* const vecs[5] = {x0, x1, x2, x3, d};
  float* const outs[4] = {o0, o1, o2, o3};
  ColSum<R, 1, 1, 4, 0> cs;
  sweep_rows<R, 1, 5, -1, NT>(mats, vecs, nullptr, N, lds, [&](long row, bool valid, float (&x)[1][R], float (&s)[5]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g1 = s[4] * s[j] + dot_row<R>(x[0], coef + j * R);         // rows past N: all zeros
      if (valid && j < ncols) stream_store<NT>(outs[j] + row, g1);
      s[j] = g1;
    }
  }, cs);
  cs.block_store(reinterpret_cast<double*>(smem), part);
}

template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_apply4_s3(const float* V, const float* d, float* o0, float* o1, float* o2,
                                                        float* o3, int ncols, long N, const float* __restrict__ coef) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 1, 5>()];
  const float* const mats[1] = {V};
  const float* const vecs[5] = {o0, o1, o2, o3, d};
  float* const outs[4] = {o0, o1, o2, o3};
  sweep_rows<R, 1, 5, -1, NT>(mats, vecs, nullptr, N, lds[wave_in_block()],
                              [&](long row, bool valid, float (&x)[1][R], float (&s)[5]) {
                                if (!valid) return;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                  if (j < ncols) stream_store<NT>(outs[j] + row, s[4] * (s[j] + dot_row<R>(x[0], coef + j * R)));
                              });
}

// update sweep 2 (row-local; psgd.py:569-601 / :603-615 given the r-vectors):
//   a = t + U s1 (Qh)            b = w - V x1 (invQtv)
//   Ph = d (a + V s2)            invPv = (b - U x2) / d
//   nablaD = Ph h - v invPv      (stored; its max|.| reduced)
//   UPDATE_U: U <- U - mu (a c1 - b c2),  c1 = atV K, c2 = btV K
//   else    : V <- V - mu ((a + V c1) c1 - (b + V c2) c2),  c1 = atU, c2 = btU
// FUSE: UVd.step applies the preconditioner right after updating it (psgd.py:732 -> :748).  Both
// reductions of that apply can be had without another pass over U or V.  With dnew = d - mu_d d .* nablaD
// (psgd.py:584) and tg = d .* g, tn = d .* g .* nablaD:
//   s1' = Vnew'(dnew .* g)             = Vnew'tg - mu_d Vnew'tn
//   s2' = Unew'(dnew .* g + Unew s1')  = Unew'tg - mu_d Unew'tn + (Unew'Unew) s1'
// and Unew'Unew is the Gram block U'U of sweep 1 plus a rank-2 correction known from the r x r algebra (k_fused_post).
// So this sweep also reduces the four r-vectors [Unew | Vnew]' [tg tn] (ColSum2, on the matrix core), and the apply
// becomes ONE more sweep (k_uvd_final) instead of a d update and two sweeps.
template <int R, bool UPDATE_U, bool NT, bool FUSE>
__global__ __launch_bounds__(kThreads) void k_update_s2(float* U, float* V, const float* d, const float* v,
                                                        const float* h, const float* g, long N,
                                                        const float* __restrict__ coef, float* nabla, float* part_max,
                                                        double* part_pq) {
  using K = UpdCoef<R>;
  constexpr int NV = FUSE ? 4 : 3;
  constexpr int kLdsSweep = kWavesPerBlock * sweep_lds_floats<R, 2, NV>() * 4;
  constexpr int kLdsRed = FUSE ? kWavesPerBlock * 4 * 64 * 8 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsSweep > kLdsRed ? kLdsSweep : kLdsRed];
  __shared__ float red[kWavesPerBlock];
  float* lds = reinterpret_cast<float*>(smem) + wave_in_block() * sweep_lds_floats<R, 2, NV>();
  float vmax = 0.0f;
  const float mu = coef[K::kMu];
  const float* const mats[2] = {U, V};
  const float* vecs_[4] = {d, v, h, g};
  const float* const (&vecs)[NV] = reinterpret_cast<const float* const (&)[NV]>(vecs_);
  auto body = [&](long row, bool valid, float (&x)[2][R], float (&s)[NV]) {
    const float dd = s[0], vv = s[1], hh = s[2];
    const float t = dd * hh;
    const float ww = valid ? vv / dd : 0.0f;
    const float a = t + dot_row<R>(x[0], coef + K::kS1);
    const float b = ww - dot_row<R>(x[1], coef + K::kX1);
    const float Ph = dd * (a + dot_row<R>(x[1], coef + K::kS2));
    const float invPv = valid ? (b - dot_row<R>(x[0], coef + K::kX2)) / dd : 0.0f;
    const float nd = Ph * hh - vv * invPv;
    if (valid) {
      stream_store<NT>(nabla + row, nd);
      vmax = amaxf(vmax, fabsf(nd));
    }
    if constexpr (UPDATE_U) {
#pragma unroll
      for (int c = 0; c < R; ++c) x[0][c] = x[0][c] - mu * (a * coef[K::kC1 + c] - b * coef[K::kC2 + c]);
    } else {
      const float al = a + dot_row<R>(x[1], coef + K::kC1);
      const float be = b + dot_row<R>(x[1], coef + K::kC2);
#pragma unroll
      for (int c = 0; c < R; ++c) x[1][c] = x[1][c] - mu * (al * coef[K::kC1 + c] - be * coef[K::kC2 + c]);
    }
    if constexpr (FUSE) {
      const float tg = dd * s[NV - 1];          // d .* g        (rows past N: zero-filled)
      s[1] = tg;                                // -> the v and h slots of the LDS tile, read back by ColSum2
      s[2] = tg * nd;                           // d .* g .* nablaD
    }
  };
  if constexpr (FUSE) {
    std::conditional_t<(2 * R <= 64), ColSum2<R, 2, 1, 2>, ColSumPair<R, 1>> cs;
    sweep_rows<R, 2, NV, (UPDATE_U ? 0 : 1), NT>(mats, vecs, UPDATE_U ? U : V, N, lds, body, cs);
    cs.block_store(reinterpret_cast<double*>(smem), part_pq);
    __syncthreads();
  } else {
    sweep_rows<R, 2, NV, (UPDATE_U ? 0 : 1), NT>(mats, vecs, UPDATE_U ? U : V, N, lds, body);
  }
  block_max_store(vmax, red, part_max + blockIdx.x);
}

// Last sweep of the fused update -> apply: d <- d - (mu_d d) nablaD (psgd.py:582-584), then with the new d
//   out = d .* (g1 + V s2'),  g1 = d .* g + U s1'        (psgd.py:625-626 on the updated state)
// s1' = coef[0, R), s2' = coef[R, 2R) (k_fused_post); U, V are the updated factors.
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_uvd_final(const float* U, const float* V, float* d, const float* nabla,
                                                        const float* g, float* out, long N,
                                                        const float* __restrict__ coef, const float* __restrict__ maxbuf,
                                                        float step, float tiny) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 2, 3>()];
  const float mu = step / (maxbuf[0] + tiny);
  const float* const mats[2] = {U, V};
  const float* const vecs[3] = {d, nabla, g};
  sweep_rows<R, 2, 3, -1, NT>(mats, vecs, nullptr, N, lds[wave_in_block()],
                              [&](long row, bool valid, float (&x)[2][R], float (&s)[3]) {
                                const float dn = s[0] - (mu * s[0]) * s[1];
                                const float g1 = dn * s[2] + dot_row<R>(x[0], coef);
                                const float o = dn * (g1 + dot_row<R>(x[1], coef + R));
                                if (valid) {
                                  stream_store<NT>(d + row, dn);
                                  stream_store<NT>(out + row, o);
                                }
                              });
}

// M <- M - (a c1' - b c2')  (row-local rank-2 update of one factor, psgd.py:600-601 / :614-615 with the step size folded
// into c1, c2): building block of the wide-rank (r > 32) path, which works on column chunks of U and V.
template <int R, bool NT, bool STR = false>
__global__ __launch_bounds__(kThreads) void k_rank2_update(float* M, const float* a, const float* b, long N,
                                                           const float* __restrict__ coef, long ld = R) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][sweep_lds_floats<R, 1, 2>()];
  const float* const mats[1] = {M};
  const float* const vecs[2] = {a, b};
  sweep_rows<R, 1, 2, 0, NT, STR>(mats, vecs, M, N, lds[wave_in_block()],
                                  [&](long, bool, float (&x)[1][R], float (&s)[2]) {
#pragma unroll
                                    for (int c = 0; c < R; ++c) x[0][c] = x[0][c] - (s[0] * coef[c] - s[1] * coef[R + c]);
                                  }, NoTileHook(), RowStrides{{ld, ld}});
}

// ------------------------------------------------------- launch table ------
struct UvdOps {
  int tile_rows;
  int gram_len;
  // all launchers: grid chosen by caller (<= kMaxGrid), return hipError_t as int
  // `nt`: non-temporal policy for the big streams (see stream_load)
  int (*colreduce)(int nt, int nvec, const float* M, const float* a, const float* b, long N, float* part, int grid, hipStream_t st);
  int (*apply_s2)(int nt, const float* U, const float* d, const float* g, float* g1out, long N, const float* coef, float* part, int grid, hipStream_t st);
  int (*apply_s3)(int nt, const float* V, const float* d, float* out, long N, const float* coef, int grid, hipStream_t st);
  int (*rowdot_axpy)(int nt, const float* M, const float* x, float* out, long N, const float* coef, int grid, hipStream_t st);
  int (*update_gram)(int nt, const float* U, const float* V, const float* d, const float* v, const float* h, long N, double* part, int grid, hipStream_t st);
  // g / part_pq non-null: fused form that also reduces [Unew | Vnew]' [d.*g, d.*g.*nablaD] (see k_update_s2, ColSum2)
  int (*update_s2)(int nt, int update_U, float* U, float* V, const float* d, const float* v, const float* h, const float* g, long N, const float* coef, float* nabla, float* part_max, double* part_pq, int grid, hipStream_t st);
  int (*colreduce4)(int nt, const float* M, const float* const* x, long N, double* part, int grid, hipStream_t st);
  int (*rowdot_axpy4)(int nt, const float* M, const float* const* x, float* const* o, int ncols, long N, const float* coef, int grid, hipStream_t st);
  int (*rank2_update)(int nt, float* M, const float* a, const float* b, long N, const float* coef, int grid, hipStream_t st);
  int (*final_sweep)(int nt, const float* U, const float* V, float* d, const float* nabla, const float* g, float* out, long N, const float* coef, const float* maxbuf, float step, float tiny, int grid, hipStream_t st);
  // precond_grad_UVd_math on four columns of a matrix g per sweep (k_apply4_s1 .. s3)
  int (*apply4_s1)(int nt, const float* V, const float* d, const float* const* x, long N, double* part, int grid, hipStream_t st);
  int (*apply4_s2)(int nt, const float* U, const float* d, const float* const* x, float* const* o, int ncols, long N, const float* coef, double* part, int grid, hipStream_t st);
  int (*apply4_s3)(int nt, const float* V, const float* d, float* const* o, int ncols, long N, const float* coef, int grid, hipStream_t st);
  // strided forms of the wide-rank building blocks (element (row, c) at M[row * ld + c]); load_vec: floats per global access --
  // ld and the byte address of the view must be multiples of it
  int load_vec;
  int (*update_gram_ld)(const float* U, long ldU, const float* V, long ldV, const float* d, const float* v, const float* h, long N, double* part, int grid, hipStream_t st);
  int (*colreduce4_ld)(const float* M, long ld, const float* const* x, long N, double* part, int grid, hipStream_t st);
  int (*rowdot_axpy4_ld)(const float* M, long ld, const float* const* x, float* const* o, int ncols, long N, const float* coef, int grid, hipStream_t st);
  int (*rank2_update_ld)(float* M, long ld, const float* a, const float* b, long N, const float* coef, int grid, hipStream_t st);
  // max resident blocks per CU for each sweep kernel (occupancy query)
  int (*occupancy)(int which);
};

// Ranks 33 .. 64 (round 5): the kernels the wide-rank path needs, on the whole [N, r] matrix instead of column chunks -- the four-column
// building blocks and the three sweeps of the apply (uvd_wide_group.hip instantiates them; psgd_uvd_wide_* in psgd_uvd.hip)
struct UvdWideOps {
  int tile_rows;
  int (*colreduce4)(int nt, const float* M, const float* const* x, long N, double* part, int grid, hipStream_t st);
  int (*rowdot_axpy4)(int nt, const float* M, const float* const* x, float* const* o, int ncols, long N, const float* coef, int grid, hipStream_t st);
  int (*rank2_update)(int nt, float* M, const float* a, const float* b, long N, const float* coef, int grid, hipStream_t st);
  int (*apply4_s1)(int nt, const float* V, const float* d, const float* const* x, long N, double* part, int grid, hipStream_t st);
  int (*apply4_s2)(int nt, const float* U, const float* d, const float* const* x, float* const* o, int ncols, long N, const float* coef, double* part, int grid, hipStream_t st);
  int (*apply4_s3)(int nt, const float* V, const float* d, float* const* o, int ncols, long N, const float* coef, int grid, hipStream_t st);
  int (*update_s2)(int nt, int update_U, float* U, float* V, const float* d, const float* v, const float* h, const float* g, long N, const float* coef, float* nabla, float* part_max, double* part_pq, int grid, hipStream_t st);
  int (*final_sweep)(int nt, const float* U, const float* V, float* d, const float* nabla, const float* g, float* out, long N, const float* coef, const float* maxbuf, float step, float tiny, int grid, hipStream_t st);
};
const UvdWideOps* uvd_wide_ops_for_rank(int r);   // nullptr outside 33 .. 64

enum { kOccColreduce = 0, kOccApplyS2, kOccApplyS3, kOccRowdot, kOccGram, kOccUpdS2U, kOccUpdS2V, kOccUpdS2F, kOccFinal };

const UvdOps* uvd_ops_for_rank(int r);   // nullptr when r is not instantiated

// shared launch policy (psgd_uvd.hip; set through psgd_set_tuning)
int policy_nt(int64_t stream_bytes);     // non-temporal streams when the operands exceed the Infinity Cache
int policy_grid_blocks(int occupancy);   // blocks per CU after the tuning override
int device_cus();

}  // namespace psgd
