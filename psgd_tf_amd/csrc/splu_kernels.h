// splu_kernels.h -- CDNA4 (gfx950) streaming kernels for the sparse-LU preconditioner
//   Q = L U,  L = [L1 0; L2 diag(l3)],  U = [U1 U2; 0 diag(u3)]      reference: psgd.py:396-524
//
// Data layout in HBM is the reference's own: L12 = [L1; L2] is [N, r] row-major (L2 = rows r..N),
// U12 = [U1, U2] is [r, N] row-major (U2 = columns r..N, i.e. r contiguous "row vectors" of length
// N - r at a stride of N floats), l3/u3 are [N - r].
//
// The N - r "tail" rows are streamed with the UVd machinery (uvd_kernels.h): the [.., r] block L2 is
// staged tile by tile through wave-private LDS and read back one row per lane; the r row vectors of U2
// are per-row scalars of the same tile (coalesced 4-byte loads, also staged through LDS).  Everything
// r x r (L1, U1, the four triangular solves, the corner gradients) is done between the sweeps by one
// small fp64 block (psgd_splu.hip).  Four dependent sweeps per update, three per apply:
//
//   apply   s1: U2 g2            -> Ug1                         (reads U2, g2)
//           s2: Qg2 = L2 Ug1 + l3 u3 g2 (stored), L2' Qg2       (reads L2, l3, u3, g2)
//           s3: out2 = U2' LtQg1 + u3 l3 Qg2                    (reads U2, l3, u3, Qg2)
//   update  s1: U2 dg2                                          (as apply s1)
//           s2: L2' Qg2, L2' iQtx2                              (reads L2, U2, l3, u3, dx2, dg2)
//           s3: U2 iPx2, max|grad L|, max|grad U|, max l3/u3    (same reads)
//           s4: new L2, l3, U2, u3                              (same reads; writes L2, U2, l3, u3)
//         The row-local Qg2 and iQtx2 are recomputed in s3 and s4 (two r-long dot products per row)
//         rather than stored by s2: a thin output stream costs far more than its bytes on this memory
//         system (profiles/r01_store_stream_microbench.txt), and the update sweeps stay read-only until s4.
//
// The first rows of the tail are handled by a scalar "head" path (block 0) so that the streamed part starts at a
// row where r + head is a multiple of 32: 16-byte aligned L2 tiles for every r, 128-byte aligned U2 column streams
// whenever N is a multiple of 32.
#pragma once
#include "uvd_kernels.h"

namespace psgd {

// rows [-head, 0) relative to the (already shifted) sweep pointers, one lane each, block 0 only
template <int R, int NVEC, int WB, class Body>
__device__ __forceinline__ void sweep_head(const float* L2s, const float* const (&vecs)[NVEC], float* mat_out,
                                           int head, int wv, Body&& body) {
  const int tid = wv * 64 + lane_from_exec();        // (not threadIdx.x: see wave_in_block in uvd_kernels.h)
  if (blockIdx.x == 0 && tid < head) {
    const long row = (long)tid - head;
    float x[1][R];
    float s[NVEC];
#pragma unroll
    for (int k = 0; k < R; ++k) x[0][k] = L2s[row * R + k];
#pragma unroll
    for (int k = 0; k < NVEC; ++k) s[k] = vecs[k][row];
    body(row, true, x, s);
    if constexpr (WB >= 0) {
#pragma unroll
      for (int k = 0; k < R; ++k) mat_out[row * R + k] = x[0][k];
    }
  }
}

// layout of the dynamic LDS of the sweep kernels: [waves][sweep_lds_floats] then the block-reduction scratch
template <int R, int NVEC, int NRED>
constexpr int splu_lds_bytes() { return (kWavesPerBlock * sweep_lds_floats<R, 1, NVEC>() + kWavesPerBlock * NRED) * 4; }

// Column sweep: NVEC length-N vectors, no row-major operand.  Same pipeline as sweep_rows -- each wave keeps
// the next tile's loads in flight while it works on the current one, and the loaded values take a round trip
// through wave-private LDS (lane-to-same-lane) so that no load destination is loop-carried.  A plain
// grid-stride loop over the same streams ran at 4.4 TB/s where this form reaches ~6 TB/s.
// body(row, valid, s[NVEC]).
template <int NVEC>
struct ColCfg {
  static constexpr int kRowsPerLane = (NVEC <= 12) ? 4 : ((NVEC <= 24) ? 2 : 1);
  static constexpr int kTileRows = 64 * kRowsPerLane;
  static constexpr int kLdsFloats = NVEC * kTileRows;   // per wave
};

template <int NVEC, bool NT, class Body>
__device__ __forceinline__ void sweep_cols(const float* const (&vecs)[NVEC], long N, float* lds, int wv, Body&& body) {
  using C = ColCfg<NVEC>;
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * kWavesPerBlock + wv;
  const long nw = (long)gridDim.x * kWavesPerBlock;
  const long nfull = N / C::kTileRows;
  float pf[NVEC][C::kRowsPerLane];
  auto issue = [&](long tile) {
    const long row0 = tile * C::kTileRows;
#pragma unroll
    for (int k = 0; k < NVEC; ++k)
#pragma unroll
      for (int i = 0; i < C::kRowsPerLane; ++i) pf[k][i] = stream_load<NT>(vecs[k] + row0 + lane + 64 * i);
  };
  long tile = gw;
  if (tile < nfull) issue(tile);
  while (tile < nfull) {
#pragma unroll
    for (int k = 0; k < NVEC; ++k)
#pragma unroll
      for (int i = 0; i < C::kRowsPerLane; ++i) lds[k * C::kTileRows + lane + 64 * i] = pf[k][i];
    const long next = tile + nw;
    issue((next < nfull) ? next : tile);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) {
      float s[NVEC];
#pragma unroll
      for (int k = 0; k < NVEC; ++k) s[k] = lds[k * C::kTileRows + lane + 64 * i];
      body(tile * C::kTileRows + lane + 64 * i, true, s);
    }
    __builtin_amdgcn_wave_barrier();
    tile = next;
  }
  const long tail_rows = N - nfull * C::kTileRows;
  if (tail_rows > 0 && (nfull % nw) == gw) {
    const int lane = lane_from_exec();
    const long row0 = nfull * C::kTileRows;
#pragma unroll
    for (int i = 0; i < C::kRowsPerLane; ++i) {
      const int rit = lane + 64 * i;
      const bool valid = rit < tail_rows;
      float s[NVEC];
#pragma unroll
      for (int k = 0; k < NVEC; ++k) s[k] = valid ? vecs[k][row0 + rit] : 0.0f;
      body(row0 + rit, valid, s);
    }
  }
}

// rows [-head, 0) relative to the (already shifted) pointers of a column sweep, one lane each, block 0 only
template <int NVEC, class Body>
__device__ __forceinline__ void cols_head(const float* const (&vecs)[NVEC], int head, int wv, Body&& body) {
  const int tid = wv * 64 + lane_from_exec();
  if (blockIdx.x == 0 && tid < head) {
    const long row = (long)tid - head;
    float s[NVEC];
#pragma unroll
    for (int k = 0; k < NVEC; ++k) s[k] = vecs[k][row];
    body(row, true, s);
  }
}

// part[k] = sum_i U2[k][i] x[i]      psgd.py:430 / :506 (second matmul)
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_splu_u2dot(const float* U2, long ldu, const float* x, long n2, int head,
                                                         float* part) {
  const int wv = wave_in_block();
  constexpr int NV = R + 1;
  __shared__ float lds[kWavesPerBlock][ColCfg<NV>::kLdsFloats];
  __shared__ float red[kWavesPerBlock * R];
  float acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0f;
  const float* vecs_[NV];
#pragma unroll
  for (int k = 0; k < R; ++k) vecs_[k] = U2 + k * ldu;
  vecs_[R] = x;
  const float* const (&vecs)[NV] = reinterpret_cast<const float* const (&)[NV]>(vecs_);
  auto body = [&](long, bool, float (&s)[NV]) {
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = fmaf(s[k], s[R], acc[k]);
  };
  sweep_cols<NV, NT>(vecs, n2, lds[wv], wv, body);
  cols_head<NV>(vecs, head, wv, body);
  block_sum_store<R>(acc, red, part);
}

// apply sweep 2: Qg2 = L2 Ug1 + l3 (u3 g2) -> qg2 (stored), part = L2' Qg2      psgd.py:507,510,512
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_splu_apply_s2(const float* L2s, const float* l3, const float* u3,
                                                            const float* g2, float* qg2, long n2s, int head,
                                                            const float* __restrict__ coef, float* part) {
  const int wv = wave_in_block();
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  constexpr int LW = sweep_lds_floats<R, 1, 3>();
  float* red = dyn_lds + kWavesPerBlock * LW;
  float acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0f;
  const float* const mats[1] = {L2s};
  const float* const vecs[3] = {l3, u3, g2};
  auto body = [&](long row, bool valid, float (&x)[1][R], float (&s)[3]) {
    const float q = s[0] * (s[1] * s[2]) + dot_row<R>(x[0], coef);
    if (valid) stream_store<NT>(qg2 + row, q);
#pragma unroll
    for (int c = 0; c < R; ++c) acc[c] = fmaf(x[0][c], q, acc[c]);
  };
  sweep_rows<R, 1, 3, -1, NT>(mats, vecs, nullptr, n2s, dyn_lds + wv * LW, body);
  sweep_head<R, 3, -1>(L2s, vecs, nullptr, head, wv, body);
  block_sum_store<R>(acc, red, part);
}

// apply sweep 3: out2 = U2' LtQg1 + u3 (l3 Qg2), in place on the buffer holding Qg2      psgd.py:513,516
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_splu_apply_s3(const float* U2, long ldu, const float* l3, const float* u3,
                                                            float* out2, long n2, int head,
                                                            const float* __restrict__ coef) {
  const int wv = wave_in_block();
  constexpr int NV = R + 3;
  __shared__ float lds[kWavesPerBlock][ColCfg<NV>::kLdsFloats];
  const float* vecs_[NV];
#pragma unroll
  for (int k = 0; k < R; ++k) vecs_[k] = U2 + k * ldu;
  vecs_[R] = l3; vecs_[R + 1] = u3; vecs_[R + 2] = out2;
  const float* const (&vecs)[NV] = reinterpret_cast<const float* const (&)[NV]>(vecs_);
  auto body = [&](long row, bool valid, float (&s)[NV]) {
    float o = s[R + 1] * (s[R] * s[R + 2]);
#pragma unroll
    for (int k = 0; k < R; ++k) o = fmaf(s[k], coef[k], o);
    if (valid) stream_store<NT>(out2 + row, o);
  };
  sweep_cols<NV, NT>(vecs, n2, lds[wv], wv, body);
  cols_head<NV>(vecs, head, wv, body);
}

// coefficient block of update sweeps 3 and 4 (floats)
template <int R>
struct SpluCoef {
  static constexpr int c0 = 0;        // LtQg1
  static constexpr int c1 = R;        // iLiQtx1
  static constexpr int c2 = 2 * R;    // s3: Qg1      s4: a = L1s' Qg1
  static constexpr int c3 = 3 * R;    // s3: iQtx1    s4: b = L1s' iQtx1
  static constexpr int c4 = 4 * R;    // s3: Pg1      s4: c = U1s Pg1
  static constexpr int c5 = 5 * R;    // s3: dx1      s4: e = U1s dx1
  static constexpr int c6 = 6 * R;    // Ug1
  static constexpr int c7 = 7 * R;    // iUtx1
  static constexpr int sc = 8 * R;    // s4: sL, sU, rho, 1/rho
};

// Qg2 (psgd.py:431,434) and iQtx2 (:437,439) of one tail row
template <int R>
__device__ __forceinline__ void splu_row_qg_iq(const float (&lrow)[R], const float* s /* U2 col */, float l, float u,
                                               float g, float xx, const float* __restrict__ ug1,
                                               const float* __restrict__ iutx1, float& q, float& iq) {
  q = l * (u * g) + dot_row<R>(lrow, ug1);
  float du = 0.0f;
#pragma unroll
  for (int k = 0; k < R; ++k) du = fmaf(s[k], iutx1[k], du);
  iq = ((xx - du) / u) / l;
}

// Pg2 (psgd.py:443,446) and iPx2 (:449,451) of one tail row
template <int R>
__device__ __forceinline__ void splu_row_pg_ipx(const float (&lrow)[R], const float* s /* U2 col */, float l, float u,
                                                float q, float iq, const float* __restrict__ coef, float& pg2,
                                                float& ipx2) {
  using K = SpluCoef<R>;
  float du = 0.0f;
#pragma unroll
  for (int k = 0; k < R; ++k) du = fmaf(s[k], coef[K::c0 + k], du);
  pg2 = du + u * (l * q);
  ipx2 = ((iq - dot_row<R>(lrow, coef + K::c1)) / l) / u;
}

// update sweep 2      psgd.py:431,434 (Qg2), :437,439 (iQtx2), and the L2' products of :440, :442
//   coef = [Ug1 | iUtx1];  per-row vectors: U2 columns, l3, u3, dx2, dg2
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_splu_upd_s2(const float* L2s, const float* U2s, long ldu,
                                                          const float* l3, const float* u3, const float* x2,
                                                          const float* g2, long n2s, int head,
                                                          const float* __restrict__ coef, float* part) {
  const int wv = wave_in_block();
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  constexpr int NV = R + 4;
  constexpr int LW = sweep_lds_floats<R, 1, NV>();
  float* red = dyn_lds + kWavesPerBlock * LW;
  float acc[2 * R];
#pragma unroll
  for (int c = 0; c < 2 * R; ++c) acc[c] = 0.0f;
  const float* const mats[1] = {L2s};
  const float* vecs_[NV];
#pragma unroll
  for (int k = 0; k < R; ++k) vecs_[k] = U2s + k * ldu;
  vecs_[R] = l3; vecs_[R + 1] = u3; vecs_[R + 2] = x2; vecs_[R + 3] = g2;
  const float* const (&vecs)[NV] = reinterpret_cast<const float* const (&)[NV]>(vecs_);
  auto body = [&](long, bool valid, float (&x)[1][R], float (&s)[NV]) {
    const float l = valid ? s[R] : 1.0f, u = valid ? s[R + 1] : 1.0f;
    float q, iq;
    splu_row_qg_iq<R>(x[0], s, l, u, s[R + 3], s[R + 2], coef, coef + R, q, iq);
#pragma unroll
    for (int c = 0; c < R; ++c) {
      acc[c] = fmaf(x[0][c], q, acc[c]);
      acc[R + c] = fmaf(x[0][c], iq, acc[R + c]);
    }
  };
  sweep_rows<R, 1, NV, -1, NT>(mats, vecs, nullptr, n2s, dyn_lds + wv * LW, body);
  sweep_head<R, NV, -1>(L2s, vecs, nullptr, head, wv, body);
  block_sum_store<2 * R>(acc, red, part);
}

// update sweep 3: part = U2 iPx2 (for :452); pmax[0..3] = max|grad2,3 of L| (:457-461), max|grad2,3 of U|
// (:470-474), max l3, max u3 (:411-412)
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_splu_upd_s3(const float* L2s, const float* U2s, long ldu,
                                                          const float* l3, const float* u3, const float* g2,
                                                          const float* x2, long n2s, int head,
                                                          const float* __restrict__ coef, float* part, float* pmax) {
  const int wv = wave_in_block();
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  using K = SpluCoef<R>;
  constexpr int NV = R + 4;
  constexpr int LW = sweep_lds_floats<R, 1, NV>();
  float* red = dyn_lds + kWavesPerBlock * LW;
  float acc[R];
#pragma unroll
  for (int c = 0; c < R; ++c) acc[c] = 0.0f;
  float mL = 0.0f, mU = 0.0f, ml3 = -INFINITY, mu3 = -INFINITY;
  const float* const mats[1] = {L2s};
  const float* vecs_[NV];
#pragma unroll
  for (int k = 0; k < R; ++k) vecs_[k] = U2s + k * ldu;
  vecs_[R] = l3; vecs_[R + 1] = u3; vecs_[R + 2] = g2; vecs_[R + 3] = x2;
  const float* const (&vecs)[NV] = reinterpret_cast<const float* const (&)[NV]>(vecs_);
  auto body = [&](long, bool valid, float (&x)[1][R], float (&s)[NV]) {
    const float l = valid ? s[R] : 1.0f, u = valid ? s[R + 1] : 1.0f;
    const float g = s[R + 2], xx = s[R + 3];
    float q, iq, pg2, ipx2;
    splu_row_qg_iq<R>(x[0], s, l, u, g, xx, coef + K::c6, coef + K::c7, q, iq);
    splu_row_pg_ipx<R>(x[0], s, l, u, q, iq, coef, pg2, ipx2);
    float a = fabsf(q * q - iq * iq), b = fabsf(pg2 * g - xx * ipx2);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      acc[k] = fmaf(s[k], ipx2, acc[k]);
      a = amaxf(a, fabsf(q * coef[K::c2 + k] - iq * coef[K::c3 + k]));
      b = amaxf(b, fabsf(coef[K::c4 + k] * g - coef[K::c5 + k] * ipx2));
    }
    if (valid) {
      mL = amaxf(mL, a);
      mU = amaxf(mU, b);
      ml3 = nmaxf(ml3, l);
      mu3 = nmaxf(mu3, u);
    }
  };
  sweep_rows<R, 1, NV, -1, NT>(mats, vecs, nullptr, n2s, dyn_lds + wv * LW, body);
  sweep_head<R, NV, -1>(L2s, vecs, nullptr, head, wv, body);
  block_sum_store<R>(acc, red, part);
  __syncthreads();
  const int G = gridDim.x;
  block_max_store(mL, red, pmax + blockIdx.x);
  __syncthreads();
  block_max_store(mU, red, pmax + G + blockIdx.x);
  __syncthreads();
  block_max_store<true>(ml3, red, pmax + 2 * G + blockIdx.x);
  __syncthreads();
  block_max_store<true>(mu3, red, pmax + 3 * G + blockIdx.x);
}

// update sweep 4: the tail of :463-465 and :476-478 on the rho-balanced factors (:414-417)
template <int R, bool NT>
__global__ __launch_bounds__(kThreads) void k_splu_upd_s4(const float* L2s, const float* U2s, long ldu,
                                                          const float* l3, const float* u3, const float* g2,
                                                          const float* x2, float* L2o, float* U2o, float* l3o,
                                                          float* u3o, long n2s, int head,
                                                          const float* __restrict__ coef) {
  const int wv = wave_in_block();
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  using K = SpluCoef<R>;
  constexpr int NV = R + 4;
  constexpr int LW = sweep_lds_floats<R, 1, NV>();
  const float* const mats[1] = {L2s};
  const float* vecs_[NV];
#pragma unroll
  for (int k = 0; k < R; ++k) vecs_[k] = U2s + k * ldu;
  vecs_[R] = l3; vecs_[R + 1] = u3; vecs_[R + 2] = g2; vecs_[R + 3] = x2;
  const float* const (&vecs)[NV] = reinterpret_cast<const float* const (&)[NV]>(vecs_);
  const float sL = coef[K::sc], sU = coef[K::sc + 1], rho = coef[K::sc + 2], irho = coef[K::sc + 3];
  auto body = [&](long row, bool valid, float (&x)[1][R], float (&s)[NV]) {
    const float l = valid ? s[R] : 1.0f, u = valid ? s[R + 1] : 1.0f;
    const float g = s[R + 2], xx = s[R + 3];
    float q, iq, pg2, ipx2;
    splu_row_qg_iq<R>(x[0], s, l, u, g, xx, coef + K::c6, coef + K::c7, q, iq);
    splu_row_pg_ipx<R>(x[0], s, l, u, q, iq, coef, pg2, ipx2);
    const float gl3 = sL * (q * q - iq * iq);
    const float gu3 = sU * (pg2 * g - xx * ipx2);
    const float sq = sL * q, siq = sL * iq, sg = sU * g, sp = sU * ipx2;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const float lk = x[0][k] * irho;
      x[0][k] = lk - (sq * coef[K::c2 + k] - siq * coef[K::c3 + k]) - gl3 * lk;
      const float uk = rho * s[k];
      const float un = uk - (coef[K::c4 + k] * sg - coef[K::c5 + k] * sp) - gu3 * uk;
      if (valid) stream_store<NT>(U2o + k * ldu + row, un);
    }
    if (valid) {
      const float ls = l * irho, us = rho * u;
      stream_store<NT>(l3o + row, ls - gl3 * ls);
      stream_store<NT>(u3o + row, us - gu3 * us);
    }
  };
  sweep_rows<R, 1, NV, 0, NT>(mats, vecs, L2o, n2s, dyn_lds + wv * LW, body);
  sweep_head<R, NV, 0>(L2s, vecs, L2o, head, wv, body);
}

// ------------------------------------------------------- launch table ------
struct SpluOps {
  int tile_rows;
  int lds_apply_s2, lds_upd_s2, lds_upd_s3, lds_upd_s4;   // dynamic LDS bytes per block
  int (*u2dot)(int nt, const float* U2, long ldu, const float* x, long n2, int head, float* part, int grid, hipStream_t st);
  int (*apply_s2)(int nt, const float* L2s, const float* l3, const float* u3, const float* g2, float* qg2, long n2s,
                  int head, const float* coef, float* part, int grid, hipStream_t st);
  int (*apply_s3)(int nt, const float* U2, long ldu, const float* l3, const float* u3, float* out2, long n2, int head,
                  const float* coef, int grid, hipStream_t st);
  int (*upd_s2)(int nt, const float* L2s, const float* U2s, long ldu, const float* l3, const float* u3, const float* x2,
                const float* g2, long n2s, int head, const float* coef, float* part, int grid, hipStream_t st);
  int (*upd_s3)(int nt, const float* L2s, const float* U2s, long ldu, const float* l3, const float* u3, const float* g2,
                const float* x2, long n2s, int head, const float* coef, float* part, float* pmax, int grid,
                hipStream_t st);
  int (*upd_s4)(int nt, const float* L2s, const float* U2s, long ldu, const float* l3, const float* u3, const float* g2,
                const float* x2, float* L2o, float* U2o, float* l3o, float* u3o, long n2s, int head, const float* coef,
                int grid, hipStream_t st);
  int (*occupancy)(int which);   // resident blocks per CU (0..5 in the order above)
};

const SpluOps* splu_ops_for_rank(int r);   // nullptr when r is not instantiated

}  // namespace psgd
