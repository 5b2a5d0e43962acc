// uvd_wide_gram.hip -- the Gram of W = [U | V | t | w] (t = d .* h, w = v ./ d; psgd.py:569-615 needs every inner product of these
// 2r + 2 columns) for ranks 32 < r <= 64 in ONE sweep over U and V (round 5).
//
// The rank-templated Gram kernel (uvd_kernels.h: k_update_gram, r <= 32) keeps one fp32 MFMA accumulator set and one fp64 set per
// pair of 16-column blocks in every wave's registers: 15 pairs at r = 32, 45 at r = 64 -- 540 registers.  The wide-rank path therefore
// ran it on PAIRS OF COLUMN CHUNKS (uvd_wide.py), one sweep per pair: three passes over U and V at r = 64 (six pair sweeps), and the
// update reached 0.23 of the specialised rate.  Here a workgroup of sixteen waves shares one 32-row tile: every element is split 3-way
// into bf16 (x = h + m + l exactly, as in k_update_gram) ONCE, into bf16 planes in LDS (16-byte units of 8 rows of one column, row-group
// major), and the block pairs are dealt, in 2 x 2 groups, to eleven multiplying waves (<= 5 pairs each at r = 64: six
// v_mfma_f32_16x16x32_bf16 per pair, fp32 chains of 256 rows folded into fp64).  Partials: one fp64 tile set per workgroup (each pair
// is owned by exactly one wave: no cross-wave reduction), reduced in block order and scattered into the dense symmetric Gram
// [ncol][ncol] by k_gram_wide_finish (fixed order: reproducible, symmetric to the bit).
#include "uvd_kernels.h"

#include "psgd_hip.h"

#include <cstdlib>

namespace psgd {

#ifndef GW_ACC64_LDS
#define GW_ACC64_LDS 0  // round 6, review item 9 (one bounded experiment): 1 = the fp64 sums of a multiplying wave live in LDS, not in 40 registers
#endif
#ifndef GW_DBG
#define GW_DBG 0        // what-if builds (wrong results): 2 = no MFMA phase, 4 = no split / plane writes (loads stay), 8 = no t / w columns
#endif
constexpr int kGwRows = 32;                  // rows per tile = the K extent of one MFMA
constexpr int kGwThreads = 1024;             // sixteen waves: four split, one for t / w, eleven multiply
constexpr int kGwMulWaves = 11;              // waves 5 .. 15 multiply: block pair p belongs to wave p mod 11, accumulator slot p / 11
constexpr int kGwDepth = 4;                  // tiles the split role requests ahead (even: a tile's plane buffer is its ordinal's parity)

__host__ __device__ constexpr int gw_pair_index(int NB, int bi, int bj) { return bi * NB - (bi * (bi - 1)) / 2 + (bj - bi); }

// The multiplying waves work on 2 x 2 groups of 16-column blocks ("super pairs"): the four fragments (x 3 planes) of a group are read
// once for its (up to) four block pairs -- 135 fragment reads per tile at r = 64 instead of 243 with one pair at a time, and LDS
// bandwidth is what the kernel runs out of first.  Super pairs are dealt to the waves at compile time, heaviest first onto the least
// loaded wave; a wave's pairs sit in consecutive accumulator slots.
struct GwPlan {
  int owner[15];       // multiplying wave of super pair sp = gw_pair_index(NS, si, sj)
  int slot0[15];       // its first accumulator slot
  int max_load;        // accumulator slots a wave needs
  int base[kGwMulWaves];   // first GLOBAL slot of a wave (prefix sums of the loads): GW_ACC64_LDS
  int load[kGwMulWaves];
};
template <int NB>
constexpr GwPlan gw_make_plan() {
  constexpr int NS = (NB + 1) / 2, NSP = NS * (NS + 1) / 2;
  GwPlan p{};
  int load[kGwMulWaves] = {};
  int wt[15] = {};
  bool placed[15] = {};
  for (int si = 0; si < NS; ++si)
    for (int sj = si; sj < NS; ++sj) {
      const int na = (NB - 2 * si) < 2 ? (NB - 2 * si) : 2, nb = (NB - 2 * sj) < 2 ? (NB - 2 * sj) : 2;
      wt[gw_pair_index(NS, si, sj)] = si == sj ? na * (na + 1) / 2 : na * nb;
    }
  for (int round = 0; round < NSP; ++round) {
    int best = -1;
    for (int sp = 0; sp < NSP; ++sp)
      if (!placed[sp] && (best < 0 || wt[sp] > wt[best])) best = sp;
    int w = 0;
    for (int k = 1; k < kGwMulWaves; ++k)
      if (load[k] < load[w]) w = k;
    p.owner[best] = w;
    p.slot0[best] = load[w];
    load[w] += wt[best];
    placed[best] = true;
  }
  for (int k = 0; k < kGwMulWaves; ++k)
    if (load[k] > p.max_load) p.max_load = load[k];
  for (int k = 0, b = 0; k < kGwMulWaves; ++k) { p.base[k] = b; p.load[k] = load[k]; b += load[k]; }
  return p;
}
template <int NB> struct GwPlanOf { static constexpr GwPlan value = gw_make_plan<NB>(); };

template <int N> struct GwInt { static constexpr int value = N; };
template <int I, int N, class F>
__device__ __forceinline__ void gw_static_for(F&& f) {
  if constexpr (I < N) {
    f(GwInt<I>{});
    gw_static_for<I + 1, N>(f);
  }
}

// 16-byte unit of the bf16 planes that holds rows 8 rg .. 8 rg + 7 of column col: row-group major, NC (a multiple of 16) columns per
// group.  A fragment read (lane -> column lane & 15 of a block, row group lane >> 4) takes 16 consecutive units of four groups, and the
// groups' bases are 16-unit aligned: distinct banks inside every lane group ds_read_b128 is served in (MI355X_MICROARCH.md, LDS).  The
// split role's writes (a wave = 64 consecutive columns of one row group) are 1 KiB contiguous; with the column-major units of the first
// version (col * 4 + rg) they were four-way bank conflicts, a third of the kernel's LDS time.
#ifndef GW_UNIT_COLMAJOR
template <int NC> __device__ __forceinline__ int gw_unit(int col, int rg) { return rg * NC + col; }
#else
template <int NC> __device__ __forceinline__ int gw_unit(int col, int rg) { return col * 4 + (rg ^ ((col >> 2) & 2)); }
#endif

// Sixteen waves, three roles, ONE barrier per tile that swaps the two plane buffers.  Waves 0-3 ("split"): thread (column, 8-row
// group) brings its 8 + 8 values of U and V straight from global memory into registers -- every load coalesced across the wave, a ring of
// kGwDepth tiles in flight -- and writes tile t + 1's planes while tile t is multiplied; there is no fp32 stage and no transposing
// pass.  Wave 4: the t = d .* h and w = v ./ d columns.  Waves 5-15 ("mfma"): the block pairs of tile t from the other buffer.  The
// matrix work of a tile is 270 MFMAs = 1080 cycles per SIMD -- most of the time HBM gives a 16-KiB tile per CU at r = 64 -- so the split
// has to run beside it, not before it: a version with all waves doing load -> split -> barrier -> MFMA -> barrier ran at 2.1 TB/s (1.6
// with four waves of 12 pairs), whatever the prefetch depth.
// TAIL = false: the tiles [0, ntiles) are whole (no row is checked against N: the loads are a per-tile scalar base + a per-thread offset
// computed once); TAIL = true: one workgroup on the last, partial tile `tile0`, every row clamped and masked.
template <int NB, bool NT, bool TAIL>
__global__ __launch_bounds__(kGwThreads) void k_gram_wide(const float* __restrict__ U, const float* __restrict__ V,
                                                          const float* __restrict__ d, const float* __restrict__ v,
                                                          const float* __restrict__ h, long N, int r, long ntiles, long tile0,
                                                          double* __restrict__ part) {
  constexpr int NC = NB * 16;                       // padded columns
  constexpr int NPW = GwPlanOf<NB>::value.max_load;
  constexpr int NP = NB * (NB + 1) / 2;
  typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) u32x4g PL[2][3][NC * 4 + 16];     // (+ 16 units: the scratch of the t / w wave)
#if GW_ACC64_LDS
  __shared__ double A64[NP][4][64];                                         // [global slot][e][lane]: 2 KiB per block pair
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long G = gridDim.x;
  const long tfirst = TAIL ? tile0 : (long)blockIdx.x;       // this workgroup's first tile; TAIL: its only one
  if (TAIL) ntiles = tile0 + 1;

  // The roles are separate code paths with their own loops (their states must not be live at the same time: 128 registers); all of
  // them execute the same number of barriers: one after the planes are zeroed, one after the prologue, one per tile.
  // Every workgroup runs its tile count rounded up to a multiple of the ring depth, with NO condition around anything that loads: the
  // loading roles clamp the tile index to the workgroup's last tile (a few re-reads at the end, into buffers nobody multiplies), the
  // multiplying role skips the padding iterations.  (With `if (t < ntiles) fetch(..)` the compiler cannot count the loads in flight at
  // the join and waits for ALL of them at the next use: the ring of four tiles was one tile deep, 0.62 us per tile whatever its size.)
  const long nmine = (ntiles - tfirst + G - 1) / G;                         // >= 1
  const long tlast = tfirst + (nmine - 1) * G;
  auto clampt = [&](long t) { return t < tlast ? t : tlast; };
  auto tile_loop = [&](auto&& prologue, auto&& step) {   // step(q, t): iteration of tile t, ordinal q mod depth (compile time)
    __syncthreads();
    prologue();
    __syncthreads();
    for (long i = 0; i < nmine; i += kGwDepth)
      gw_static_for<0, kGwDepth>([&](auto ic) {
        step(ic, tfirst + (i + decltype(ic)::value) * G);
        __syncthreads();
      });
  };
  if (wave < 4) {
    // ---- split role: thread (column c < r, row group g < 4) -- 4r <= 256 threads -- owns the items (c, g) of U and (r + c, g) of V:
    // 8 + 8 loads per tile, each coalesced across the wave, every address a per-tile SCALAR base plus one 32-bit per-thread offset.
    // Every load is unconditional and nothing else in this role loads: a branch between load variants, or a conditional load beside
    // the ring, makes the compiler wait for EVERY outstanding load at the join (0.6 TB/s with branches per element).  kGwDepth tiles
    // are requested ahead (a ring of register buffers, the tile loop unrolled by the depth).
    const int col = tid % r, g0 = tid / r;
    const bool on = g0 < 4;
    const unsigned e0 = on ? (unsigned)(g0 * 8 * r + col) : 0u;       // element offset of the item's first value inside a tile
    const unsigned eb0 = e0 * 4u;
    const int un0 = gw_unit<NC>(col, on ? g0 : 0), un1 = gw_unit<NC>(r + col, on ? g0 : 0);
    for (int i = tid; i < 2 * 3 * (NC * 4 + 16); i += 256) (&PL[0][0][0])[i] = u32x4g{0u, 0u, 0u, 0u};     // (pad columns stay zero)
    float px[kGwDepth][16];
    auto fetch = [&](long t, auto ic) {
      constexpr int ib = decltype(ic)::value;
      if constexpr (!TAIL) {
        // Buffer loads: descriptor = the tile of U (V), built from scalars only; the thread's byte offset in the 32-bit voffset, the
        // row's in the scalar soffset -- no 64-bit vector add per load, and nothing the compiler can turn into 64 induction pointers
        // (with plain pointers it did: 128 registers of addresses, spilled).
        const long toff0 = t * kGwRows * r;
        const long toff = ((long)__builtin_amdgcn_readfirstlane((int)(toff0 >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)toff0);
        const unsigned tbytes = (unsigned)(kGwRows * r * 4);
        const auto ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U + toff), 0, tbytes, 0x00020000);
        const auto rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V + toff), 0, tbytes, 0x00020000);
        constexpr int aux = NT ? 2 : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          px[ib][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, eb0, j * r * 4, aux));
          px[ib][8 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, eb0, j * r * 4, aux));
        }
      } else {
        const long row0 = t * kGwRows + (on ? g0 : 0) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long row = row0 + j, rc = (row < N ? row : N - 1) * r + col;
          const float xu = stream_load<NT>(U + rc), xv = stream_load<NT>(V + rc);
          px[ib][j] = (on && row < N) ? xu : 0.0f;
          px[ib][8 + j] = (on && row < N) ? xv : 0.0f;
        }
      }
    };
    auto split_to = [&](int buf, auto ic) {
      constexpr int ib = decltype(ic)::value;
      if (GW_DBG & 4) {                              // (what-if: the loads stay, consumed by nothing but this statement)
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("" ::"v"(px[ib][j]));
        return;
      }
      if (!on) return;
      bf16x8 fh, fm, fl;
      float x[8];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = px[ib][hh * 8 + j];
        split3_bf16(x, fh, fm, fl);
        const int un = hh ? un1 : un0;
        PL[buf][0][un] = __builtin_bit_cast(u32x4g, fh);
        PL[buf][1][un] = __builtin_bit_cast(u32x4g, fm);
        PL[buf][2][un] = __builtin_bit_cast(u32x4g, fl);
      }
    };
    const long t0 = tfirst;
    tile_loop(
        [&]() {      // tiles 0 .. depth - 1 requested (ring slot = ordinal mod depth), tile 0 split into buffer 0, its slot refilled
          gw_static_for<0, kGwDepth>([&](auto ic) { fetch(clampt(t0 + decltype(ic)::value * G), ic); });
          split_to(0, GwInt<0>{});
          fetch(clampt(t0 + kGwDepth * G), GwInt<0>{});
        },
        [&](auto ic, long t) {                       // while tile t is multiplied: tile t + 1 into the other buffer, its slot refilled
          constexpr int q = decltype(ic)::value, slot = (q + 1) % kGwDepth;
          split_to((q + 1) & 1, GwInt<slot>{});
          fetch(clampt(t + (1 + kGwDepth) * G), GwInt<slot>{});
        });
    return;
  }
  if (wave == 4) {
    // ---- the columns t = d .* h and w = v ./ d: lane l computes ONE value of the tile's 8 items (item l >> 3 = (column (l >> 3) & 1,
    // row group l >> 4), row l & 7), the wave passes the 64 values through a 256-byte LDS scratch, lanes 0-7 split 8 values each.
    // Same ring as the split role.  (Under `if (tid < 8)` inside the split role these loads made the compiler wait for that role's
    // whole ring at the join; one tile ahead in the mfma role an iteration lasted one memory latency.)
    float* TW = reinterpret_cast<float*>(&PL[1][2][NC * 4]);          // (behind the planes: 16 extra units)
    const int tw_un = gw_unit<NC>(2 * r + (lane & 1), (lane >> 1) & 3);
    // The ring holds the RAW d, h, v of the lane's row: the quotient and the product are formed when the tile is split.  (Forming them
    // in `fetch` -- or pinning the three loads there with an empty asm -- makes that statement wait for the loads it has just issued:
    // one full memory latency per tile on this wave, 0.58 us, which every other wave then spends at the barrier.)
    float pd[kGwDepth], ph[kGwDepth], pv[kGwDepth];
    bool pok[kGwDepth];                                 // (rows past N exist in the TAIL launch only; they are loaded from row N - 1)
    const int it = lane >> 3;
    auto fetch = [&](long t, auto ic) {
      const long row = t * kGwRows + (it >> 1) * 8 + (lane & 7), rc = (!TAIL || row < N) ? row : N - 1;
      pd[decltype(ic)::value] = d[rc];
      ph[decltype(ic)::value] = h[rc];
      pv[decltype(ic)::value] = v[rc];
      pok[decltype(ic)::value] = !TAIL || row < N;
    };
    auto split_to = [&](int b, auto ic) {
      const float dd = pd[decltype(ic)::value], hh = ph[decltype(ic)::value], vv = pv[decltype(ic)::value];
      if (GW_DBG & 8) {                              // (what-if: no t / w columns)
        asm volatile("" ::"v"(dd), "v"(hh), "v"(vv));
        return;
      }
      const float val = (it & 1) ? vv / dd : dd * hh;
      TW[lane] = pok[decltype(ic)::value] ? val : 0.0f;        // [item][row]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane < 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = TW[lane * 8 + j];
        bf16x8 fh, fm, fl;
        split3_bf16(x, fh, fm, fl);
        PL[b][0][tw_un] = __builtin_bit_cast(u32x4g, fh);
        PL[b][1][tw_un] = __builtin_bit_cast(u32x4g, fm);
        PL[b][2][tw_un] = __builtin_bit_cast(u32x4g, fl);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();                 // (the scratch is free again)
    };
    const long t0 = tfirst;
    tile_loop(
        [&]() {
          gw_static_for<0, kGwDepth>([&](auto ic) { fetch(clampt(t0 + decltype(ic)::value * G), ic); });
          split_to(0, GwInt<0>{});
          fetch(clampt(t0 + kGwDepth * G), GwInt<0>{});
        },
        [&](auto ic, long t) {
          constexpr int q = decltype(ic)::value, slot = (q + 1) % kGwDepth;
          split_to((q + 1) & 1, GwInt<slot>{});
          fetch(clampt(t + (1 + kGwDepth) * G), GwInt<slot>{});
        });
    return;
  }
  // ---- mfma role: waves 5 .. 15
  const int mw = wave - 5;
  f32x4 acc[NPW];
#if GW_ACC64_LDS
  int my_base = 0, my_load = 0;
  gw_static_for<0, kGwMulWaves>([&](auto kc) {
    if (mw == decltype(kc)::value) { my_base = GwPlanOf<NB>::value.base[decltype(kc)::value]; my_load = GwPlanOf<NB>::value.load[decltype(kc)::value]; }
  });
  double (*my64)[4][64] = A64 + my_base;
#pragma unroll
  for (int p = 0; p < NPW; ++p) {
    acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p < my_load)
#pragma unroll
      for (int e = 0; e < 4; ++e) my64[p][e][lane] = 0.0;
  }
  auto fold = [&]() {
#pragma unroll
    for (int p = 0; p < NPW; ++p) {
      if (p < my_load)
#pragma unroll
        for (int e = 0; e < 4; ++e) my64[p][e][lane] += (double)acc[p][e];
      acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
#else
  double acc64[NPW][4];
#pragma unroll
  for (int p = 0; p < NPW; ++p) {
    acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) acc64[p][e] = 0.0;
  }
  auto fold = [&]() {
#pragma unroll
    for (int p = 0; p < NPW; ++p) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc64[p][e] += (double)acc[p][e];
      acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
#endif
  int since = 0;
  tile_loop([]() {}, [&](auto ic, long t) {
    if (t >= ntiles) return;                         // (padding iteration of the last round)
    const int buf = decltype(ic)::value & 1;         // (depth is even: the buffer of a tile is its ordinal's parity)
    auto frag = [&](int b, bf16x8 (&f)[3]) {
      const int un = gw_unit<NC>(b * 16 + (lane & 15), lane >> 4);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) f[pl] = __builtin_bit_cast(bf16x8, PL[buf][pl][un]);
    };
    constexpr int NS = (NB + 1) / 2;
    gw_static_for<0, NS>([&](auto sic) {
      constexpr int si = decltype(sic)::value;
      gw_static_for<si, NS>([&](auto sjc) {
        constexpr int sj = decltype(sjc)::value;
        constexpr int sp = gw_pair_index(NS, si, sj);
        constexpr int own = GwPlanOf<NB>::value.owner[sp], s0 = GwPlanOf<NB>::value.slot0[sp];
        constexpr int na = (NB - 2 * si) < 2 ? (NB - 2 * si) : 2, nb = (NB - 2 * sj) < 2 ? (NB - 2 * sj) : 2;
        if (mw == own && !(GW_DBG & 2)) {
          bf16x8 fa[2][3], fb[2][3];
          gw_static_for<0, nb>([&](auto c) { frag(2 * sj + decltype(c)::value, fb[decltype(c)::value]); });
          if constexpr (si != sj) gw_static_for<0, na>([&](auto c) { frag(2 * si + decltype(c)::value, fa[decltype(c)::value]); });
          // the six products of a pair keep their order (smallest first); the group's pairs are interleaved product by product, so
          // that consecutive MFMAs belong to different accumulators
          gw_static_for<0, 6>([&](auto tc) {
            constexpr int term = decltype(tc)::value;
            constexpr int pa = term == 0 ? 1 : term == 1 ? 0 : term == 2 ? 2 : term == 3 ? 0 : term == 4 ? 1 : 0;   // m h l h m h
            constexpr int pb = term == 0 ? 1 : term == 1 ? 2 : term == 2 ? 0 : term == 3 ? 1 : term == 4 ? 0 : 0;   // m l h m h h
            gw_static_for<0, na>([&](auto ac) {
              constexpr int a = decltype(ac)::value;
              gw_static_for<(si == sj ? a : 0), nb>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                constexpr int slot = s0 + (si == sj ? (a == 0 ? b : nb + (b - 1)) : a * nb + b);
                if constexpr (si == sj) acc[slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[a][pa], fb[b][pb], acc[slot], 0, 0, 0);
                else acc[slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][pa], fb[b][pb], acc[slot], 0, 0, 0);
              });
            });
          });
        }
      });
    });
    if (++since == 256 / kGwRows) {
      fold();
      since = 0;
    }
  });
  fold();
  // every pair is owned by one wave: its partial goes straight out, [block][pair][e * 64 + lane]
  double* dst = part + (long)blockIdx.x * NP * 256;
  {
    constexpr int NS = (NB + 1) / 2;
    gw_static_for<0, NS>([&](auto sic) {
      constexpr int si = decltype(sic)::value;
      gw_static_for<si, NS>([&](auto sjc) {
        constexpr int sj = decltype(sjc)::value;
        constexpr int sp = gw_pair_index(NS, si, sj);
        constexpr int own = GwPlanOf<NB>::value.owner[sp], s0 = GwPlanOf<NB>::value.slot0[sp];
        constexpr int na = (NB - 2 * si) < 2 ? (NB - 2 * si) : 2, nb = (NB - 2 * sj) < 2 ? (NB - 2 * sj) : 2;
        if (mw == own) {
          gw_static_for<0, na>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            gw_static_for<(si == sj ? a : 0), nb>([&](auto bc) {
              constexpr int b = decltype(bc)::value;
              constexpr int slot = s0 + (si == sj ? (a == 0 ? b : nb + (b - 1)) : a * nb + b);
              constexpr int pi = gw_pair_index(NB, 2 * si + a, 2 * sj + b);
#pragma unroll
#if GW_ACC64_LDS
              for (int e = 0; e < 4; ++e) dst[pi * 256 + e * 64 + lane] = my64[slot][e][lane];
#else
              for (int e = 0; e < 4; ++e) dst[pi * 256 + e * 64 + lane] = acc64[slot][e];
#endif
            });
          });
        }
      });
    });
  }
}

// the workgroups' partials summed in block order, scattered into the dense symmetric Gram G [ncol][ncol] (fp64)
__global__ __launch_bounds__(kThreads) void k_gram_wide_finish(const double* __restrict__ part, int nblocks, int NB, int ncol,
                                                               double* __restrict__ G) {
  const int NP = NB * (NB + 1) / 2;
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  if (idx >= NP * 256) return;
  const int p = idx >> 8, q = idx & 255, e = q >> 6, l = q & 63;
  int bi = 0, rem = p;
  while (rem >= NB - bi) { rem -= NB - bi; ++bi; }
  const int bj = bi + rem;
  double s = 0.0;
  const double* src = part + (long)p * 256 + q;
  const long st = (long)NP * 256;
  int b = 0;
  for (; b + 8 <= nblocks; b += 8) {                 // eight loads in flight, the additions in block order
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = src[(b + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u];
  }
  for (; b < nblocks; ++b) s += src[b * st];
  const int i = 16 * bi + 4 * (l >> 4) + e, j = 16 * bj + (l & 15);
  // (a diagonal block holds both triangles, summed in different orders: the upper one is kept, so that G is symmetric to the bit)
  if (i < ncol && j < ncol && (bi != bj || i <= j)) {
    G[(long)i * ncol + j] = s;
    G[(long)j * ncol + i] = s;
  }
}

template <int NB>
static int launch_gram_wide(int nt, const float* U, const float* V, const float* d, const float* v, const float* h, long N, int r,
                            double* part, int grid, hipStream_t st) {
  const long nfull = N / kGwRows;                      // whole tiles: `grid` workgroups; the partial last tile: one more, its partial behind
  if (nfull > 0) {
    if (nt) hipLaunchKernelGGL((k_gram_wide<NB, true, false>), dim3(grid), dim3(kGwThreads), 0, st, U, V, d, v, h, N, r, nfull, 0L, part);
    else hipLaunchKernelGGL((k_gram_wide<NB, false, false>), dim3(grid), dim3(kGwThreads), 0, st, U, V, d, v, h, N, r, nfull, 0L, part);
    if (hipGetLastError() != hipSuccess) return 1;
  }
  if (N % kGwRows) {
    double* tail = part + (long)(nfull > 0 ? grid : 0) * (NB * (NB + 1) / 2) * 256;
    hipLaunchKernelGGL((k_gram_wide<NB, false, true>), dim3(1), dim3(kGwThreads), 0, st, U, V, d, v, h, N, r, nfull + 1, nfull, tail);
  }
  return (int)hipGetLastError();
}

}  // namespace psgd

using namespace psgd;

static int gram_wide_grid(int64_t N) {
  const int64_t tiles = N / kGwRows;
  int64_t g = (int64_t)device_cus();                // one 16-wave workgroup per CU
  if (g > tiles) g = tiles;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" {

int64_t psgd_uvd_gram_wide_scratch_bytes(int64_t N, int r) {
  if (N <= 0 || r <= PSGD_UVD_MAX_RANK || r > 2 * PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  const int NB = (2 * r + 2 + 15) / 16;
  return (int64_t)(gram_wide_grid(N) + 1) * (NB * (NB + 1) / 2) * 256 * 8;      // (+ 1: the partial last tile's workgroup)
}

/* G [2r + 2][2r + 2] (fp64, row-major, symmetric) = W'W for W = [U | V | d .* h | v ./ d], 32 < r <= 64, in one sweep over U and V
 * (contiguous [N, r] fp32).  scratch: psgd_uvd_gram_wide_scratch_bytes(N, r) bytes, 256-aligned. */
int psgd_uvd_gram_wide_f32(const float* U, const float* V, const float* d, const float* v, const float* h, int64_t N, int r,
                           double* G, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!U || !V || !d || !v || !h || !G || N <= 0) return PSGD_ERR_BAD_ARG;
  if (r <= PSGD_UVD_MAX_RANK || r > 2 * PSGD_UVD_MAX_RANK) return PSGD_ERR_RANK;
  const int64_t need = psgd_uvd_gram_wide_scratch_bytes(N, r);
  if (!scratch || (reinterpret_cast<uintptr_t>(scratch) & 255) || scratch_bytes < need) return PSGD_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int NB = (2 * r + 2 + 15) / 16, grid = gram_wide_grid(N);
  static const int env_nt = getenv("PSGD_GW_NT") ? atoi(getenv("PSGD_GW_NT")) : -1;      // (A/B runs)
  const int nt = env_nt >= 0 ? env_nt : policy_nt(2 * N * r * 4);
  double* part = static_cast<double*>(scratch);
  int e = 1;
  switch (NB) {
    case 5: e = launch_gram_wide<5>(nt, U, V, d, v, h, N, r, part, grid, st); break;
    case 6: e = launch_gram_wide<6>(nt, U, V, d, v, h, N, r, part, grid, st); break;
    case 7: e = launch_gram_wide<7>(nt, U, V, d, v, h, N, r, part, grid, st); break;
    case 8: e = launch_gram_wide<8>(nt, U, V, d, v, h, N, r, part, grid, st); break;
    case 9: e = launch_gram_wide<9>(nt, U, V, d, v, h, N, r, part, grid, st); break;
    default: return PSGD_ERR_RANK;
  }
  if (e) return PSGD_ERR_LAUNCH;
  const int NP = NB * (NB + 1) / 2, ncol = 2 * r + 2;
  const int nsets = (N / kGwRows > 0 ? grid : 0) + (N % kGwRows ? 1 : 0);
  hipLaunchKernelGGL(k_gram_wide_finish, dim3((NP * 256 + kThreads - 1) / kThreads), dim3(kThreads), 0, st, part, nsets, NB, ncol, G);
  return hipGetLastError() == hipSuccess ? PSGD_OK : PSGD_ERR_LAUNCH;
}

}  // extern "C"
