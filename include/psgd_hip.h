/*
 * psgd_hip.h -- C ABI of the MI355X (gfx950) PSGD preconditioner engine.
 *
 * This is the drop-in boundary for the hot path of the reference module
 * preconditioned_stochastic_gradient_descent.py ("psgd.py" below).  The
 * reference has no FFI of its own (it is pure Python on TensorFlow), so each
 * entry point cites the reference *function* (file:line) whose arithmetic it
 * replaces.  The Python module of the same name in this repository binds these
 * symbols through ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Every data pointer is a DEVICE pointer (HBM) unless marked [host].
 *   - Matrices are row-major fp32.  U, V are [N, r] contiguous (leading
 *     dimension r), vectors are [N] (the reference's [N,1] columns).
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on
 *     it and performs no host synchronisation and no allocation.
 *   - Return value: PSGD_OK (0) or a negative PSGD_ERR_* code.  Nothing is
 *     written when an argument check fails.
 *   - `ws` is a caller-owned device workspace of at least
 *     psgd_uvd_workspace_bytes(N, r) bytes, 256-byte aligned.  Its contents
 *     carry the small reduced vectors between the stages of one call.
 *
 * Multi-GPU (row-sharded flat parameter vector, one process per GPU): call the
 * *_sweepK stage functions on the local shard and all-reduce the region that
 * psgd_uvd_ws_region() reports between stages (SUM on the fp64 sums, MAX on
 * the fp32 max buffer); pass sums_reduced=1 to the following stage.
 */
#ifndef PSGD_HIP_H
#define PSGD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* History: 1 = round 1; 2 = round 2 (psgd_uvd_fused_s1_f32 removed, SUMS region of stage 13 holds 4r entries,
 * workspace layout changed); 3 = round 3 (entry points added, bf16 apply workspace carries a hand-off route word);
 * 4 = round 4 (Kron dense (x) dense workspaces of small layers carry the scratch of the fused strip kernels);
 * 5 = round 4, late (the *_ld row-stride forms of the wide-rank building blocks and psgd_kron_dd_apply_direct_f32 added; Kron tuning
 * keys retired / renumbered -- bf16 key 4 now selects the stream-K gradient launches, not the DMA position; the workspace of
 * psgd_kron_dd_update_bf16 gained the stream-K partial tiles (up to 128 MiB more for M, N multiples of 256) and padded W1 / W2
 * row strides: always size it with psgd_kron_dd_update_workspace_bytes_bf16);
 * 6 = round 5 (psgd_uvd_apply_cols_f32 added: precond_grad_UVd_math on a matrix g; psgd_uvd_gram_wide_f32 and the psgd_uvd_wide_*
 * entry points of ranks 33 .. 64 added (building blocks, update, fused update -> apply); the sparse-LU entry points take ranks up to PSGD_SPLU_MAX_RANK = 64 and their workspace
 * regions have that capacity -- offsets from psgd_splu_ws_region changed; the fused strip kernels of small Kron layers and their
 * tuning key 21 removed: Kron workspaces of small layers shrink back by that scratch);
 * 7 = round 6 (psgd_kron_bf16_handoff_counter_offset added; bf16 tuning key 7: XCD patch of the fused pair; no layout change).
 * psgd_tf_amd/_lib.py refuses a library whose psgd_abi_version() differs from the one it was written for. */
#define PSGD_ABI_VERSION 7

#define PSGD_OK                 0
#define PSGD_ERR_BAD_ARG       (-1)   /* null pointer, N <= 0, r <= 0 ...            */
#define PSGD_ERR_RANK          (-2)   /* rank r outside the entry point's range      */
#define PSGD_ERR_WORKSPACE     (-3)   /* workspace missing, too small or misaligned  */
#define PSGD_ERR_ALIGN         (-4)   /* a matrix pointer is not 16-byte aligned     */
#define PSGD_ERR_LAUNCH        (-5)   /* HIP reported a launch error                 */
#define PSGD_ERR_SHAPE         (-6)   /* Kron: shapes inconsistent / unsupported     */

#define PSGD_UVD_MAX_RANK 32
#define PSGD_SPLU_MAX_RANK 64   /* the sparse-LU entry points (round 5: the tail kernels are instantiated for 1 .. 64) */

/* workspace regions that the multi-GPU driver exchanges between stages.  Two equivalent protocols:
 *  (a) all-gather + fold (what psgd_tf_amd/sharded.py does): all-gather every rank's PSGD_WS_SEND_F64 region of the
 *      stage into a [world][count] fp64 buffer and call psgd_*_fold_gathered_f64, which folds the copies in rank
 *      order (SUM or MAX per entry).  One collective per exchange point, and the reduced values are bit-identical on
 *      every rank whatever algorithm the collective library picks.
 *  (b) all-reduce in place: SUM on the PSGD_WS_SUMS_F64 region and MAX on the PSGD_WS_MAX_F32 region of the stage. */
#define PSGD_WS_SUMS_F64   0   /* double[len]: SUM over ranks  */
#define PSGD_WS_MAX_F32    1   /* float[len]:  MAX over ranks  */
#define PSGD_WS_SEND_F64   2   /* double[len]: this rank's contribution, to be all-gathered and folded */

int         psgd_abi_version(void);
const char *psgd_error_string(int code);

/* ------------------------------------------------------------------ UVd ---
 * Q = (I + U V') diag(d), preconditioner P = Q'Q  (psgd.py:527-627).        */

/* Bytes of device workspace needed for a shard of N rows at rank r
 * (includes the N-float temporary for nablaD, psgd.py:581). <0 on error.   */
int64_t psgd_uvd_workspace_bytes(int64_t N, int r);

/* Byte offset (into ws) and element count of a region to all-reduce.
 * `which` = PSGD_WS_SUMS_F64 / PSGD_WS_MAX_F32; `stage` selects what the next
 * stage consumes: apply: 1 (after sweep1: r sums), 2 (after sweep2: r sums);
 * update: 10 (balance maxima: 2 floats), 11 (after sweep1: Gram sums),
 * 12 (after sweep2: 1 float max), 13 (after the fused sweep2: the 4r sums [pU | pV | qU | qV]).
 * Returns 0 or an error code.                                                */
int psgd_uvd_ws_region(int which, int stage, int64_t N, int r,
                       int64_t *offset_bytes, int64_t *count);
/* Protocol (a): PSGD_WS_SEND_F64 regions exist for stages 1, 2, 11 (sums), 10 (2 maxima), 12 (1 maximum) and
 * 13 ([pU | pV | qU | qV | max]: the 4r sums and the maximum of the fused sweep 2 in ONE region, so the fused step
 * needs two exchanges: 11, 13).  `gathered` = the all-gathered regions, [world][count] doubles in rank order (device).   */
int psgd_uvd_fold_gathered_f64(int stage, const double *gathered, int world, int64_t N, int r,
                               void *ws, int64_t ws_bytes, void *stream);

/* precond_grad_UVd_math(U, V, d, g)   psgd.py:619-627 (IpUVtmatvec :540-544)
 *   out = d .* (I + V U') (I + U V') (d .* g)
 * Three streaming sweeps (the second reduction depends on the first); V is read twice and
 * U once:
 *   sweep1: s1 = V'(d.*g)                          reads V,d,g
 *   sweep2: g1 = d.*g + U s1 -> out;  s2 = U'g1    reads U,d,g   writes out (= g1)
 *   sweep3: out = d.*(g1 + V s2)  in place         reads V,d,out writes out
 * `out` must not alias any input; between sweep2 and sweep3 it holds g1.          */
int psgd_uvd_apply_f32(const float *U, const float *V, const float *d,
                       const float *g, float *out, int64_t N, int r,
                       void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_apply_sweep1_f32(const float *V, const float *d, const float *g,
                              int64_t N, int r, void *ws, int64_t ws_bytes,
                              void *stream);
int psgd_uvd_apply_sweep2_f32(const float *U, const float *d, const float *g,
                              float *out, int64_t N, int r, int sums_reduced,
                              void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_apply_sweep3_f32(const float *V, const float *d, float *out,
                              int64_t N, int r, int sums_reduced,
                              void *ws, int64_t ws_bytes, void *stream);

/* update_precond_UVd_math_(U, V, d, v, h, step, tiny)   psgd.py:554-617
 * Mutates U or V (branch update_U, psgd.py:588) and d in place.
 *   balance  : the branch of psgd.py:562-567 (rescale U/rho, rho*V)
 *   update_U : 1 -> psgd.py:589-601, 0 -> psgd.py:603-615
 * Stages: [balance_max, balance_scale]  sweep1 (Gram of [U V t w])
 *         sweep2 (r x r solves, row-local update of U or V, nablaD, max)
 *         sweep3 (d <- d - mu d nablaD).                                    */
int psgd_uvd_update_f32(float *U, float *V, float *d, const float *v,
                        const float *h, int64_t N, int r, float step,
                        float tiny, int balance, int update_U,
                        void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_balance_max_f32(const float *U, const float *V, int64_t N, int r,
                             void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_balance_scale_f32(float *U, float *V, int64_t N, int r,
                               void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_update_sweep1_f32(const float *U, const float *V, const float *d,
                               const float *v, const float *h, int64_t N,
                               int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_update_sweep2_f32(float *U, float *V, const float *d,
                               const float *v, const float *h, int64_t N,
                               int r, float step, float tiny, int update_U,
                               void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_update_sweep3_f32(float *d, int64_t N, int r, float step,
                               float tiny, void *ws, int64_t ws_bytes,
                               void *stream);

/* Fused update -> apply (the UVd.step call pattern, psgd.py:732 -> :748): THREE sweeps for both calls.
 * Same U, V, d as psgd_uvd_update_f32; `out` = precond_grad_UVd_math on the updated state.  Update sweep 2 also
 * reduces the four r-vectors [Unew | Vnew]' [d.*g, d.*g.*nablaD] (on the matrix core).  With dnew = d - mu_d d.*nablaD
 * (psgd.py:584) both reductions of the apply follow from them and from the Gram of sweep 1:
 *     s1' = Vnew'(dnew.*g) = pV - mu_d qV,      s2' = Unew'(dnew.*g + Unew s1') = pU - mu_d qU + (Unew'Unew) s1'
 * (Unew'Unew = U'U plus a rank-2 correction known from the r x r algebra), so the d update and the whole apply are
 * ONE last sweep: out = dnew .* (dnew.*g + Unew s1' + Vnew s2').  628 -> 612 bytes per parameter at r = 20 and 9 -> 6 launches.
 * Multi-GPU stages: [balance] -> update_sweep1 -> X(11) -> update_sweep2_fused -> X(13: 4r sums | max) ->
 * fused_post -> fused_final: two exchanges per step. */
int psgd_uvd_update_apply_f32(float *U, float *V, float *d, const float *v, const float *h,
                              const float *g, float *out, int64_t N, int r, float step, float tiny,
                              int balance, int update_U, void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_update_sweep2_fused_f32(float *U, float *V, const float *d, const float *v,
                                     const float *h, const float *g, int64_t N, int r, float step,
                                     float tiny, int update_U, void *ws, int64_t ws_bytes,
                                     void *stream);
int psgd_uvd_fused_post_f32(int64_t N, int r, float step, float tiny, int update_U, void *ws,
                            int64_t ws_bytes, void *stream);
int psgd_uvd_fused_final_f32(const float *U, const float *V, float *d, const float *g, float *out,
                             int64_t N, int r, float step, float tiny, void *ws, int64_t ws_bytes,
                             void *stream);

/* IpUVtmatvec(U, V, x)   psgd.py:540-544:  out = x + U (V' x), x is [N].   */
int psgd_uvd_ipuvt_matvec_f32(const float *U, const float *V, const float *x,
                              float *out, int64_t N, int r,
                              void *ws, int64_t ws_bytes, void *stream);

/* IpUVtmatvec with a matrix x (psgd.py:542 "matrices or column vectors"): k columns, given as HOST arrays of k device
 * pointers to contiguous [N] vectors (xs in, outs out; outs[j] may not alias any input).  One sweep of V and one of U
 * per group of four columns.                                                                                       */
int psgd_uvd_ipuvt_matvec_cols_f32(const float *U, const float *V, const float *const *xs,
                                   float *const *outs, int k, int64_t N, int r,
                                   void *ws, int64_t ws_bytes, void *stream);

/* precond_grad_UVd_math(U, V, d, g) with a MATRIX g (psgd.py:619-627; docstring :623 "either matrices or column vectors": d
 * broadcasts over the columns, psgd.py:625-626): k columns as HOST arrays of k device pointers to contiguous [N] vectors (gs in,
 * outs out; outs[j] may be gs[j], otherwise no aliasing).  Per group of four columns: S1 = V'(d.*G), G1 = d.*G + U S1 with
 * S2 = U'G1 from the same sweep, out = d.*(G1 + V S2) -- three sweeps (V, U, V) per FOUR columns.                      */
int psgd_uvd_apply_cols_f32(const float *U, const float *V, const float *d, const float *const *gs,
                            float *const *outs, int k, int64_t N, int r,
                            void *ws, int64_t ws_bytes, void *stream);

/* The Gram of update_precond_UVd_math_ (psgd.py:569-615: every inner product of the columns of W = [U | V | d .* h | v ./ d]) for
 * ranks PSGD_UVD_MAX_RANK < r <= 2 PSGD_UVD_MAX_RANK in ONE sweep over U and V (round 5; the chunked wide-rank path needed three):
 * G [2r + 2][2r + 2], fp64, row-major, symmetric.  U, V contiguous [N, r] (16-byte aligned when r % 4 == 0); scratch:
 * psgd_uvd_gram_wide_scratch_bytes(N, r) bytes, 256-byte aligned, contents need not survive.                                      */
int64_t psgd_uvd_gram_wide_scratch_bytes(int64_t N, int r);
int psgd_uvd_gram_wide_f32(const float *U, const float *V, const float *d, const float *v, const float *h,
                           int64_t N, int r, double *G, void *scratch, int64_t scratch_bytes, void *stream);

/* Ranks PSGD_UVD_MAX_RANK < r <= 2 PSGD_UVD_MAX_RANK on the whole contiguous [N, r] matrices (round 5; 16-byte aligned): the three
 * sweeps of precond_grad_UVd_math on k columns (k = 1: the column-vector call), and the four-column building blocks of the update's
 * row-local part -- S = M'[x_0 ..] (fp64 [k][r]), out_j = x_j + M S_j (S fp32 [k][r]), M <- M - (a c1' - b c2') (c = [c1 | c2] fp32).
 * scratch: psgd_uvd_wide_scratch_bytes(N, r) bytes, 256-byte aligned, contents need not survive.  (Ranks above 64, and strided
 * views, stay on the column-chunk blocks below.)                                                                                     */
int64_t psgd_uvd_wide_scratch_bytes(int64_t N, int r);
int psgd_uvd_wide_apply_cols_f32(const float *U, const float *V, const float *d, const float *const *gs, float *const *outs,
                                 int k, int64_t N, int r, void *scratch, int64_t scratch_bytes, void *stream);
int psgd_uvd_wide_colsums_f32(const float *M, const float *const *xs, int k, double *S, int64_t N, int r,
                              void *scratch, int64_t scratch_bytes, void *stream);
int psgd_uvd_wide_axpy_cols_f32(const float *M, const float *const *xs, float *const *outs, int k, const float *S,
                                int64_t N, int r, void *stream);
int psgd_uvd_wide_rank2_update_f32(float *M, const float *a, const float *b, const float *c, int64_t N, int r, void *stream);

/* update_precond_UVd_math_ (psgd.py:554-617 without the balancing of :562-567, which stays with the caller) for
 * PSGD_UVD_MAX_RANK < r <= 2 PSGD_UVD_MAX_RANK on one GPU, in place on U or V (update_U) and d: the Gram of
 * psgd_uvd_gram_wide_f32, the r x r algebra in one workgroup (fp64, partial pivoting), one sweep that reads U and V and
 * writes the updated factor and nablaD, and the update of d -- 5r + 10 floats per row, the traffic of the specialised ranks.
 * U, V contiguous [N, r], 16-byte aligned, d 16-byte aligned; scratch: psgd_uvd_wide_update_scratch_bytes(N, r) bytes
 * (about 4 N), 256-byte aligned, contents need not survive.                                                              */
int64_t psgd_uvd_wide_update_scratch_bytes(int64_t N, int r);
/* ... followed by precond_grad_UVd_math on the updated state (the UVd.step pattern, psgd.py:732 -> :748; what
 * psgd_uvd_update_apply_f32 is for r <= 32): sweep 2 also reduces [Unew | Vnew]' [d.*g, d.*g.*nablaD], one small kernel turns those
 * sums and the Gram into the two r-vectors of the apply, one last sweep updates d and writes out -- U and V are read three times.
 * g, out: [N], out must not alias g.  Same results as the two calls within rounding (tests: 2e-5).                              */
int64_t psgd_uvd_wide_update_apply_scratch_bytes(int64_t N, int r);
int psgd_uvd_wide_update_apply_f32(float *U, float *V, float *d, const float *v, const float *h, const float *g, float *out,
                                   int64_t N, int r, float step, float tiny, int update_U,
                                   void *scratch, int64_t scratch_bytes, void *stream);
int psgd_uvd_wide_update_f32(float *U, float *V, float *d, const float *v, const float *h, int64_t N, int r,
                             float step, float tiny, int update_U, void *scratch, int64_t scratch_bytes, void *stream);

/* Building blocks of the wide-rank path.  The sweep kernels are instantiated for ranks 1..PSGD_UVD_MAX_RANK; a
 * preconditioner of larger rank (the reference has no limit, psgd.py:663) is handled one level up
 * (psgd_tf_amd/uvd_wide.py) on column chunks of U and V -- each chunk a contiguous [N, rc] matrix, rc <= 32 -- with the
 * Gram sweep (psgd_uvd_update_sweep1_f32 on pairs of chunks) and these three calls.  xs / outs: HOST arrays of k device
 * pointers to contiguous [N] vectors; S, c: DEVICE arrays.
 *   colsums:    S[j][0..r) = M' x_j                (fp64)
 *   axpy_cols:  out_j = x_j + M S_j                (S fp32 [k][r]; out_j may be x_j itself)
 *   rank2:      M <- M - (a c1' - b c2')           (c = [c1 | c2] fp32; psgd.py:600-601 / :614-615 with mu folded in)  */
int psgd_uvd_colsums_f32(const float *M, const float *const *xs, int k, double *S, int64_t N, int r,
                         void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_axpy_cols_f32(const float *M, const float *const *xs, float *const *outs, int k,
                           const float *S, int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_rank2_update_f32(float *M, const float *a, const float *b, const float *c, int64_t N, int r,
                              void *ws, int64_t ws_bytes, void *stream);

/* The same building blocks on a strided VIEW (round 4): M (U, V) is an [N, r] column block of a wider row-major matrix, element
 * (row, c) at M[row * ld + c], ld >= r in floats -- psgd_tf_amd/uvd_wide.py and splu_wide.py pass column views of U, V, L2
 * instead of copies of them (SURVEY 8b's `ldU, ldV`).  ld and the byte address of the view must be multiples of the rank's access
 * width (r % 4 == 0: 16 bytes, r % 2 == 0: 8, odd r: 4), else PSGD_ERR_ALIGN (the callers then copy that chunk).  ld == r is the
 * contiguous call.  psgd_uvd_update_sweep1_ld_f32 is the Gram sweep (stage 11) of a pair of such views. */
int psgd_uvd_colsums_ld_f32(const float *M, int64_t ld, const float *const *xs, int k, double *S, int64_t N, int r,
                            void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_axpy_cols_ld_f32(const float *M, int64_t ld, const float *const *xs, float *const *outs, int k,
                              const float *S, int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_rank2_update_ld_f32(float *M, int64_t ld, const float *a, const float *b, const float *c, int64_t N, int r,
                                 void *ws, int64_t ws_bytes, void *stream);
int psgd_uvd_update_sweep1_ld_f32(const float *U, int64_t ldU, const float *V, int64_t ldV, const float *d,
                                  const float *v, const float *h, int64_t N, int r, void *ws, int64_t ws_bytes,
                                  void *stream);

/* Tuning knobs for experiments (not part of the stable ABI).
 * key 0: streaming policy (0 = automatic: non-temporal when U,V exceed the Infinity Cache,
 *        1 = never non-temporal, 2 = always non-temporal).
 * key 1: cap on blocks per CU for the sweeps (0 = occupancy query).
 * key 2: r x r algebra of the update: 0 (default) one-row-per-lane register kernel, 1 = block-cooperative reference.
 * key 3: short sweeps: tiles a wave should stream at least (default 8; grids shrink to that, never below one block per CU). */
int psgd_set_tuning(int key, int value);

/* Live kernel timing for bench.py (measurement aid, not part of the reference's surface).
 * While enabled, every sweep-kernel launch is bracketed by HIP events recorded on the launch
 * stream.  psgd_prof_collect() waits for the recorded events of `slot`, returns the summed
 * kernel time and the launch count since the last collect, and clears the slot.
 * It is the only call here that synchronises with the device.                            */
#define PSGD_PROF_APPLY_S1   0
#define PSGD_PROF_APPLY_S2   1
#define PSGD_PROF_APPLY_S3   2
#define PSGD_PROF_UPDATE_S1  3
#define PSGD_PROF_UPDATE_S2  4
#define PSGD_PROF_UPDATE_S3  5
int psgd_prof_enable(int on);
int psgd_prof_collect(int slot, double *total_ms, int *count);

/* ------------------------------------------------------------ sparse LU ---
 * P = Q'Q, Q = L U, L = [L1 0; L2 diag(l3)], U = [U1 U2; 0 diag(u3)]  (psgd.py:396-404).
 * Layout is the reference's: L12 = [L1; L2] is [N, r] row-major (L1 lower triangular), U12 = [U1, U2]
 * is [r, N] row-major (U1 upper triangular), l3 and u3 are [N - r].  1 <= r <= PSGD_SPLU_MAX_RANK
 * (64 since round 5: ranks above 32 take 64-row tiles, one workgroup per CU, and about 0.6 of the rate of the ranks below),
 * N >= r.  L12 (and L12_new) must be 16-byte aligned; vectors need 4-byte alignment only.
 *
 * psgd_splu_apply_f32  replaces precond_grad_splu(L12, l3, U12, u3, grads)   psgd.py:483-524
 *   g, out: the concatenated gradient / preconditioned gradient, [N] (the list <-> flat vector plumbing of
 *   :495-497,:518-522 stays on the host side).  out must not alias g.
 * psgd_splu_update_f32 replaces update_precond_splu(L12, l3, U12, u3, dxs, dgs, step)   psgd.py:396-480
 *   dx, dg: the concatenated perturbations [N].  Pure like the reference: the new factors go to the *_new
 *   buffers (which may alias the inputs for an in-place update).  tiny = psgd.py:22 `_tiny`.          */
int64_t psgd_splu_workspace_bytes(int64_t N, int r);
int psgd_splu_apply_f32(const float *L12, const float *l3, const float *U12, const float *u3, const float *g,
                        float *out, int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_splu_update_f32(const float *L12, const float *l3, const float *U12, const float *u3, const float *dx,
                         const float *dg, float *L12_new, float *l3_new, float *U12_new, float *u3_new, int64_t N,
                         int r, float step, float tiny, void *ws, int64_t ws_bytes, void *stream);

/* Stage forms of the two calls above, for tail rows sharded over several GPUs (one process per GPU): every rank passes
 * its local tensors -- L12 = [L1; its rows of L2] ([r + n_local, r]), U12 = [U1, its columns of U2], its slices of
 * l3 / u3, and flat vectors [the r corner entries (replicated); its slice of the tail] -- with N = r + n_local, and
 * all-reduces the workspace region psgd_splu_ws_region(which, stage) reports after stage `stage`:
 *   which 0 (fp64, SUM): stage 1: r sums (U2 x2); stage 2: 2r sums (apply uses the first r); stage 3: r sums (update)
 *   which 1 (fp32, MAX): stage 3: 4 maxima (update)
 * Every rank then redoes the r x r corner algebra on identical inputs and gets identical corner results.
 * psgd_splu_stage1_f32 serves both paths (x = the flat gradient for the apply, dg for the update).
 * has_tail (update stage 4): the GLOBAL problem has tail rows (N_global > r).                                       */
int psgd_splu_ws_region(int which, int stage, int64_t N, int r, int64_t *offset_bytes, int64_t *count);
/* all-gather + fold form (protocol (a) above): PSGD_WS_SEND_F64 regions for stage 1 (r sums), 2 (2r sums) and
 * 3 ([r sums | 4 maxima] in one region: three exchanges per update, two per apply).                                */
int psgd_splu_fold_gathered_f64(int stage, const double *gathered, int world, int64_t N, int r, void *ws,
                                int64_t ws_bytes, void *stream);
int psgd_splu_stage1_f32(const float *U12, const float *x, int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_splu_apply_stage2_f32(const float *L12, const float *l3, const float *U12, const float *u3, const float *g,
                               float *out, int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_splu_apply_stage3_f32(const float *L12, const float *l3, const float *U12, const float *u3, float *out,
                               int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_splu_update_stage2_f32(const float *L12, const float *l3, const float *U12, const float *u3, const float *dx,
                                const float *dg, int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_splu_update_stage3_f32(const float *L12, const float *l3, const float *U12, const float *u3, const float *dx,
                                const float *dg, int64_t N, int r, void *ws, int64_t ws_bytes, void *stream);
int psgd_splu_update_stage4_f32(const float *L12, const float *l3, const float *U12, const float *u3, const float *dx,
                                const float *dg, float *L12_new, float *l3_new, float *U12_new, float *u3_new, int64_t N,
                                int r, float step, float tiny, int has_tail, void *ws, int64_t ws_bytes, void *stream);

/* ----------------------------------------------------------------- Kron ---
 * P = kron(Qr'Qr, Ql'Ql) with dense upper-triangular Ql [M,M], Qr [N,N]
 * (psgd.py:156-192).  G, dX, dG are [M,N] row-major.                       */

int64_t psgd_kron_dd_workspace_bytes(int M, int N);

/* Experiment knobs (not stable ABI; defaults in brackets).  Settled A/B keys were frozen into constants in round 4 (5, 8, 10, 13-15,
 * 17, 19, 20).  Keys 1, 4, 12, 16 change what prepared state in a workspace means: prepare again after changing them.
 *  0 fp32 GEMM tile: [0] auto, 1 = 64, 2 = 128, 3 = 32        1 [1] 128-tile products as fp32-accurate bf16 x 3 GEMMs, 0 = exact fp32 MFMA
 *  2 solve strips: [0] register-resident, 1 = LDS-resident      3 32-tile products: [1] k_gemm_small (ring of 4 K tiles), 0 = generic body
 *  4 [1] large products on operands split ONCE into planes (k_split3 / k_gemm_p3), 0 = split inside every GEMM tile
 *  6 [1] K split of the tail of the gradient grid (large fp32 update, M = N), 0 = whole tiles
 *  7 bits [3]: 1 = batched small update shares launches between product and solve stages; 2 = single small updates as a batch of one
 *  9 [1] large updates fork the products of psgd.py:173 onto a side stream (events only: capturable), 0 = one stream
 * 11 [1] M, N >= 2048: the solves of :174 as products with explicit inverses of diagonal blocks, 0 = substitution strips
 * 12 plane format: [2] f16 x 2 (x 2^e = h + 2^-11 M) in the large apply and update, 1 = apply only, 0 = bf16 x 3 everywhere
 * 16 [1] chained products write fp32 + max|C| and a split launch makes exact-scale planes, 0 = epilogue planes at a bound scale
 * 23 [1] 512-blocks of an inverse from one strip launch (k_tri_inv512), 0 = k_tri_inv128 + doubling levels 128, 256
 * 24 [2048] block size h of the blocked solves on inverses of diagonal h-blocks, 0 = whole inverses, one product per solve;
 *    a value that would make more than 4 block columns per side (ceil(max(M, N) / h) > 4) is replaced by 2048 for that call
 * 25 [-1] stream order of the inverse route: by shape (both inversions first from 4096^2 on), 0 / 1 = products / inversions first
 * 27 [1] the factor updates (:179) of the large update walk their tiles in 4 x 4 patches per XCD (from 32 x 32 tiles on), 0 = whole tile rows
 * 28 [1] chained f16 x 2 plane products write their planes from the epilogue at TILE scales (each 128 x 128 tile at its own maximum; the
 *    consumer's K loop shifts its accumulators when the scale changes), 0 = fp32 out + max|C| + a split launch (one scale per matrix)
 * 29 [1] large fp32 update with both inversions first: dX's planes on the side stream ahead of Ql's inversion, 0 = on the caller's stream
 * 30 [-1] (round 6) large fp32 update: 1 = the products of psgd.py:173 on a third stream from the fork point on (beside both inversions),
 *      0 = behind Ql's inversion on the side stream, -1 = 1 when both factors reach 4096
 * 31 [1] (round 6) large fp32 update on the tile-scale inverse route: the prologue is rho + ONE sweep (balanced upper tiles in fp32, both
 *      plane forms of both factors at tile scales, the inverted 32-blocks) and dX's / dG's planes are one sweep each at tile scales;
 *      0 = the round-5 prologue (two sweeps, one scale per factor; max|dX|, max|dG| launches ahead of their splits)
 * 32 [1] (round 6, shapes below key 30's rule) dX's and dG's planes on the third stream, Ql's inversion from the fork point on
 * 33 [1] (round 6) M != N: every tile of the smaller factor's gradient is split along its K (the longer side) into chunks as long as the
 *      other gradient's tiles; 0 = whole tiles
 * 34 [1] (round 6) products of the blocked solves with at most 256 output tiles and >= 64 K steps: two workgroups per tile; 0 = one */
int psgd_kron_set_tuning(int key, int value);

/* _precond_grad_dense_dense(Ql, Qr, Grad)  psgd.py:182-192.
 * Association order follows the reference: M < N uses ((Ql'Ql) G) Qr' Qr,
 * otherwise Ql' (Ql (G (Qr'Qr))).  Ql, Qr are upper-triangular by the reference's
 * invariant (psgd.py:175-179); entries below the diagonal are not read.     */
int psgd_kron_dd_apply_f32(const float *Ql, const float *Qr, const float *G,
                           float *out, int M, int N, void *ws,
                           int64_t ws_bytes, void *stream);
/* The same call in two halves: what depends on the factors only (their Grams, kept in ws) and what depends on the
 * gradient.  psgd_kron_dd_apply_f32 == prepare + apply_prepared; a caller that applies unchanged factors to several
 * gradients prepares once.  For M, N <= 512 (every launch latency-bound) BOTH Grams are prepared and the gradient half
 * is two products, (Ql'Ql) G (Qr'Qr) -- the reference's association on one side, re-associated on the other.        */
int psgd_kron_dd_prepare_f32(const float *Ql, const float *Qr, int M, int N, void *ws,
                             int64_t ws_bytes, void *stream);
int psgd_kron_dd_apply_prepared_f32(const float *Ql, const float *Qr, const float *G, float *out,
                                    int M, int N, void *ws, int64_t ws_bytes, void *stream);
/* The apply for factors that are NEW on every call (the reference's pattern: an apply right after an update): no Gram, nothing
 * prepared -- out = Ql' (Ql ((G Qr') Qr)) as four chained triangular products where the f16 x 2 plane products apply (large
 * layers: 4096^2 1.01 -> 0.93 ms); the same as psgd_kron_dd_apply_f32 elsewhere.  Leaves no prepared state in ws:
 * psgd_kron_dd_apply_prepared_f32 needs psgd_kron_dd_prepare_f32 (or psgd_kron_dd_apply_f32) for these factors first.  */
int psgd_kron_dd_apply_direct_f32(const float *Ql, const float *Qr, const float *G, float *out, int M, int N, void *ws,
                                  int64_t ws_bytes, void *stream);
/* 1 when the call above is a path of its own for this shape under the current tuning, 0 when it is psgd_kron_dd_apply_f32 */
int psgd_kron_dd_apply_direct_distinct(int M, int N);

/* _update_precond_dense_dense(Ql, Qr, dX, dG, step)  psgd.py:156-179.
 * Pure: Ql, Qr are read, the new factors are written to QlOut, QrOut.      */
int psgd_kron_dd_update_f32(const float *Ql, const float *Qr, const float *dX,
                            const float *dG, float *QlOut, float *QrOut,
                            int M, int N, float step, float tiny, void *ws,
                            int64_t ws_bytes, void *stream);

/* Batched forms for networks of small layers (LeNet5: mnist_with_lenet5.py:12-16, where every
 * GEMM is launch-bound): the same stage of every layer runs in ONE launch.  Arrays are HOST
 * arrays of `count` device pointers / sizes; `ws` holds the per-layer workspaces back to back
 * (psgd_kron_dd_workspace_bytes_batched).  Results are identical to the per-layer calls.
 * The batched update requires M, N <= 512 for every layer (PSGD_ERR_SHAPE otherwise).       */
int64_t psgd_kron_dd_workspace_bytes_batched(const int *M, const int *N, int count);
int psgd_kron_dd_apply_batched_f32(const float *const *Ql, const float *const *Qr,
                                   const float *const *G, float *const *out,
                                   const int *M, const int *N, int count,
                                   void *ws, int64_t ws_bytes, void *stream);
int psgd_kron_dd_prepare_batched_f32(const float *const *Ql, const float *const *Qr, const int *M,
                                     const int *N, int count, void *ws, int64_t ws_bytes, void *stream);
int psgd_kron_dd_apply_prepared_batched_f32(const float *const *Ql, const float *const *Qr,
                                            const float *const *G, float *const *out,
                                            const int *M, const int *N, int count,
                                            void *ws, int64_t ws_bytes, void *stream);
int psgd_kron_dd_update_batched_f32(const float *const *Ql, const float *const *Qr,
                                    const float *const *dX, const float *const *dG,
                                    float *const *QlOut, float *const *QrOut,
                                    const int *M, const int *N, int count,
                                    float step, float tiny,
                                    void *ws, int64_t ws_bytes, void *stream);

/* Sparse Kronecker factors (psgd.py:198-391), canonical orientations; the dispatcher's mirrored
 * formats (psgd.py:86,102,104 / :128,144,146) call these on transposed views.
 *   fmt 0 "ds": (dense, scaling)          Ql [M,M], qr [N]          psgd.py:276-322
 *   fmt 1 "nd": (normalization, dense)    ql [2,M], Qr [N,N]        psgd.py:198-270
 *   fmt 2 "ns": (normalization, scaling)  ql [2,M], qr [N]          psgd.py:328-391
 * dX, dG, G are strided views: element (m,n) at p[m*rs + n*cs] (dX and dG share strides).
 * Factor outputs and `out` are contiguous ([M,M] / [2,M] / [N] / [N,N] / [M,N]).  Pure: inputs
 * are never written.                                                                        */
int64_t psgd_kron_sparse_workspace_bytes(int fmt, int M, int N);
int psgd_kron_ds_update_f32(const float *Ql, const float *qr, const float *dX, const float *dG,
                            int64_t xrs, int64_t xcs, float *QlOut, float *qrOut, int M, int N,
                            float step, float tiny, void *ws, int64_t ws_bytes, void *stream);
int psgd_kron_ds_apply_f32(const float *Ql, const float *qr, const float *G, int64_t grs,
                           int64_t gcs, float *out, int M, int N, void *ws, int64_t ws_bytes,
                           void *stream);
int psgd_kron_nd_update_f32(const float *ql, const float *Qr, const float *dX, const float *dG,
                            int64_t xrs, int64_t xcs, float *qlOut, float *QrOut, int M, int N,
                            float step, float tiny, void *ws, int64_t ws_bytes, void *stream);
int psgd_kron_nd_apply_f32(const float *ql, const float *Qr, const float *G, int64_t grs,
                           int64_t gcs, float *out, int M, int N, void *ws, int64_t ws_bytes,
                           void *stream);
int psgd_kron_ns_update_f32(const float *ql, const float *qr, const float *dX, const float *dG,
                            int64_t xrs, int64_t xcs, float *qlOut, float *qrOut, int M, int N,
                            float step, float tiny, void *ws, int64_t ws_bytes, void *stream);
int psgd_kron_ns_apply_f32(const float *ql, const float *qr, const float *G, int64_t grs,
                           int64_t gcs, float *out, int M, int N, void *ws, int64_t ws_bytes,
                           void *stream);

/* bf16-operand variant of _precond_grad_dense_dense (psgd.py:182-192) for Transformer-scale
 * matrices (BASELINE config 5).  Outside the reference's contract (its Kron API is pinned to
 * fp32, psgd.py:113-115): Ql, Qr are the fp32 master factors (rounded to bf16 per call), G and
 * out are bf16 [M,N]; every product runs on the bf16 matrix cores with fp32 accumulation and
 * bf16 intermediates, in the reference's association order.  M and N must be multiples of 8.
 * Upper-triangular Ql, Qr are assumed (entries below the diagonal are not read).            */
int64_t psgd_kron_dd_workspace_bytes_bf16(int M, int N);
/* Experiment knob (not stable ABI). key 0: bf16 GEMM variant: 0 auto (256^2 8-phase kernel for large dense
 * products, fused triangular pair when every 256^2 tile gets its own CU, 128^2 register-staged otherwise);
 * 1 128^2 register-staged everywhere; 2 128^2 LDS-DMA ring; 3 256^2 for every product; 4 auto without the fused pair.
 * key 1: 1 (default) two fused triangular pairs, (G Qr') Qr then Ql' (Ql .), where legal (M, N multiples of 256 and at
 *        least 16 tiles of 256^2; the Python side pads the apply to such shapes when that costs <= 20 %); 0 keep the Gram-first chain.
 * key 2: log2 of the hand-off poll bound of the fused pair (default 22 ~ 0.5 s; tests set 0 to provoke time-outs).
 * key 3: bf16-operand update: 1 (default) the trailing products of its two triangular solves keep the three leading
 *        terms of the bf16 x 3 split (h h' + h m' + m h': 2^-16 relative per product, below the bf16 rounding of the
 *        operands); 0 = all six terms.
 * key 4: bf16-operand update, the two gradient products (psgd.py:175-176): 1 (default) on the 256^2 8-phase loop as whole-tile
 *        rounds + a stream-K tail finished by a second launch (M = N multiples of 256 with at least one tile per CU: 4096^2
 *        502 -> 262 us); 0 = the 128^2 one-tile-per-workgroup kernel for every shape; 2 / 3 = the stream-K launches for every
 *        shape the kernel can take, with / without whole-tile rounds (tests).
 * key 5: bf16-operand update, the factor updates (:179): 1 (default) tiles in 4 x 4 patches per XCD, 0 = whole tile rows.
 * key 7: fused triangular pair of the apply: tile rows of the patch an XCD works on (4: 4 x 8 tiles per XCD at 4096^2, 12 operand panels per
 *        K step and L2 instead of 18); 0 = whole tile columns per XCD (rounds 1-5); -1 (default) = 8 from 32 tile rows on, 0 below (the
 *        only shapes where it was measured to gain).  Results are bitwise the same. */
int psgd_kron_bf16_set_tuning(int key, int value);
int psgd_kron_dd_apply_bf16(const float *Ql, const float *Qr, const void *G_bf16,
                            void *out_bf16, int M, int N, void *ws,
                            int64_t ws_bytes, void *stream);
/* The same call in two halves.  The bf16 copies of the factors (plain and transposed, kept in ws) change only when
 * the factors do -- after update_precond_kron -- so a caller that applies the same factors to several gradients
 * prepares once: psgd_kron_dd_apply_bf16 == psgd_kron_bf16_prepare_factors + psgd_kron_dd_apply_bf16_prepared.      */
int psgd_kron_bf16_prepare_factors(const float *Ql, const float *Qr, int M, int N, void *ws,
                                   int64_t ws_bytes, void *stream);
int psgd_kron_dd_apply_bf16_prepared(const void *G_bf16, void *out_bf16, int M, int N, void *ws,
                                     int64_t ws_bytes, void *stream);
/* bf16-operand variant of _update_precond_dense_dense (psgd.py:156-180), the update-side companion of the entry
 * point above (same contract: outside the reference's fp32-only API, M and N multiples of 8, upper-triangular fp32
 * master factors in, fp32 updated factors out; dX, dG are bf16 [M,N]).  fp32 throughout: the balance (:166-170), the
 * two triangular solves of :174, the max-norms and step sizes of :177-178 and the subtraction Q - step*grad*Q.  bf16
 * operands with fp32 accumulation: A = Ql dG Qr' (:173), the Grams of :175-176 and grad * Q (:179-180).            */
int64_t psgd_kron_dd_update_workspace_bytes_bf16(int M, int N);
int psgd_kron_dd_update_bf16(const float *Ql, const float *Qr, const void *dX_bf16, const void *dG_bf16,
                             float *QlOut, float *QrOut, int M, int N, float step, float tiny,
                             void *ws, int64_t ws_bytes, void *stream);
/* The fused triangular pairs of the bf16 apply hand tiles between workgroups inside one launch.  Their schedule
 * assumes every workgroup of the launch resident (checked against the CU count -- which cannot see other streams or
 * processes on the GPU); their RESULT does not: a consumer that has waited for a tile for its whole bound (~0.5 s)
 * produces that tile itself (same instructions, same inputs: the same bytes its owner would write), publishes it and
 * redoes its own accumulation from the start, so every call returns the undisturbed bits whatever the residency.  The
 * workspace counts such recoveries: psgd_kron_bf16_handoff_timeouts (synchronises: call it at a point where the host
 * waits anyway) returns how many happened on this workspace since it was created or reset -- a diagnostic (a device
 * that is persistently shared wastes up to the bound per recovery; psgd_kron_bf16_set_tuning(0, 4) selects the kernels
 * without in-launch hand-offs).  A fresh workspace must be reset once before its first use (the word lives in
 * caller-owned memory).                                                                                              */
int psgd_kron_bf16_handoff_timeouts(const void *ws, int M, int N);
int psgd_kron_bf16_handoff_reset(void *ws, int M, int N, void *stream);
/* Byte offset of that 32-bit recovery counter inside a workspace of shape (M, N) (round 6): a caller that wants to WATCH it without
 * synchronising copies the word to pinned host memory asynchronously now and then (psgd_tf_amd/kron.py does, and switches to the
 * hand-off-free kernels by itself after three recoveries).  < 0 on a bad shape.                                                     */
int64_t psgd_kron_bf16_handoff_counter_offset(int M, int N);
#ifdef __cplusplus
}
#endif
#endif /* PSGD_HIP_H */
