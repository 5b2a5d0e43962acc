// kron_shared.h -- pieces of the fp32 Kronecker engine (psgd_kron.hip) that the bf16-operand update
// (psgd_kron_bf16.hip) reuses unchanged: the factor balance and the blocked triangular solve stay fp32.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace psgdk {

// hipFuncSetAttribute (MaxDynamicSharedMemorySize above the 64 KiB default) is PER DEVICE: one flag per device ordinal, so that a
// process which drives a second GPU sets the attribute there too (a single process-wide flag made that launch fail).
struct DeviceOnce {
  unsigned long long mask = 0;
  int dev = 0;
  bool needed() { (void)hipGetDevice(&dev); return !((mask >> (dev & 63)) & 1ull); }
  void done() { mask |= 1ull << (dev & 63); }
};

// QlS = Ql * sqrt(max|Qr| / max|Ql|), QrS = Qr / that  (psgd.py:166-170).  0 on success.
int kron_balance(const float* Ql, const float* Qr, int M, int N, float* QlS, float* QrS, hipStream_t st,
                 float* scal = nullptr,    // scal: 64 scratch words to zero in the same launch (or null)
                 float* dinv = nullptr,    // dinv: (ceil(N/32) + ceil(M/32)) * 1024 floats: the same launch inverts the 32 x 32
                                           // diagonal blocks of QrS (first) and QlS (after them) for kron_trsm_ut(..., inv_ready)
                 void* inv_ws = nullptr);  // the workspace of kron_inv_solves_*: the launch also does kron_inv_prepare's zeroing and leaves
                                           // the partial maxima of QlS / QrS there (then kron_inv_solves_front(..., maxima_ready = true))

// (round 6) The same on the tile-scale inverse route as rho + ONE sweep: QlS / QrS (upper 128-tiles, and zeros in the lower tiles inside
// the diagonal 512-blocks: NOTHING else below the diagonals is written -- the caller's last product must write zeros there itself), the
// inverted 32-blocks, and the factors' column-form planes at tile scales straight into the workspace of kron_inv_solves_* (then
// kron_inv_solves_front(..., maxima_ready = true, planes_ready = true)).  kron_fused_prologue_on: shape rule and tuning key 31.
bool kron_fused_prologue_on(int M, int N);
int kron_balance_planes(const float* Ql, const float* Qr, int M, int N, float* QlS, float* QrS, hipStream_t st, float* dinv, void* inv_ws,
                        float* scal = nullptr);      // scal: 64 scratch words to zero in the first launch (or null)

// Solve y Q = x for nvec vectors (vector i at stride si, element j at stride sj; Q upper triangular [n][n]);
// dinv: scratch of ceil(n/32) * 1024 floats.  0 on success.
// lite: the trailing products of the blocked solve keep only the three leading terms of the bf16 x 3 split (2^-16 relative)
// inv_ready: dinv already holds the inverted diagonal blocks of Q (kron_balance made them)
int kron_trsm_ut(const float* Q, int n, const float* X, float* Y, int nvec, long si, long sj, float* dinv, hipStream_t st,
                 int lite = 0, bool inv_ready = false);

// The two solves of psgd.py:174 as products with explicit inverses of the balanced factors (psgd_kron.hip: tri_inverse; fp32-
// accurate f16 x 2 plane products), for callers that hold QlS / QrS / the inverted 32-blocks of kron_balance themselves:
//   Bt = QlS^-T X0 QrS^-1,  X0, X1 (scratch), Bt fp32 [M x N] row-major.
// kron_inv_solves_bytes: workspace (256-aligned), 0 when the route does not apply to the shape (kron_inv_route); kron_inv_solves_on:
// the shape rule and the tuning keys (11, 4, 12) say "use it".
// kron_inv_prepare on `main` BEFORE the fork; kron_inv_solves_front puts Qr's side and X1 on `main` and Ql's inversion on `side`
// (behind whatever the caller has queued there; side == main is fine); after the caller's join, kron_inv_solves_back makes Bt.
int64_t kron_inv_solves_bytes(int M, int N);
bool kron_inv_solves_on(int M, int N);
int kron_inv_prepare(void* ws, int M, int N, hipStream_t main);
int kron_inv_solves_front(const float* QlS, const float* QrS, const float* dinv_r, const float* dinv_l, const float* X0, float* X1,
                          float* Bt, int M, int N, void* ws, hipStream_t main, hipStream_t side,      // (Bt: scratch here)
                          hipEvent_t l_ready = nullptr,      // l_ready: recorded on `side` behind Ql's inversion (for callers that queue
                                                             // more work on `side` and let `main` wait for this point only)
                          bool maxima_ready = false,         // kron_balance(..., inv_ws) left the factors' partial maxima: no k_absmax launches
                          int x0_parts = 0,                  // > 0: kron_inv_part(ws)[0 .. x0_parts) hold the partial maxima of |X0| already
                          bool planes_ready = false,         // kron_balance_planes made the factors' planes (tile scales): no split launches;
                                                             // X0's planes are then one sweep at tile scales too (x0_parts is not used)
                          hipEvent_t x0_ready = nullptr,     // with x0_stream: X0's planes are made THERE (the caller has put whatever makes X0
                          hipStream_t x0_stream = nullptr);  // on that stream), x0_ready is recorded behind them and `main` waits for it before X1
float* kron_inv_part(void* ws, int M, int N);                // the array for X0's partial maxima (kron_inv_part_max() floats)
int kron_inv_part_max();
bool kron_inv_first(int M, int N);                           // the order rule (tuning key 25): both inversions ahead of the products of :173
int kron_inv_solves_back(const float* QlS, float* X1, float* Bt, int M, int N, void* ws, hipStream_t main, bool planes_ready = false);

// The update has two chains that meet only at the gradient products: the products dG QrS' -> QlS (.) (psgd.py:173) and
// the solves (:174).  kron_fork makes `side` (a default-priority stream kept per device and caller stream; tuning key 10) wait for
// everything already on `main`; the caller puts one chain on it and kron_join makes `main` wait for that chain.
// Event fork/join only, so it is legal inside a stream capture of `main`.  nullptr = no side stream: stay on `main`.
struct KronFork { hipStream_t side; hipEvent_t fork, join, mid, aux; hipStream_t bg; hipEvent_t bg_done; int bg_live; };     // mid: a point inside the side chain the caller's stream waits for
// bg: a second forked stream (kron_fork_bg makes it wait for the fork point; kron_join waits for it too once it has been used)
int kron_fork_bg(KronFork* f);                         // 0 on success: f->bg waits for the fork point, f->bg_live = 1
KronFork* kron_fork(hipStream_t main);
int kron_join(KronFork* f, hipStream_t main);          // 0 on success
bool kron_overlap_chains(int M, int N);                // tuning key 9 and the shape rule

// Joins on every exit path: once a fork succeeded, `main` waits for the side chain whether the function returns through
// join() or through an early error return -- the side stream must never be left writing the workspace behind the caller's
// back, and an unjoined stream would invalidate a capture of `main`.  The fork/join events are per (device, caller
// stream): the entry points that fork are single-threaded per stream (two host threads must not issue on one stream).
struct KronForkScope {
  KronFork* f;
  hipStream_t main;
  bool joined;
  KronForkScope(KronFork* f_, hipStream_t m) : f(f_), main(m), joined(false) {}
  KronForkScope(const KronForkScope&) = delete;
  KronForkScope& operator=(const KronForkScope&) = delete;
  int join() {
    joined = true;
    return f ? kron_join(f, main) : 0;
  }
  ~KronForkScope() {
    if (f && !joined) (void)kron_join(f, main);
  }
};

}  // namespace psgdk
