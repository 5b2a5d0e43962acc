// kron_shared.h -- pieces of the fp32 Kronecker engine (psgd_kron.hip) that the bf16-operand update
// (psgd_kron_bf16.hip) reuses unchanged: the factor balance and the blocked triangular solve stay fp32.
#pragma once
#include <hip/hip_runtime.h>

namespace psgdk {

// QlS = Ql * sqrt(max|Qr| / max|Ql|), QrS = Qr / that  (psgd.py:166-170).  0 on success.
int kron_balance(const float* Ql, const float* Qr, int M, int N, float* QlS, float* QrS, hipStream_t st,
                 float* scal = nullptr);   // scal: 64 scratch words to zero in the same launch (or null)

// Solve y Q = x for nvec vectors (vector i at stride si, element j at stride sj; Q upper triangular [n][n]);
// dinv: scratch of ceil(n/32) * 1024 floats.  0 on success.
// lite: the trailing products of the blocked solve keep only the three leading terms of the bf16 x 3 split (2^-16 relative)
int kron_trsm_ut(const float* Q, int n, const float* X, float* Y, int nvec, long si, long sj, float* dinv, hipStream_t st,
                 int lite = 0);

}  // namespace psgdk
