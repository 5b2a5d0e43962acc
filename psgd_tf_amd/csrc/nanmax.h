// nanmax.h -- maxima that propagate NaN.  tf.reduce_max returns NaN if any element is NaN (psgd.py:41,166-167,177-178,
// 563-564,582 feed such maxima into step sizes: a NaN anywhere turns the whole updated factor into NaN, README.md:56
// "NaN/Inf propagate silently"); fmaxf / v_max_f32 return the non-NaN operand and would hide it behind a finite step.
#pragma once
#include <hip/hip_runtime.h>

namespace psgd {

// any operands
__device__ __forceinline__ float nmaxf(float a, float b) {
  const float m = fmaxf(a, b);
  return (a != a || b != b) ? __uint_as_float(0x7fc00000u) : m;
}
__device__ __forceinline__ double nmax(double a, double b) {
  const double m = fmax(a, b);
  return (a != a || b != b) ? __longlong_as_double(0x7ff8000000000000LL) : m;
}
// non-negative operands (|x| and maxima of |x|, sign bit clear): the unsigned order of the bit patterns is the float
// order and puts NaN above +inf -- one v_max_u32, the cost of the fmaxf it replaces
__device__ __forceinline__ float amaxf(float a, float b) {
  const unsigned x = __float_as_uint(a), y = __float_as_uint(b);
  return __uint_as_float(x > y ? x : y);
}

}  // namespace psgd
