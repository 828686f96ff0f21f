// portable_log.hpp -- natural logarithm of a double from IEEE add / multiply / divide and integer operations only, so that the
// host build and the device build (both compiled with -ffp-contract=off) return the SAME bits for the same argument.
//
// Used where the lattice rerank of search mode turns perplexities and path costs into normalised logs
// (/root/reference/src/lib.rs:2388-2411): with the C library's log on the host and the device library's on the GPU the two
// decoders could disagree in the last bit and pick different paths on a near tie.  The reference calls f64::ln; this routine is
// within 1 ulp of the correctly rounded value, like the C libraries'.
//
// Algorithm: the classic argument reduction x = 2^k (1 + f), log(1 + f) = 2s + s R(s^2) with s = f / (2 + f) and a degree-14
// minimax polynomial R, as published in FreeBSD's msun / fdlibm e_log.c:
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunSoft, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define ANX_PL_HD __host__ __device__ inline
#else
#define ANX_PL_HD inline
#endif

namespace anx {

ANX_PL_HD double portable_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  uint64_t bits;
  memcpy(&bits, &x, sizeof bits);
  int32_t hx = (int32_t)(bits >> 32);
  uint32_t lx = (uint32_t)bits;
  int32_t k = 0;
  if (hx < 0x00100000) {  // x < 2^-1022
    if (((hx & 0x7fffffff) | (int32_t)lx) == 0) return -two54 / 0.0;  // log(+-0) = -inf
    if (hx < 0) return (x - x) / 0.0;                                  // log(-#) = NaN
    k -= 54;
    x *= two54;  // subnormal: scale up
    memcpy(&bits, &x, sizeof bits);
    hx = (int32_t)(bits >> 32);
  }
  if (hx >= 0x7ff00000) return x + x;  // inf, NaN
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  bits = (bits & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);  // normalise x or x / 2
  memcpy(&x, &bits, sizeof x);
  k += i >> 20;
  const double f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2^-20
    if (f == 0.0) {
      if (k == 0) return 0.0;
      const double dk = (double)k;
      return dk * ln2_hi + dk * ln2_lo;
    }
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    const double dk = (double)k;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s;
  i = hx - 0x6147a;
  const double w = z * z;
  const int32_t j = 0x6b851 - hx;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  const double R = t2 + t1;
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

}  // namespace anx
