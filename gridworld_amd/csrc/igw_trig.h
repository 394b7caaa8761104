// igw_trig.h -- double-precision sin/cos/atan2 used where the walking LUT does not apply
// (flying mode, non-5-degree poses).  PLACEHOLDER: forwards to the device math library.
#pragma once
#include <hip/hip_runtime.h>

namespace igw {
__device__ inline void igw_sincos(double x, double* s, double* c) {
    *s = ::sin(x);
    *c = ::cos(x);
}
__device__ inline double igw_atan2(double y, double x) { return ::atan2(y, x); }
}  // namespace igw
