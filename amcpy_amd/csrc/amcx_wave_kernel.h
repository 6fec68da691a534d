// AMCX_VARIANT_WAVE placeholder (replaced by the register-FFT kernel).
#pragma once
#include "amcx_math.h"
namespace amcx {
inline bool wave_supports(int) { return false; }
inline const char* wave_kernel_name(int) { return ""; }
inline hipError_t launch_wave(const float2*, int64_t, int32_t, int64_t, float*, int64_t, hipStream_t, int) {
  return hipErrorNotSupported;
}
}  // namespace amcx
