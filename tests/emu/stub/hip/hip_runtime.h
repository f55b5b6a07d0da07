/* Host stand-ins for the gfx950 builtins ort_device.hpp uses — TEST INFRASTRUCTURE ONLY (tests/emu/): lets the CPU
 * suite run the device's per-surface step functions (both arithmetic policies, the near-branch `odd` logic, the
 * polynomial forms) against the oracle without a GPU.  Seeds are rounded to binary32 like the hardware's
 * (2^-24-accurate v_rcp_f64 / v_rsq_f64); everything after the seed is the header's own arithmetic. */
#pragma once
#include <cmath>
#include <cstdint>
#define __device__
#define __host__
#define __constant__ static
#define __forceinline__ inline
#define __global__
static inline double __builtin_amdgcn_rcp(double a) { return (double)(float)(1.0 / a); }
static inline float __builtin_amdgcn_rcpf(float a) { return 1.0f / a; }
static inline double __builtin_amdgcn_rsq(double a) { return (double)(float)(1.0 / std::sqrt(a)); }
static inline float __builtin_amdgcn_rsqf(float a) { return 1.0f / std::sqrt(a); }
static inline float __builtin_amdgcn_sqrtf(float a) { return std::sqrt(a); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline double __builtin_amdgcn_div_fixup(double q, double b, double a)
{
    if (std::isnan(a) || std::isnan(b) || std::isinf(a) || std::isinf(b) || a == 0.0 || b == 0.0) return a / b;
    return q;
}
template <typename F> static inline bool emu_class(F x, int mask)
{
    int bit;
    if (std::isnan(x)) bit = 1;                                       /* quiet NaN (signalling: bit 0) */
    else if (std::isinf(x)) bit = x < 0 ? 2 : 9;
    else if (x == 0) bit = std::signbit(x) ? 5 : 6;
    else if (std::fpclassify(x) == FP_SUBNORMAL) bit = x < 0 ? 4 : 7;
    else bit = x < 0 ? 3 : 8;
    return ((mask >> bit) & 1) || (bit == 1 && (mask & 1));
}
static inline bool __builtin_amdgcn_class(double x, int mask) { return emu_class(x, mask); }
static inline bool __builtin_amdgcn_classf(float x, int mask) { return emu_class(x, mask); }
