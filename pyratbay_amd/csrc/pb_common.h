// Shared host/device helpers of libpbhip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "pbhip.h"

namespace pb {

// Physical constants: the reference's kernels use legacy CODATA values
// (src_c/include/constants.h:5-21); parity requires these exact numbers.
constexpr double kPi = 3.141592653589793;
constexpr double kSqrtLn2 = 0.83255461115769775635;
constexpr double kTwoOSqrtPi = 1.12837916709551257389;
constexpr double kSqrtLn2Pi = 0.46971863934982566689;
constexpr double kLS = 2.99792458e10;
constexpr double kKB = 1.380658e-16;
constexpr double kAMU = 1.66053886e-24;
constexpr double kH = 6.6260755e-27;
constexpr double kEC = 4.8032068e-10;
constexpr double kME = 9.1093897e-28;
constexpr double kSigCte = kPi * kEC * kEC / kLS / kLS / kME;
constexpr double kExpCte = kH * kLS / kKB;

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define PB_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            pb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                          __FILE__, __LINE__);                                    \
            return PB_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

#define PB_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            pb::set_error(__VA_ARGS__);  \
            return PB_ERR_ARG;           \
        }                                \
    } while (0)

#define PB_LAUNCH_CHECK()                                                       \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            pb::set_error("kernel launch failed: %s (%s:%d)",                   \
                          hipGetErrorString(e_), __FILE__, __LINE__);           \
            return PB_ERR_HIP;                                                  \
        }                                                                       \
    } while (0)

inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Index of the element of a[lo..hi] closest to v; bisection keeps the upper half
// when a[mid] <= v, ties go to the lower index (restates binsearchapprox,
// src_c/include/utils.h:75-89, iteratively).
__host__ __device__ inline int nearest_index(const double *a, double v, int lo, int hi)
{
    while (hi - lo > 1) {
        int mid = (hi + lo) / 2;
        if (a[mid] > v)
            hi = mid;
        else
            lo = mid;
    }
    return (fabs(a[hi] - v) < fabs(a[lo] - v)) ? hi : lo;
}

// C integer division helpers on possibly negative numerators
__host__ __device__ inline int64_t floor_div(int64_t a, int64_t b)
{
    int64_t q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}
__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b)
{
    return -floor_div(-a, b);
}

}  // namespace pb
