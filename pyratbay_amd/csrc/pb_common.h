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

// per-(device, stream) scratch that persists between calls (pb_core.hip); nullptr when it cannot
// be provided (out of memory, or it would have to grow while the stream is being captured)
void *stream_scratch(hipStream_t s, size_t bytes);

// rocTX range around a stage or a collective (no-ops without the profiler's marker library);
// stage_boundary: a fused entry point tells the calling thread's running pb_timer, if any, that
// the stage `name` ends here (pb_core.hip)
void range_push(const char *name);
void range_pop();
void stage_boundary(const char *name, const char *next_stage, hipStream_t s);

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

// x / d with inv = RN(1 / d) prepared once for many numerators: the closing steps of the division
// the compiler would emit -- product, exact residual, one correction -- without its reciprocal
// refinement (3 instructions instead of ~12).  Claim: within 1 ulp of x / d for d, x / d and the
// residual in the normal range (no proof of correct rounding: Markstein's theorem wants a faithful
// q, which RN(x RN(1/d)) is not guaranteed to be); measured: the same bits as x / d on every one of
// 6.9e10 random (x, d) pairs of tools/quot_probe.hip, drawn over the ranges the callers use
// (temperatures, mu, kT; numerators over 600 binades).  Special values: x = +-inf, d = 0 and
// d = +-inf make the short form NaN where the division is +-inf or 0 (exp(-inf / mu) must be 0,
// B(T = 0) must be 0 like in the reference), and x = -0 gives +0; a result that is NaN or zero
// (one compare: also a quotient that underflowed) therefore takes the true division.
__device__ __forceinline__ double quot(double x, double d, double inv)
{
    const double q = x * inv;
    const double r = fma(fma(-q, d, x), inv, q);
    if (__builtin_expect(!(fabs(r) > 0.0), 0))
        return x / d;
    return r;
}

// The same without the guard, for callers that keep the special values away from it themselves
// (the guard's compare + branch per quotient cost 5 % of k_records and of the emission kernels):
//   * divisors through sane_divisor(): 0 -> (2^-1000, 2^1000), +inf -> (2^1000, 2^-1000) -- the
//     quotient is then huge / tiny but finite, and what the callers do with it (exp, then a
//     product or a division) ends in the same 0 the true division gives;
//   * numerators clamped to a finite value with the same result (clamp_depth: exp(-1e5 / mu) is
//     exactly 0 for mu <= 1, like exp(-inf / mu));
//   * or one test of the END result and a recomputation with true divisions (line_strength).
__device__ __forceinline__ double quot_fast(double x, double d, double inv)
{
    const double q = x * inv;
    return fma(fma(-q, d, x), inv, q);
}
__device__ __forceinline__ void sane_divisor(double d, double &ds, double &inv)
{
    ds = d;
    inv = 1.0 / d;
    if (d == 0.0) {
        ds = 0x1p-1000;
        inv = 0x1p+1000;
    } else if (d > 0x1p+1000) {
        ds = 0x1p+1000;
        inv = 0x1p-1000;
    }
}
// an optical depth as the numerator of exp(-depth / mu): NaN stays NaN, +inf and anything above 1e5
// become 1e5 (the exponential is exactly 0 either way)
__device__ __forceinline__ double clamp_depth(double t)
{
    return t > 1e5 ? 1e5 : t;
}

// exp() with its constants in SGPRs.  The arithmetic is the device library's (range reduction by
// ln2 in two parts, degree-11 polynomial, ldexp, the same overflow/underflow selects), so the
// result is the same bit pattern; what changes is that the polynomial coefficients sit in scalar
// registers (an empty asm with an "s" constraint hides each value from constant folding and pins
// it to an SGPR pair, loop-invariant) instead of being re-materialised with two v_mov each per
// call: ~25 vector instructions per exp instead of ~50.  The column kernels evaluate one exp
// per (layer, sample) and are bound by exactly that instruction count.
__device__ __forceinline__ double sgpr_const(double v)
{
    asm("" : "+s"(v));
    return v;
}

__device__ __forceinline__ double exp_s(double x)
{
    const double log2e = sgpr_const(0x1.71547652b82fep+0);
    const double nln2h = sgpr_const(-0x1.62e42fefa39efp-1);
    const double nln2l = sgpr_const(-0x1.abc9e3b39803fp-56);
    const double c11 = sgpr_const(0x1.ade156a5dcb37p-26), c10 = sgpr_const(0x1.28af3fca7ab0cp-22),
                 c9 = sgpr_const(0x1.71dee623fde64p-19), c8 = sgpr_const(0x1.a01997c89e6b0p-16),
                 c7 = sgpr_const(0x1.a01a014761f6ep-13), c6 = sgpr_const(0x1.6c16c1852b7b0p-10),
                 c5 = sgpr_const(0x1.1111111122322p-7), c4 = sgpr_const(0x1.55555555502a1p-5),
                 c3 = sgpr_const(0x1.5555555555511p-3), c2 = sgpr_const(0x1.000000000000bp-1);
    const double n = rint(x * log2e);
    double r = fma(n, nln2h, x);
    r = fma(nln2l, n, r);
    double p = fma(c11, r, c10);
    p = fma(r, p, c9);
    p = fma(r, p, c8);
    p = fma(r, p, c7);
    p = fma(r, p, c6);
    p = fma(r, p, c5);
    p = fma(r, p, c4);
    p = fma(r, p, c3);
    p = fma(r, p, c2);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    double v = ldexp(p, (int)n);
    v = x > 1024.0 ? __builtin_huge_val() : v;
    return x < -1075.0 ? 0.0 : v;
}

// The layers nobody reads of a retrieval batch with ordered columns (pb_batch.hip, above
// k_interp_ec_batch): limits per block of 256 columns, the repair pass's gate.
struct TileLimit {
    const int32_t *tile;      // [ceil(nwave / 256)] last row tile available, or null: every layer
    int row0;                 // itop: layers above it are not read by the transit pass either
    const int32_t *gate;      // null, or: run only if *gate != 0 (interp) / gate[walker] != 0 (transit)
};

__device__ __forceinline__ int uniform_i32(const int32_t *p)
{
    typedef const int32_t __attribute__((address_space(4))) *cptr;
    return *((cptr)(unsigned long long)p);
}

// is layer k of the samples [s0, s1) wanted?  (wave-uniform)
__device__ __forceinline__ bool layer_wanted(const TileLimit &lim, int k, int s0, int s1, int nwave)
{
    if (lim.gate && uniform_i32(lim.gate) == 0)
        return false;
    if (!lim.tile)
        return true;
    if (k < lim.row0)
        return false;
    const int b1 = (min(s1, nwave) - 1) >> 8;
    int t = 0;
    for (int b = max(s0, 0) >> 8; b <= b1; b++)
        t = max(t, uniform_i32(lim.tile + b));
    return k <= lim.row0 + 16 * (t + 1) - 1;
}


}  // namespace pb
