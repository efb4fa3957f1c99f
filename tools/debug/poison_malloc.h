// Debug allocator for A/B builds (never in the product): every hipMalloc of the library is filled
// with 0xFF bytes (NaN doubles, -1 integers), so that a read of memory the library never wrote
// shows up in the results instead of reading the zero pages a fresh process hands out.
#pragma once
#include <hip/hip_runtime.h>
namespace pbpoison {
inline hipError_t raw_malloc(void **p, size_t n) { return hipMalloc(p, n); }
template <class T> inline hipError_t pmalloc(T **p, size_t n)
{
    hipError_t e = raw_malloc((void **)p, n);
    // (hipMemset of device memory returns before the fill has run, and the null stream does not
    // order it against the library's non-blocking streams: without the wait the fill can land
    // AFTER the first kernels that write the buffer -- seen as NaNs in valid results)
    if (e == hipSuccess && n) {
        e = hipMemset(*p, 0xFF, n);
        if (e == hipSuccess)
            e = hipDeviceSynchronize();
    }
    return e;
}
}  // namespace pbpoison
#define hipMalloc(p, n) pbpoison::pmalloc((p), (n))
