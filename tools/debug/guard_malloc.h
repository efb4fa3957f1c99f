// Debug allocator for A/B builds (never in the product): every hipMalloc of the library becomes its
// own virtual-memory mapping whose END coincides with the end of the mapped range, followed by an
// unmapped guard range -- the first read or write past the end of a buffer faults at once
// (16-byte granularity), whatever else is allocated in the process.  hipFree is a no-op.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

namespace pbguard {
inline hipError_t guard_malloc(void **p, size_t n)
{
    static size_t gran = 0;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    int dev = 0;
    (void)hipGetDevice(&dev);
    prop.location.id = dev;
    if (!gran && hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess)
        return hipErrorOutOfMemory;
    const size_t need = n ? ((n + 15) & ~(size_t)15) : 16;
    const size_t mapped = ((need + gran - 1) / gran) * gran;
    void *va = nullptr;
    if (hipMemAddressReserve(&va, mapped + gran, gran, nullptr, 0) != hipSuccess)
        return hipErrorOutOfMemory;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, mapped, &prop, 0) != hipSuccess)
        return hipErrorOutOfMemory;
    if (hipMemMap(va, mapped, 0, h, 0) != hipSuccess)
        return hipErrorOutOfMemory;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, mapped, &acc, 1) != hipSuccess)
        return hipErrorOutOfMemory;
    *p = (char *)va + (mapped - need);
    if (getenv("PB_GUARD_TRACE"))
        fprintf(stderr, "guard_malloc %zu B -> [%p, %p)\n", n, *p, (char *)*p + need);
    return hipSuccess;
}
template <class T> inline hipError_t guard_malloc_t(T **p, size_t n) { return guard_malloc((void **)p, n); }
inline hipError_t guard_free(void *) { return hipSuccess; }
}  // namespace pbguard
#define hipMalloc(p, n) pbguard::guard_malloc_t((p), (n))
#define hipFree(p) pbguard::guard_free((void *)(p))
