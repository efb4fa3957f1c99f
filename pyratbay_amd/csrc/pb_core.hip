// Library-level entry points: error reporting, device selection.
#include "pb_common.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace pb {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// Small per-stream device scratch that outlives the call (the blocked ray paths of a transit
// launch): one buffer per (device, stream), grown by doubling, never handed to another stream, so
// work queued on a stream may keep using it.  Plain hipMalloc on first use instead of
// hipMallocAsync / hipFreeAsync per call: no allocation on the launch path, and a step captured
// into a HIP graph carries no memory nodes (transit_launch does not use the scratch at all while
// its stream is being captured).  Outgrown buffers are retired, not freed: kernels already queued
// may still read them (total < 2x the largest).
void *stream_scratch(hipStream_t s, size_t bytes)
{
    struct Entry {
        void *p = nullptr;
        size_t cap = 0;
    };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Entry> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mu);
    Entry &e = table[std::make_pair(dev, s)];
    if (e.cap < bytes) {
        size_t cap = std::max<size_t>(e.cap, (size_t)64 << 10);
        while (cap < bytes)
            cap *= 2;
        void *p = nullptr;
        if (hipMalloc(&p, cap) != hipSuccess) {    // (also: not allowed while s is being captured)
            (void)hipGetLastError();
            return nullptr;
        }
        e.p = p;                                   // the old buffer is retired, see above
        e.cap = cap;
    }
    return e.p;
}

}  // namespace pb

extern "C" {

const char *pb_last_error(void) { return pb::g_error; }

int pb_version(void) { return 100; }

int pb_device_count(int *count)
{
    PB_REQUIRE(count, "pb_device_count: null pointer");
    *count = 0;
    PB_HIP(hipGetDeviceCount(count));
    return PB_OK;
}

int pb_set_device(int device)
{
    PB_HIP(hipSetDevice(device));
    return PB_OK;
}

}  // extern "C"
