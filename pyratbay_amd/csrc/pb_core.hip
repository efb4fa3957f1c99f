// Library-level entry points: error reporting, device selection.
#include "pb_common.h"

#include <dlfcn.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

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

// ---------------------------------------------------------------------------
// rocTX ranges (what rocprofv3 --marker-trace shows): resolved once from the profiler SDK's
// marker library when it is on the loader path, otherwise no-ops.  This is a profiling aid
// only -- nothing on the compute path depends on it.
// ---------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (const char *e = getenv("PB_ROCTX"))
            if (atoi(e) == 0)
                return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                                 "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!h)
                continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop)
                return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
const Roctx &roctx()
{
    static const Roctx r;
    return r;
}
thread_local pb_timer *t_timer = nullptr;    // timer the fused calls of this thread mark
}  // namespace

void range_push(const char *name)
{
    if (roctx().push)
        roctx().push(name);
}

void range_pop()
{
    if (roctx().pop)
        roctx().pop();
}

}  // namespace pb

// Stage timer: HIP events recorded on the caller's stream at the END of each named stage, read
// back lazily (Pyrat.timestamps: pyratbay/pyrat/pyrat_obj.py:203-214, tools/tools.py:832-843).
struct pb_timer {
    std::vector<hipEvent_t> ev;          // ev[0] = start, ev[i + 1] = end of stage i
    std::vector<std::string> names;
    int used = 0;                        // events recorded since the last start
    int open_ranges = 0;
    // a run is OPEN from pb_timer_start until a pb_timer_mark that names no next stage: only then
    // may a fused entry point mark its internal boundary on this timer (a finished run must not
    // collect the boundaries of later, unrelated calls of the same thread)
    bool open = false;
};

namespace {
// events recorded on a capturing stream become graph nodes: reading them later synchronises
// nothing meaningful.  A stage timer records nothing while its stream is being captured.
bool capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}
}  // namespace

namespace pb {

// called by fused entry points between their internal stages
void stage_boundary(const char *name, const char *next_stage, hipStream_t s)
{
    pb_timer *t = t_timer;
    if (!t || !t->open || t->used == 0 || t->used >= (int)t->ev.size() || capturing(s))
        return;
    if (hipEventRecord(t->ev[t->used], s) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    t->names[t->used - 1] = name;
    t->used++;
    if (t->open_ranges > 0) {
        range_pop();
        t->open_ranges--;
        if (next_stage) {
            range_push(next_stage);
            t->open_ranges++;
        }
    }
}

}  // namespace pb

extern "C" {

int pb_range_push(const char *name)
{
    PB_REQUIRE(name, "pb_range_push: null name");
    pb::range_push(name);
    return PB_OK;
}

int pb_range_pop(void)
{
    pb::range_pop();
    return PB_OK;
}

int pb_roctx_available(void) { return pb::roctx().push != nullptr ? 1 : 0; }

int pb_timer_create(pb_timer **out, int max_stages)
{
    PB_REQUIRE(out && max_stages >= 1 && max_stages <= 64, "pb_timer_create: bad argument");
    *out = nullptr;
    pb_timer *t = new (std::nothrow) pb_timer();
    if (!t)
        return PB_ERR_NOMEM;
    for (int i = 0; i <= max_stages; i++) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            pb::set_error("pb_timer_create: hipEventCreate failed");
            pb_timer_destroy(t);
            return PB_ERR_HIP;
        }
        t->ev.push_back(e);
    }
    t->names.assign((size_t)max_stages, std::string());
    *out = t;
    return PB_OK;
}

int pb_timer_start(pb_timer *t, const char *first_stage, void *stream)
{
    PB_REQUIRE(t, "pb_timer_start: null timer");
    while (t->open_ranges > 0) {
        pb::range_pop();
        t->open_ranges--;
    }
    if (capturing(pb::as_stream(stream))) {
        // (the run being captured keeps the timestamps of the last eager run)
        t->open = false;
        if (pb::t_timer == t)
            pb::t_timer = nullptr;
        return PB_OK;
    }
    PB_HIP(hipEventRecord(t->ev[0], pb::as_stream(stream)));
    t->used = 1;
    t->open = true;
    pb::t_timer = t;
    if (first_stage) {
        pb::range_push(first_stage);
        t->open_ranges = 1;
    }
    return PB_OK;
}

int pb_timer_mark(pb_timer *t, const char *name, const char *next_stage, void *stream)
{
    PB_REQUIRE(t && name, "pb_timer_mark: null pointer");
    if (!t->open && capturing(pb::as_stream(stream)))
        return PB_OK;                    // pb_timer_start skipped this (captured) run
    PB_REQUIRE(t->used >= 1, "pb_timer_mark: timer not started");
    PB_REQUIRE(t->used < (int)t->ev.size(), "pb_timer_mark: more than %d stages",
               (int)t->ev.size() - 1);
    PB_HIP(hipEventRecord(t->ev[t->used], pb::as_stream(stream)));
    t->names[t->used - 1] = name;
    t->used++;
    if (t->open_ranges > 0) {
        pb::range_pop();
        t->open_ranges--;
    }
    if (next_stage) {
        pb::range_push(next_stage);
        t->open_ranges++;
    } else {
        // the run is over: later fused calls of this thread belong to nobody's timer
        t->open = false;
        if (pb::t_timer == t)
            pb::t_timer = nullptr;
    }
    return PB_OK;
}

int pb_timer_count(const pb_timer *t, int *nstages)
{
    PB_REQUIRE(t && nstages, "pb_timer_count: null pointer");
    *nstages = t->used > 0 ? t->used - 1 : 0;
    return PB_OK;
}

int pb_timer_read(pb_timer *t, int stage, char *name_out, int name_cap, double *seconds)
{
    PB_REQUIRE(t && seconds, "pb_timer_read: null pointer");
    PB_REQUIRE(stage >= 0 && stage + 1 < t->used, "pb_timer_read: stage %d was not recorded", stage);
    PB_HIP(hipEventSynchronize(t->ev[stage + 1]));
    float ms = 0.f;
    PB_HIP(hipEventElapsedTime(&ms, t->ev[stage], t->ev[stage + 1]));
    *seconds = 1e-3 * (double)ms;
    if (name_out && name_cap > 0) {
        strncpy(name_out, t->names[stage].c_str(), (size_t)name_cap - 1);
        name_out[name_cap - 1] = 0;
    }
    return PB_OK;
}

void pb_timer_destroy(pb_timer *t)
{
    if (!t)
        return;
    if (pb::t_timer == t)
        pb::t_timer = nullptr;
    while (t->open_ranges > 0) {
        pb::range_pop();
        t->open_ranges--;
    }
    for (hipEvent_t e : t->ev)
        (void)hipEventDestroy(e);
    delete t;
}

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
