// Library-level entry points: error reporting, device selection.
#include "pb_common.h"

namespace pb {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
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
