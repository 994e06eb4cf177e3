// ABI version, thread-local error text, device properties.
#include "ggnn_common.h"
#include <atomic>

namespace ggnn {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int num_cus() {
    static std::atomic<int> cached{0};
    int v = cached.load(std::memory_order_relaxed);
    if (v > 0) return v;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;   // MI355X
    cached.store(n, std::memory_order_relaxed);
    return n;
}

}  // namespace ggnn

extern "C" int ggnn_abi_version(void) { return GGNN_ABI_VERSION; }
extern "C" const char* ggnn_last_error(void) { return ggnn::error_buffer(); }
