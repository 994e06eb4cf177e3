// Shared host-side helpers of libggnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include "../../include/ggnn_hip.h"

namespace ggnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNumXcd = 8;   // MI355X: 8 XCDs, block b is observed on XCD b % 8

// thread-local error text behind ggnn_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

#define GGNN_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return ::ggnn::fail(GGNN_E_INVALID, __VA_ARGS__); } while (0)

#define GGNN_CHECK_HIP(expr)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::ggnn::fail(GGNN_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Number of CUs of the current device (cached per process; MI355X = 256).
int num_cus();

// Raise a kernel's dynamic-LDS limit once per (kernel, device): `done` is a per-kernel bitmask of the devices that have
// it (the attribute is per device; one process may drive several).  Thread-safe; a lost race only repeats the call.
template <class Kernel>
inline hipError_t allow_dynamic_lds(Kernel* kernel, size_t bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ULL << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes); e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

}  // namespace ggnn
