// Shared host-side helpers of libggnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

}  // namespace ggnn
