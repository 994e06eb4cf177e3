// Operand-range check of the two-piece f16 operand format (GGNN_GRU_FMT_F16X2, include/ggnn_hip.h "Operand formats"): max |x| of a
// list of tensors in one launch.  HBM-bound (a [V,D] state of 40 MB: ~8 us); the maximum is taken over the BIT PATTERNS of |x| --
// for non-negative floats integer order is float order, Inf sits above every finite value and every NaN above Inf -- so a
// non-finite input gives a non-finite result and the host's `max <= bound` test fails on it.
#include "ggnn_common.h"

namespace ggnn {
namespace {

constexpr int kAbsmaxBatch = 32;                 // tensors per launch
struct AbsmaxArgs {
    const float* p[kAbsmaxBatch];
    long long n[kAbsmaxBatch];
    unsigned* out;
};

__global__ __launch_bounds__(256) void absmax_kernel(AbsmaxArgs a) {
    const int i = blockIdx.y;
    const float* __restrict__ x = a.p[i];
    const long long n = a.n[i];
    unsigned m = 0u;
    const long long start = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    // 16-byte loads over the aligned middle, scalar head / tail
    const long long head = ((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / 4;
    const long long h = head < n ? head : n;
    const long long nv = (n - h) / 4;
    const uint4* xv = reinterpret_cast<const uint4*>(x + h);
    for (long long j = start; j < nv; j += stride) {
        const uint4 v = xv[j];
        const unsigned a0 = v.x & 0x7fffffffu, a1 = v.y & 0x7fffffffu, a2 = v.z & 0x7fffffffu, a3 = v.w & 0x7fffffffu;
        const unsigned b0 = a0 > a1 ? a0 : a1, b1 = a2 > a3 ? a2 : a3;
        const unsigned b = b0 > b1 ? b0 : b1;
        m = b > m ? b : m;
    }
    for (long long j = start; j < h; j += stride) { const unsigned v = __float_as_uint(x[j]) & 0x7fffffffu; m = v > m ? v : m; }
    for (long long j = h + 4 * nv + start; j < n; j += stride) { const unsigned v = __float_as_uint(x[j]) & 0x7fffffffu; m = v > m ? v : m; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    __shared__ unsigned sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r = sm[0];
        for (int w = 1; w < 4; ++w) r = sm[w] > r ? sm[w] : r;
        if (r) atomicMax(a.out + i, r);
    }
}

}  // namespace
}  // namespace ggnn

using namespace ggnn;

extern "C" int ggnn_absmax_f32(const float* const* ptrs, const int64_t* numel, int n, float* out, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(n >= 0 && (n == 0 || (ptrs && numel && out)), "null pointer");
    if (n == 0) return GGNN_OK;
    hipStream_t st = (hipStream_t)stream;
    GGNN_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)n * sizeof(float), st));
    for (int i0 = 0; i0 < n; i0 += kAbsmaxBatch) {
        AbsmaxArgs a{};
        const int nb = n - i0 < kAbsmaxBatch ? n - i0 : kAbsmaxBatch;
        long long largest = 0;
        for (int i = 0; i < nb; ++i) {
            GGNN_CHECK_ARG(numel[i0 + i] >= 0 && (numel[i0 + i] == 0 || ptrs[i0 + i]), "tensor %d: null pointer or negative size", i0 + i);
            GGNN_CHECK_ARG((reinterpret_cast<uintptr_t>(ptrs[i0 + i]) & 3) == 0, "tensor %d is not 4-byte aligned", i0 + i);
            a.p[i] = ptrs[i0 + i]; a.n[i] = numel[i0 + i];
            if (a.n[i] > largest) largest = a.n[i];
        }
        a.out = reinterpret_cast<unsigned*>(out) + i0;
        // enough blocks for the largest tensor of the batch to stream at HBM rate (4 KiB per block and iteration), at most 8 per CU
        long long bx = (largest + 1023) / 1024;
        const long long cap = 8LL * num_cus();
        if (bx > cap) bx = cap;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)bx, (unsigned)nb), dim3(256), 0, st, a);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    return GGNN_OK;
}
