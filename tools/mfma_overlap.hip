// Micro-benchmark (not part of the product): do VALU / transcendental instructions of one wave overlap with the
// MFMA burst of the OTHER wave on the same SIMD of gfx950?  (Decides whether the fused GRU's epilogues can hide under
// the partner wave's MFMAs.)   hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
//
// One workgroup of 8 waves per CU (waves w and w+4 share a SIMD).  Modes: what waves 0-3 / waves 4-7 do
//   0: mfma / idle   1: idle / valu   2: mfma / valu   3: mfma / mfma   4: valu / valu   5: mfma / trans   6: idle / trans
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef BF16     // -DBF16: the same question for v_mfma_f32_16x16x32_bf16 (the 3-way split kernels, ggnn_split.hpp): 16 clocks each
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void mfma_burst(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, float x, float y, int n) {
    const u32x4 xa = {__float_as_uint(x), __float_as_uint(y), 0x3f803f80u, 0x3f003f00u}, ya = {0x3f803f80u, __float_as_uint(y), 0x3e803e80u, __float_as_uint(x)};
    const bf16x8 xb = __builtin_bit_cast(bf16x8, xa), yb = __builtin_bit_cast(bf16x8, ya);
    for (int i = 0; i < 2 * n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a3, 0, 0, 0);
    }
}
#else
__device__ __forceinline__ void mfma_burst(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, float x, float y, int n) {
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
}

#endif

__device__ __forceinline__ float valu_burst(float v, int n) {          // 8 independent FMA chains
    float c0 = v, c1 = v + 1, c2 = v + 2, c3 = v + 3, c4 = v + 4, c5 = v + 5, c6 = v + 6, c7 = v + 7;
    for (int i = 0; i < n; ++i) {
        c0 = fmaf(c0, 1.0001f, 0.5f); c1 = fmaf(c1, 1.0001f, 0.5f); c2 = fmaf(c2, 1.0001f, 0.5f); c3 = fmaf(c3, 1.0001f, 0.5f);
        c4 = fmaf(c4, 1.0001f, 0.5f); c5 = fmaf(c5, 1.0001f, 0.5f); c6 = fmaf(c6, 1.0001f, 0.5f); c7 = fmaf(c7, 1.0001f, 0.5f);
    }
    return c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

__device__ __forceinline__ float trans_burst(float v, int n) {         // exp + rcp: the sigmoid / tanh epilogue mix
    float c0 = v, c1 = v + 0.1f, c2 = v + 0.2f, c3 = v + 0.3f;
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_rcpf(1.0f + __expf(-c0)); c1 = __builtin_amdgcn_rcpf(1.0f + __expf(-c1));
        c2 = __builtin_amdgcn_rcpf(1.0f + __expf(-c2)); c3 = __builtin_amdgcn_rcpf(1.0f + __expf(-c3));
    }
    return c0 + c1 + c2 + c3;
}

__global__ __launch_bounds__(512) void k(int mode, int n_mfma, int n_valu, float* out, unsigned long long* t) {
    extern __shared__ float pad[];                      // 100 KiB: one workgroup per CU
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float r = 0.f;
    const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    const int what = second ? (mode == 1 || mode == 2 || mode == 7 || mode == 8 ? 2 : mode == 3 ? 1 : mode == 4 ? 2 : mode == 5 || mode == 6 ? 3 : 0)
                            : (mode == 0 || mode == 2 || mode == 3 || mode == 5 || mode == 7 || mode == 8 ? 1 : mode == 4 ? 2 : 0);
    __syncthreads();
    if (mode == 7 && second) __builtin_amdgcn_s_setprio(2);     // 7: the VALU wave at raised issue priority
    if (mode == 8 && !second) __builtin_amdgcn_s_setprio(2);    // 8: the MFMA wave at raised issue priority
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (what == 1) mfma_burst(a0, a1, a2, a3, x, y, n_mfma);
    else if (what == 2) r = valu_burst(x, n_valu);
    else if (what == 3) r = trans_burst(x, n_valu / 8);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { t[wave * 2] = t1 - t0; t[wave * 2 + 1] = t2 - t0; }
    out[blockIdx.x * 512 + threadIdx.x] = a0.x + a1.y + a2.z + a3.w + r + pad[threadIdx.x & 7] * 0.f;
}

int main() {
    float* out; unsigned long long* t;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&t, 16 * sizeof(unsigned long long));
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const char* names[] = {"mfma / idle", "idle / valu", "mfma / valu", "mfma / mfma", "valu / valu", "mfma / trans", "idle / trans", "mfma / VALU+", "MFMA+ / valu"};
    const int n_mfma = 200;            // x4 = 800 MFMAs = 25.6k clocks
    const int n_valu = 800;            // x8 FMAs = 6400 VALU ops = 25.6k clocks alone (4 clocks each)
    for (int mode = 0; mode < 9; ++mode) {
        std::vector<unsigned long long> h(16);
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 100 * 1024, 0, mode, n_mfma, n_valu, out, t);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), t, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        printf("mode %d (%-12s): wave0 own %6llu  wave4 own %6llu  | until barrier: %6llu clocks\n", mode, names[mode],
               h[0], h[8], h[1]);
    }
    return 0;
}
