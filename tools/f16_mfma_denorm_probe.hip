// Does v_mfma_f32_16x16x32_f16 keep SUBNORMAL f16 inputs?  (GGNN_SPLIT2, csrc/ggnn_split.hpp: the lo piece of an unscaled operand
// below 0.125 is an f16 subnormal.)  A = 2^-20 in every slot (an f16 subnormal: 16 x 2^-24), B = 1024: every output element is
// 32 x 2^-20 x 2^10 = 2^-5 when the inputs are kept, 0 when they are flushed.
//   hipcc --offload-arch=gfx950 -O2 tools/f16_mfma_denorm_probe.hip -o tools/_bin/f16_mfma_denorm_probe && tools/_bin/f16_mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(float a, float b, float* out) {
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)A[0]; }
}
int main() {
    float* d; float h[2];
    if (hipMalloc(&d, 8) != hipSuccess) { printf("no device\n"); return 1; }
    const float cases[3][2] = {{9.5367431640625e-07f /* 2^-20 */, 1024.f}, {5.9604644775390625e-08f /* 2^-24, the smallest */, 16384.f}, {0.5f, 0.25f}};
    const float want[3] = {32 * 9.5367431640625e-07f * 1024.f, 32 * 5.9604644775390625e-08f * 16384.f, 32 * 0.125f};
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, cases[i][0], cases[i][1], d);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %.9g (as f16: %.9g)  b = %g   mfma sum = %.9g   want %.9g   %s\n", cases[i][0], h[1], cases[i][1], h[0], want[i],
               h[0] == want[i] ? "kept" : (h[0] == 0.f ? "FLUSHED" : "other"));
    }
    return 0;
}
