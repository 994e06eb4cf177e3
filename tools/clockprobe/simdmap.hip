// Which SIMD does wave w of a workgroup run on?  (HW_ID.simd_id of every wave of a few workgroups, for several workgroup sizes.)
//   hipcc --offload-arch=gfx950 -O3 tools/clockprobe/simdmap.hip -o tools/clockprobe/simdmap && tools/clockprobe/simdmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
    __builtin_amdgcn_s_sleep(100);
}
int main() {
    unsigned* d;
    hipMalloc(&d, 1024 * 16 * 4);
    for (int nw : {4, 8, 13, 16}) {
        hipMemset(d, 0xff, 1024 * 16 * 4);
        hipLaunchKernelGGL(probe, dim3(512), dim3(nw * 64), 0, 0, d);
        hipDeviceSynchronize();
        std::vector<unsigned> h(1024 * 16);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        for (int b : {0, 1, 2, 300}) {
            printf("nw=%2d block %3d: simd of wave 0..%d:", nw, b, nw - 1);
            for (int w = 0; w < nw; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
            printf("   (cu %u se %u, wave slots:", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
            for (int w = 0; w < nw; ++w) printf(" %u", h[b * 16 + w] & 15);
            printf(")\n");
        }
    }
    return 0;
}
