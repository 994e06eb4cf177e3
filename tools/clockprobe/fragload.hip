// How fast does a CU move 16-row x 64-byte fragment pieces?  One wave instruction of 64 x 16 bytes in three lane orders:
//   A  linear      lane l -> 16 B chunk l of 1 KiB contiguous
//   B  fragment    lane (li = l % 16, kq = l / 16) -> row li, bytes 64c + 16kq   (the MFMA operand layout: quads of lanes = 4 ROWS)
//   C  row quads   lane l -> row l / 4, bytes 64c + 16 (l % 4)                   (quads of lanes = 64 contiguous bytes of ONE row)
// over [rows, 100] float32 (row pitch 400 B, 6 pieces of 64 B per row), loads and stores, 8 waves per workgroup, 1 workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/clockprobe/fragload.hip -o tools/clockprobe/fragload && tools/clockprobe/fragload
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool STORE>
__global__ __launch_bounds__(512) void k(float* __restrict__ p, int rows, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles = rows / 16;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int t = blockIdx.x * 8 + wave; t < tiles; t += gridDim.x * 8) {
        char* base = reinterpret_cast<char*>(p) + (size_t)t * 16 * 400;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            unsigned off;
            if (PAT == 0) off = (unsigned)(c * 64 + lane) * 16u;                       // 6 KiB contiguous (= 15.36 rows)
            else if (PAT == 1) off = (unsigned)(lane & 15) * 400u + 64u * c + 16u * (lane >> 4);
            else off = (unsigned)(lane >> 2) * 400u + 64u * c + 16u * (lane & 3);
            f32x4* q = reinterpret_cast<f32x4*>(base + off);
            if (STORE) *q = f32x4{(float)t, (float)c, 1.f, 2.f};
            else s += *q;
        }
    }
    if (!STORE && s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
}

static int g_blocks = 256;
template <int PAT, bool STORE>
static void run(const char* name, float* d, int rows, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, STORE>), dim3(g_blocks), dim3(512), 0, 0, d, rows, sink);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PAT, STORE>), dim3(g_blocks), dim3(512), 0, 0, d, rows, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)(rows / 16) * 6 * 1024;
    printf("%-28s rows %7d, %3d workgroups: %7.1f us  %6.2f TB/s = %5.1f B/clock per workgroup (2.4 GHz)\n", name, rows, g_blocks, ms / reps * 1e3,
           bytes / (ms / reps * 1e-3) / 1e12, bytes / (ms / reps * 1e-3) / g_blocks / 2.4e9);
}

int main() {
    float *d, *sink;
    const int maxrows = 1600000;
    hipMalloc(&d, (size_t)maxrows * 400); hipMalloc(&sink, 64);
    hipMemset(d, 0, (size_t)maxrows * 400);
    for (int nb : {16, 64, 256})
    for (int rows : {100000, 1600000}) {              // 40 MB (Infinity Cache), 640 MB (HBM); 16 / 64 / 256 workgroups: per-CU or shared limit?
        g_blocks = nb;
        run<0, false>("load  A linear", d, rows, sink);
        run<1, false>("load  B fragment (4 rows/quad)", d, rows, sink);
        run<2, false>("load  C row quads", d, rows, sink);
        run<0, true>("store A linear", d, rows, sink);
        run<1, true>("store B fragment", d, rows, sink);
        run<2, true>("store C row quads", d, rows, sink);
    }
    return 0;
}
