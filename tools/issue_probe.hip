// Micro-benchmark (not part of the product): what does one more instruction of a given class cost a wave that is issuing
// bf16 MFMAs back to back on gfx950 -- (a) in the wave's OWN stream, one wave per SIMD (a 512-register kernel's situation),
// (b) issued by the PARTNER wave of the SIMD (an 8-wave workgroup's situation)?  Decides where the fused GRU's vector work,
// LDS reads and LDS-DMA pieces should live.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void;

#define SB() __builtin_amdgcn_sched_barrier(0)

enum Fill { NONE = 0, FMA = 1, TRANS = 2, DSREAD = 3, DMA = 4, GLOAD = 5, PERM = 6, PKFMA = 7, GSTORE = 8, MIX = 9, MIXDMA = 10, MIXALL = 11 };

struct Ctx {
    float f[8];
    unsigned u[4];
    f32x4 ld[4];
    const float* gsrc;
    float* gdst;
    unsigned lds_off;        // per-lane LDS byte address (ds_read)
    unsigned lds_base;       // wave-uniform LDS byte address (LDS-DMA destination)
    unsigned voff;
};

template <int KIND>
__device__ __forceinline__ void filler(Ctx& c, int i) {
    if constexpr (KIND == FMA) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c.f[i & 7]) : "v"(1.0001f), "v"(0.5f));
    } else if constexpr (KIND == TRANS) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(c.f[i & 7]));
    } else if constexpr (KIND == PERM) {
        asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(c.u[i & 3]) : "v"(c.u[(i + 1) & 3]), "v"(0x07060302u));
    } else if constexpr (KIND == PKFMA) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2& p = *reinterpret_cast<f32x2*>(&c.f[(i & 3) * 2]);
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
    } else if constexpr (KIND == DSREAD) {
        asm volatile("ds_read_b128 %0, %1" : "+v"(c.ld[i & 3]) : "v"(c.lds_off + (unsigned)(i & 15) * 1024u));
    } else if constexpr (KIND == DMA) {
        const unsigned l = __builtin_amdgcn_readfirstlane(c.lds_base + (unsigned)(i & 15) * 1024u);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(c.voff), "s"(c.gsrc + (size_t)(i & 63) * 256) : "memory");
    } else if constexpr (KIND == GLOAD) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(c.ld[i & 3]) : "v"(c.voff + (unsigned)(i & 63) * 1024u), "s"(c.gsrc) : "memory");
    } else if constexpr (KIND == MIX || KIND == MIXDMA || KIND == MIXALL) {
        // the fused GRU's side work per 32x32x16 MFMA, roughly: 3 plain VALU; every 2nd MFMA a weight-fragment ds_read_b128 and a
        // transcendental; MIXDMA: every 8th an LDS-DMA KiB; MIXALL: also every 16th a global load and every 32nd a store
        filler<FMA>(c, i); filler<PERM>(c, i + 1); filler<FMA>(c, i + 2);
        if (i % 2 == 1) { filler<DSREAD>(c, i); filler<TRANS>(c, i); }
        if (KIND != MIX && i % 8 == 7) filler<DMA>(c, i);
        if (KIND == MIXALL && i % 16 == 11) filler<GLOAD>(c, i);
        if (KIND == MIXALL && i % 32 == 19) filler<GSTORE>(c, i);
    } else if constexpr (KIND == GSTORE) {
        asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(c.voff + (unsigned)(i & 63) * 1024u), "v"(c.ld[i & 3]), "s"(c.gdst) : "memory");
    }
}

// SHAPE 32: v_mfma_f32_32x32x16_bf16 (32 clocks), 16: v_mfma_f32_16x16x32_bf16 (16 clocks).  NACC independent accumulators,
// DEP consecutive MFMAs on the same accumulator (the split form chains 6 products per accumulator).
// Every EVERY-th MFMA is followed by NFILL fillers of KIND.
template <int SHAPE, int NACC, int DEP, int KIND, int NFILL, int EVERY, bool do_mfma, bool do_fill>
__device__ __forceinline__ void mfma_stream(int n_outer, Ctx& c, float& sink) {
    const u32x4 xa = {0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, xa), b = __builtin_bit_cast(bf16x8, xa);
    f32x16 acc32[SHAPE == 32 ? NACC : 1];
    f32x4 acc16[SHAPE == 16 ? NACC : 1];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        if constexpr (SHAPE == 32) acc32[i] = f32x16{0}; else acc16[i] = f32x4{0, 0, 0, 0};
    }
    for (int o = 0; o < n_outer; ++o) {
#pragma unroll
        for (int k = 0; k < NACC * DEP; ++k) {
            const int ai = k / DEP;
            if constexpr (do_mfma) {
                if constexpr (SHAPE == 32) acc32[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[ai], 0, 0, 0);
                else acc16[ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[ai], 0, 0, 0);
            }
            SB();
            if (do_fill && (k % EVERY) == EVERY - 1) {
#pragma unroll
                for (int j = 0; j < NFILL; ++j) filler<KIND>(c, k * NFILL + j);
            }
            SB();
        }
        if constexpr (KIND == DSREAD || KIND == GLOAD || KIND >= MIX) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        if constexpr (SHAPE == 32) sink += acc32[i][0] + acc32[i][7]; else sink += acc16[i].x;
    }
}

// role of (first half | second half) of the workgroup's waves: 0 idle, 1 mfma only, 2 filler only, 3 mfma + fillers
template <int NWAVES, int SHAPE, int NACC, int DEP, int KIND, int NFILL, int EVERY>
__global__ __launch_bounds__(NWAVES * 64) void probe(int role_a, int role_b, int n_outer, const float* gsrc, float* gdst, float* out,
                                                     unsigned long long* t) {
    extern __shared__ __attribute__((aligned(16))) float lds[];            // 100 KiB+: one workgroup per CU
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool second = wave >= NWAVES / 2 && NWAVES == 8;
    int role = second ? role_b : role_a;
    Ctx c;
    for (int i = 0; i < 8; ++i) c.f[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 4; ++i) { c.u[i] = threadIdx.x * 77u + i; c.ld[i] = f32x4{0, 0, 0, 0}; }
    auto uni = [](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));   // (the builtin returns int)
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    c.gsrc = uni(gsrc + (size_t)blockIdx.x * 65536);
    c.gdst = const_cast<float*>(uni(gdst + (size_t)blockIdx.x * 65536 + wave * 16384 / 4));
    c.lds_base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void*)lds + (unsigned)(threadIdx.x >> 6) * 16384u);
    c.lds_off = c.lds_base + (unsigned)lane * 16u;
    c.voff = (unsigned)lane * 16u;
    float sink = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role & 4) __builtin_amdgcn_s_setprio(3);
    role &= 3;
    if (role == 1) mfma_stream<SHAPE, NACC, DEP, KIND, NFILL, EVERY, true, false>(n_outer, c, sink);
    else if (role == 2) mfma_stream<SHAPE, NACC, DEP, KIND, NFILL, EVERY, false, true>(n_outer, c, sink);
    else if (role == 3) mfma_stream<SHAPE, NACC, DEP, KIND, NFILL, EVERY, true, true>(n_outer, c, sink);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) { t[wave * 2] = t1 - t0; t[wave * 2 + 1] = t2 - t0; }
    for (int i = 0; i < 8; ++i) sink += c.f[i];
    for (int i = 0; i < 4; ++i) sink += c.u[i] * 1e-30f + c.ld[i].x * 1e-30f;
    out[blockIdx.x * NWAVES * 64 + threadIdx.x] = sink;
}

struct Bufs { float *gsrc, *gdst, *out; unsigned long long* t; };

template <int NWAVES, int SHAPE, int NACC, int DEP, int KIND, int NFILL, int EVERY>
void run(const Bufs& b, const char* what, int role_a, int role_b, int n_outer) {
    auto k = probe<NWAVES, SHAPE, NACC, DEP, KIND, NFILL, EVERY>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    std::vector<unsigned long long> h(16);
    printf("%-58s ", what); fflush(stdout);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(256), dim3(NWAVES * 64), 128 * 1024, 0, role_a, role_b, n_outer, b.gsrc, b.gdst, b.out, b.t);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); fflush(stdout); return; }
    }
    hipMemcpy(h.data(), b.t, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const int n_mfma = n_outer * NACC * DEP;
    const int wb = NWAVES == 8 ? 4 : 0;
    printf("waves=%d roles=%d|%d  mfma/wave=%5d fill/wave=%5d : waveA %7llu clk (%.1f/mfma)  waveB %7llu  until-barrier %7llu (%.1f/mfma)\n",
           NWAVES, role_a, role_b, n_mfma, n_mfma / EVERY * NFILL, h[0], (double)h[0] / n_mfma, h[wb * 2], h[1], (double)h[1] / n_mfma);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    Bufs b;
    hipMalloc(&b.gsrc, 256 * 65536 * sizeof(float) + (1 << 20)); hipMalloc(&b.gdst, 256 * 65536 * sizeof(float) + (1 << 20));
    hipMalloc(&b.out, 256 * 512 * sizeof(float)); hipMalloc(&b.t, 16 * sizeof(unsigned long long));
    hipMemset(b.gsrc, 0, 256 * 65536 * sizeof(float));
    const int N = 100;
    printf("== ONE wave per SIMD (4-wave workgroup), 32x32x16 bf16, 6 dependent products per accumulator, 4 accumulators (24 MFMAs per outer) ==\n");
    run<4, 32, 4, 6, NONE, 0, 1>(b, "32x32x16 bare", 1, 0, N);
    run<4, 32, 8, 1, NONE, 0, 1>(b, "32x32x16 bare, 8 independent accumulators", 1, 0, N);
    run<4, 32, 4, 6, FMA, 2, 1>(b, "+2 v_fma per MFMA", 3, 0, N);
    run<4, 32, 4, 6, FMA, 4, 1>(b, "+4 v_fma per MFMA", 3, 0, N);
    run<4, 32, 4, 6, FMA, 6, 1>(b, "+6 v_fma per MFMA", 3, 0, N);
    run<4, 32, 4, 6, FMA, 8, 1>(b, "+8 v_fma per MFMA", 3, 0, N);
    run<4, 32, 4, 6, PERM, 4, 1>(b, "+4 v_perm per MFMA", 3, 0, N);
    run<4, 32, 4, 6, PKFMA, 2, 1>(b, "+2 v_pk_fma per MFMA", 3, 0, N);
    run<4, 32, 4, 6, TRANS, 1, 1>(b, "+1 v_exp per MFMA", 3, 0, N);
    run<4, 32, 4, 6, TRANS, 2, 1>(b, "+2 v_exp per MFMA", 3, 0, N);
    run<4, 32, 4, 6, TRANS, 1, 2>(b, "+1 v_exp per 2 MFMA", 3, 0, N);
    run<4, 32, 4, 6, DSREAD, 1, 2>(b, "+1 ds_read_b128 per 2 MFMA", 3, 0, N);
    run<4, 32, 4, 6, DSREAD, 1, 1>(b, "+1 ds_read_b128 per MFMA", 3, 0, N);
    run<4, 32, 4, 6, DMA, 1, 8>(b, "+1 LDS-DMA KiB per 8 MFMA", 3, 0, N);
    run<4, 32, 4, 6, DMA, 1, 4>(b, "+1 LDS-DMA KiB per 4 MFMA", 3, 0, N);
    run<4, 32, 4, 6, GLOAD, 1, 8>(b, "+1 global_load_dwordx4 per 8 MFMA", 3, 0, N);
    run<4, 32, 4, 6, GLOAD, 1, 4>(b, "+1 global_load_dwordx4 per 4 MFMA", 3, 0, N);
    run<4, 32, 4, 6, GSTORE, 1, 8>(b, "+1 global_store_dwordx4 per 8 MFMA", 3, 0, N);
    run<4, 32, 4, 6, MIX, 1, 1>(b, "GRU-like mix (3 VALU; /2: ds_read + exp)", 3, 0, N);
    run<4, 32, 4, 6, MIXDMA, 1, 1>(b, "GRU-like mix + LDS-DMA KiB per 8", 3, 0, N);
    run<4, 32, 4, 6, MIXALL, 1, 1>(b, "GRU-like mix + DMA/8 + gload/16 + gstore/32", 3, 0, N);
    run<4, 32, 4, 6, FMA, 4, 1>(b, "fillers alone: 4 v_fma x n", 2, 0, N);
    run<4, 32, 4, 6, TRANS, 1, 1>(b, "fillers alone: 1 v_exp x n", 2, 0, N);
    run<4, 32, 4, 6, DMA, 1, 8>(b, "fillers alone: LDS-DMA KiB x n/8", 2, 0, N);
    run<4, 32, 4, 6, GLOAD, 1, 8>(b, "fillers alone: global_load x n/8", 2, 0, N);
    printf("== ONE wave per SIMD, 16x16x32 bf16, 6 dependent products per accumulator, 7 accumulators ==\n");
    run<4, 16, 7, 6, NONE, 0, 1>(b, "16x16x32 bare", 1, 0, N);
    run<4, 16, 7, 6, FMA, 1, 1>(b, "+1 v_fma per MFMA", 3, 0, N);
    run<4, 16, 7, 6, FMA, 2, 1>(b, "+2 v_fma per MFMA", 3, 0, N);
    run<4, 16, 7, 6, FMA, 3, 1>(b, "+3 v_fma per MFMA", 3, 0, N);
    run<4, 16, 7, 6, TRANS, 1, 2>(b, "+1 v_exp per 2 MFMA", 3, 0, N);
    run<4, 16, 7, 6, DSREAD, 1, 2>(b, "+1 ds_read_b128 per 2 MFMA", 3, 0, N);
    run<4, 16, 7, 6, DMA, 1, 16>(b, "+1 LDS-DMA KiB per 16 MFMA", 3, 0, N);
    printf("== TWO waves per SIMD (8-wave workgroup): waves 0-3 = A, waves 4-7 = B (same SIMDs) ==\n");
    run<8, 16, 7, 6, NONE, 0, 1>(b, "16x16x32: A mfma | B idle", 1, 0, N);
    run<8, 16, 7, 6, NONE, 0, 1>(b, "16x16x32: A mfma | B mfma", 1, 1, N);
    run<8, 16, 7, 6, FMA, 2, 1>(b, "16x16x32: A mfma | B 2 v_fma per (A's) MFMA", 1, 2, N);
    run<8, 16, 7, 6, FMA, 2, 1>(b, "16x16x32: A idle | B 2 v_fma x n", 0, 2, N);
    run<8, 16, 7, 6, PERM, 2, 1>(b, "16x16x32: A mfma | B 2 v_perm", 1, 2, N);
    run<8, 16, 7, 6, TRANS, 1, 2>(b, "16x16x32: A mfma | B v_exp per 2", 1, 2, N);
    run<8, 16, 7, 6, TRANS, 1, 2>(b, "16x16x32: A idle | B v_exp per 2", 0, 2, N);
    run<8, 16, 7, 6, DMA, 1, 16>(b, "16x16x32: A mfma | B LDS-DMA per 16", 1, 2, N);
    run<8, 16, 7, 6, DSREAD, 1, 2>(b, "16x16x32: A mfma | B ds_read per 2", 1, 2, N);
    run<8, 16, 7, 6, FMA, 2, 1>(b, "16x16x32: A mfma+2fma | B mfma+2fma", 3, 3, N);
    run<8, 16, 7, 6, MIXDMA, 1, 2>(b, "16x16x32: A,B both mfma + mix/2 (+DMA)", 3, 3, N);
    printf("-- who wins the issue port: waves 0-3 (A) are the OLDER waves of their SIMDs; role +4 = s_setprio 3 --\n");
    run<8, 16, 7, 6, FMA, 4, 1>(b, "16x16x32: A 4 v_fma x n | B mfma", 2, 1, N);
    run<8, 16, 7, 6, FMA, 4, 1>(b, "16x16x32: A 4 v_fma x n | B mfma PRIO", 2, 5, N);
    run<8, 16, 7, 6, FMA, 4, 1>(b, "16x16x32: A mfma | B 4 v_fma x n", 1, 2, N);
    run<8, 16, 7, 6, FMA, 4, 1>(b, "16x16x32: A mfma PRIO | B 4 v_fma x n", 5, 2, N);
    run<8, 16, 7, 6, TRANS, 1, 1>(b, "16x16x32: A v_exp x n | B mfma", 2, 1, N);
    run<8, 16, 7, 6, TRANS, 1, 1>(b, "16x16x32: A v_exp x n | B mfma PRIO", 2, 5, N);
    run<8, 16, 7, 6, MIX, 1, 1>(b, "16x16x32: A mix x n (no mfma) | B mfma", 2, 1, N);
    run<8, 16, 7, 6, MIX, 1, 1>(b, "16x16x32: A mix x n (no mfma) | B mfma PRIO", 2, 5, N);
    run<8, 32, 4, 6, NONE, 0, 1>(b, "32x32x16: A mfma | B idle", 1, 0, N);
    run<8, 32, 4, 6, FMA, 4, 1>(b, "32x32x16: A mfma | B 4 v_fma per MFMA", 1, 2, N);
    run<8, 32, 4, 6, FMA, 4, 1>(b, "32x32x16: A idle | B 4 v_fma x n", 0, 2, N);
    run<8, 32, 4, 6, TRANS, 1, 1>(b, "32x32x16: A mfma | B v_exp per MFMA", 1, 2, N);
    run<8, 32, 4, 6, DMA, 1, 8>(b, "32x32x16: A mfma | B LDS-DMA per 8", 1, 2, N);
    return 0;
}
