// GEMMs of the backward pass of one propagation timestep (what TF autodiff derives from chem_tensorflow_sparse.py:153-216
// through optimizer.compute_gradients, chem_tensorflow.py:184), hand-written for gfx950:
//
//   ggnn_xty_f32      dW = X^T dY       [M,K]^T [M,N] -> [K,N], M ~ 1e5 rows reduced into a small matrix.  X is a list of
//                     column segments (the [x | incoming | h] concat is never materialised), optionally row-gathered
//                     (edge-weight gradients: X rows = h[pair_node[r]]), optionally batched over row ranges (one output per
//                     edge type).  Rows are split over the whole GPU; partial products are added in a FIXED order.
//   ggnn_colsum_f32   db = 1^T dY       deterministic two-phase column sum
//   ggnn_gru_bwd_dx_cand_f32 / _gates_f32
//                     dX = dY W^T on the FP32-MFMA GEMM core (ggnn_gemm.hpp) with the GRU gate algebra in the epilogue, so
//                     that neither the [V,K] product nor r*h / dpr round-trip through HBM as separate passes.
//
// ggnn_xty_f32 design (FP32 MFMA 16x16x4, 157 TF peak):  the contraction runs over ROWS, so both operands are needed as
// "4 consecutive rows x 16 consecutive columns" fragments.  Row slabs (32 or 64 rows) of X and dY are brought into LDS by LDS-DMA
// (global_load_lds, no staging registers, two slabs in flight) in their global row-major layout, with a row pitch == 16 or 48
// (mod 64 floats) so that the four rows a ds_read_b32 operand fetch touches sit in disjoint banks.
// A workgroup is 16 waves, one per (SIMD s, group i): the <= 16 column tiles of X (rows of dW) and the <= 16 column tiles of dY
// are each cut into 4 groups of <= 4 tiles, and wave 4*i + s multiplies X group i with dY group (i + s) % 4 -- every (X group,
// dY group) pair exactly once, and each SIMD gets one wave of every X group and of every dY group, so the MFMA load of the four
// SIMDs differs by at most one tile product (13 x 13 tiles: 43/42/42/42; the earlier one-wave-per-X-tile layout: 52/39/39/39).
// A wave reads MT + NT <= 8 operand registers for MT * NT <= 16 MFMAs per 4-row step.
#include "ggnn_gemm.hpp"
#include "ggnn_split.hpp"

namespace ggnn {

constexpr int kXtyMaxBatch = 64;
constexpr int kXtyWaves = 16;
constexpr int kXtyMaxI = 4;                    // decoded DMA instructions per wave, operand and slab
#ifndef GGNN_XTY_TIMELINE
#define GGNN_XTY_TIMELINE 0
#endif

// 16-byte DMA sources that are not operand data: the chunk of X that holds the "ones" column K (Kout == K + 1: row K of the
// product is then the column sum of dY, the bias gradient), and the X chunks of slab rows past the end of a workgroup's row range
// (zeros: those rows then add nothing, whatever dY row the clamped address fetched) -- so the MFMA loop needs no masking at all.
__device__ const float kXtyOnesChunk[4] = {1.0f, 0.0f, 0.0f, 0.0f};
__device__ const float kXtyZeroChunk[4] = {0.0f, 0.0f, 0.0f, 0.0f};

struct XtyArgs {
    const float* X[4]; int ldx[4]; int nseg; int Dseg;     // X(v, k) = X[k / Dseg][row(v) * ldx + k % Dseg]
    const int* x_rows;                                      // optional: row(v) = x_rows[v]
    const float* Y; int ldy;                                // dY(v, n)
    float* part;                                            // partial products [workgroup row][Kout][N]
    int K, N, nbatch;
    int Kout;                                               // K, or K + 1: row K of the product is 1^T dY (column sums = bias gradient)
    int row_off[kXtyMaxBatch + 1];                          // batch b owns rows row_off[b] .. row_off[b+1]-1
    int wg_off[kXtyMaxBatch + 1];                           // ... and is worked by workgroup rows (blockIdx.y) wg_off[b] .. wg_off[b+1]-1
    int kb_tiles, n_tiles;                                  // 16-column tiles of X per workgroup (blockIdx.x = K block); of dY
    int pitch_x, pitch_y;                                   // LDS row pitches in floats
    unsigned long long* tdbg;                               // (debug) s_memtime stamps of workgroup (0,0): tools/xty_timeline.py
};

static inline int xty_pitch(int width) {                    // smallest pitch >= width, a multiple of 16, == 16 or 48 (mod 64)
    int p = (width + 15) / 16 * 16;
    while (p % 64 != 16 && p % 64 != 48) p += 16;
    return p;
}

// group g of 4 over n tiles: tiles [off, off + cnt)
__host__ __device__ inline void xty_group(int n, int g, int& off, int& cnt) {
    const int base = n >> 2, rem = n & 3;
    cnt = base + (g < rem ? 1 : 0);
    off = g * base + (g < rem ? g : rem);
}

// Slab loader: a slab is a linear run of 16-byte chunks (row-major, `cpr` chunks per row); DMA instruction j of a wave moves
// chunks [64*(j*16 + wave), +64).  chunk -> (row in slab, segment, column in segment) is fixed per lane, so it is decoded once
// and kept packed: bits 0-15 column within the segment, 16-17 segment, 18-24 row in slab, 31 = padding chunk.
template <bool GATHER, int ROWS, int MTM, int NTM>   // MTM / NTM: tiles of the largest X / dY group (the others have one less or as many)
__global__ __launch_bounds__(kXtyWaves * 64) void xty_kernel(XtyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float slab[];     // [2][ROWS][pitch_x] | [2][ROWS][pitch_y]
    constexpr int nw = kXtyWaves;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    int batch = 0;
    while (batch + 1 < a.nbatch && (int)blockIdx.y >= a.wg_off[batch + 1]) ++batch;
    const int split = (int)blockIdx.y - a.wg_off[batch], splits = a.wg_off[batch + 1] - a.wg_off[batch];
    const int rb = a.row_off[batch], re = a.row_off[batch + 1];
    int rows_per = (re - rb + splits - 1) / splits;
    rows_per = (rows_per + ROWS - 1) / ROWS * ROWS;
    const int r_beg = rb + split * rows_per;
    const int r_end = min(re, r_beg + rows_per);
    const int kcol0 = blockIdx.x * a.kb_tiles * 16;                  // first X column of this workgroup
    const int px = a.pitch_x, py = a.pitch_y;
    float* sx = slab;
    float* sy = slab + 2 * ROWS * px;
    float* out = a.part + (size_t)blockIdx.y * a.Kout * a.N;

    // this wave's tile groups
    int kt0, my_mt, nt0, my_nt;
    xty_group(a.kb_tiles, wave >> 2, kt0, my_mt);
    xty_group(a.n_tiles, ((wave >> 2) + (wave & 3)) & 3, nt0, my_nt);

    // stamps of waves 0 and 15 of workgroup (0,0): [which][0] start, [1] first slab landed, [2+i] slab i consumed, [62] loop done, [63] stored
    int stamp_i = 2;
#if GGNN_XTY_TIMELINE          // (build with -DGGNN_XTY_TIMELINE=1 for tools/xty_timeline.py: the stamps cost scalar registers in the hot loop)
#define GGNN_XT(K) if (a.tdbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == nw - 1) && (K) < 64) \
        a.tdbg[(wave ? 64 : 0) + (K)] = __builtin_amdgcn_s_memtime();
#else
#define GGNN_XT(K)
#endif
    GGNN_XT(0)

    // ---- per-lane decode of the DMA chunks ---------------------------------------------------------------------------
    constexpr int MAXI = kXtyMaxI;
    const int cprx = px / 4, cpry = py / 4;
    const int nix = ROWS * cprx / 64, niy = ROWS * cpry / 64;      // 1-KiB instructions per slab
    unsigned mx[MAXI], my[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int ix = j * nw + wave;
        mx[j] = 0x80000000u; my[j] = 0x80000000u;
        if (ix < nix) {
            const int c = ix * 64 + lane, row = c / cprx, q = c - row * cprx;
            int col = kcol0 + 4 * q;
            const bool ones = a.Kout > a.K && col == a.K && 4 * q < a.kb_tiles * 16;      // the chunk [1 0 0 0] of the ones column
            if (!(4 * q < a.kb_tiles * 16 && col < a.K)) col = 0;            // padding chunk: any valid address
            const int seg = col / a.Dseg, within = col - seg * a.Dseg;
            mx[j] = (unsigned)within | ((unsigned)seg << 16) | ((unsigned)row << 18) | (ones ? 0x40000000u : 0u);
        }
        if (ix < niy) {
            const int c = ix * 64 + lane, row = c / cpry, q = c - row * cpry;
            my[j] = (unsigned)(4 * q < a.N ? 4 * q : 0) | ((unsigned)row << 18);
        }
    }
    // X row of slab row r (row gather for the edge-weight gradients); rows past the range are clamped here and zeroed in issue()
    auto xrow = [&](int r0, unsigned m) -> int {
        int r = r0 + (int)((m >> 18) & 127u); r = r < r_end ? r : r_end - 1;
        return r;
    };
    int rid[GATHER ? MAXI : 1];                                     // (GATHER) X rows of the NEXT slab to be issued
    auto fetch_rows = [&](int r0) {
        if constexpr (GATHER) {
#pragma unroll
            for (int j = 0; j < MAXI; ++j) rid[j] = (mx[j] >> 31) ? 0 : a.x_rows[xrow(r0 < r_end ? r0 : r_beg, mx[j])];
        }
    };
    // segment bases / strides as VALUES in scalar registers (readfirstlane: otherwise the compiler selects the ADDRESS
    // of the kernel argument and loads through it, see below)
    auto sgpr_ptr = [](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    const float* X0 = sgpr_ptr(a.X[0]); const float* X1 = sgpr_ptr(a.X[1]);
    const float* X2 = sgpr_ptr(a.X[2]); const float* X3 = sgpr_ptr(a.X[3]);
    const int L0 = __builtin_amdgcn_readfirstlane(a.ldx[0]), L1 = __builtin_amdgcn_readfirstlane(a.ldx[1]);
    const int L2 = __builtin_amdgcn_readfirstlane(a.ldx[2]), L3 = __builtin_amdgcn_readfirstlane(a.ldx[3]);
    const float* Yb = sgpr_ptr(a.Y);
    const int ldy = __builtin_amdgcn_readfirstlane(a.ldy);
    auto issue = [&](int buf, int r0) {                             // uses rid[] (fetched one slab ahead)
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            if (!(mx[j] >> 31)) {
                const unsigned m = mx[j];
                // (the segment is a per-lane value: select base and stride with compares -- indexing the kernel-argument
                //  arrays with it makes the compiler spill them to memory, and the loads that fetch them back sit in the
                //  same in-order queue as the DMA: every slab instruction then waited for the previous one)
                const int seg = (int)((m >> 16) & 3u);
                const float* xb = seg == 0 ? X0 : (seg == 1 ? X1 : (seg == 2 ? X2 : X3));
                const int xl = seg == 0 ? L0 : (seg == 1 ? L1 : (seg == 2 ? L2 : L3));
                const int xr = GATHER ? rid[GATHER ? j : 0] : xrow(r0, m);
                const float* src = xb + (size_t)xr * xl + (m & 0xFFFFu);
                if (m & 0x40000000u) src = kXtyOnesChunk;
                if (r0 + (int)((m >> 18) & 127u) >= r_end) src = kXtyZeroChunk;
                float* dst = sx + buf * ROWS * px + (size_t)(j * nw + wave) * 256;      // 1 KiB = 256 floats per instruction
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            if (!(my[j] >> 31)) {
                const unsigned m = my[j];
                int r = r0 + (int)((m >> 18) & 127u); r = r < r_end ? r : r_end - 1;
                const float* src = Yb + (size_t)r * ldy + (m & 0xFFFFu);
                float* dst = sy + buf * ROWS * py + (size_t)(j * nw + wave) * 256;
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
            }
        }
    };

    // The whole slab loop is instantiated per (tiles of X, tiles of dY) of a wave -- MTM or MTM-1, NTM or NTM-1, or none -- and
    // selected ONCE: with run-time tile counts the compiler cannot count the outstanding LDS reads (s_waitcnt lgkmcnt(0) at every
    // step) and branches around every MFMA, which left the matrix pipe idle a quarter of the time; with the selection inside the
    // loop the accumulators pass through phi copies at every iteration and spill.
    auto run = [&](auto mt_c, auto nt_c) {
        constexpr int MT = decltype(mt_c)::value, NT = decltype(nt_c)::value;
        constexpr bool ON = MT > 0 && NT > 0;
        f32x4 acc[ON ? MT : 1][ON ? NT : 1];
#pragma unroll
        for (int mt = 0; mt < (ON ? MT : 1); ++mt)
#pragma unroll
            for (int nt = 0; nt < (ON ? NT : 1); ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int cls = wave >> 2;
        if (r_beg < r_end) {
            int buf = 0;
            fetch_rows(r_beg);
            issue(0, r_beg);
            fetch_rows(r_beg + ROWS);
            __syncthreads();                                              // (vmcnt(0) + barrier: slab 0 has landed)
            GGNN_XT(1)
            for (int r0 = r_beg; r0 < r_end; r0 += ROWS) {
                // The DMA of the next slab is issued from INSIDE the MFMA steps, staggered: the four waves of a SIMD (waves c, c+4,
                // c+8, c+12 -> classes 0..3) issue their share at step 0, 1, 2, 3 (x2 with 64-row slabs), so that at any moment at
                // most one wave of a SIMD is computing addresses instead of feeding the matrix pipe.  (All sixteen waves issuing at
                // the top of the slab left the pipe idle for ~2k of 15k clocks; the last class still has half the slab to land.)
                const bool has_next = r0 + ROWS < r_end;
                if (has_next && !ON) { issue(buf ^ 1, r0 + ROWS); fetch_rows(r0 + 2 * ROWS); }
                if (stamp_i < 22) { GGNN_XT(20 + stamp_i) }          // [22+i] slab i: top of the slab
                if constexpr (ON) {
                    // MT x NT tile products per 4-row step, operands of step s+1 read under the MFMAs of step s
                    const float* bx = sx + buf * ROWS * px + kt0 * 16 + li;
                    const float* by = sy + buf * ROWS * py + nt0 * 16 + li;
                    float xa[2][MT], yb[2][NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) xa[0][mt] = bx[kq * px + mt * 16];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) yb[0][nt] = by[kq * py + nt * 16];
#pragma unroll
                    for (int s = 0; s < ROWS / 4; ++s) {
                        if (s + 1 < ROWS / 4) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) xa[(s + 1) & 1][mt] = bx[(4 * (s + 1) + kq) * px + mt * 16];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) yb[(s + 1) & 1][nt] = by[(4 * (s + 1) + kq) * py + nt * 16];
                        }
                        if (s % (ROWS / 32) == 0 && s / (ROWS / 32) < 4) {
                            if (has_next && cls == s / (ROWS / 32)) { issue(buf ^ 1, r0 + ROWS); fetch_rows(r0 + 2 * ROWS); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s & 1][mt], yb[s & 1][nt], acc[mt][nt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (stamp_i < 22) { GGNN_XT(40 + stamp_i) }          // [42+i] slab i: MFMAs issued
                __syncthreads();                                          // slab[buf] consumed by all waves, slab[buf^1] landed
                buf ^= 1;
                if (stamp_i < 22) { GGNN_XT(stamp_i) }
                ++stamp_i;
            }
        }
        GGNN_XT(62)
        // accumulator tile (mt, nt): lane (li, kq) holds dW[k0 + 4*kq + e][16*(nt0+nt) + li], e = 0..3  (a workgroup without rows
        // still writes its zeros: the reduction adds every workgroup row)
        if constexpr (ON) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int k0 = kcol0 + (kt0 + mt) * 16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = 16 * (nt0 + nt) + li;
                    if (n < a.N) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = k0 + 4 * kq + e;
                            if (k < a.Kout) out[(size_t)k * a.N + n] = acc[mt][nt][e];
                        }
                    }
                }
            }
        }
        GGNN_XT(63)
    };
    using I0 = std::integral_constant<int, 0>;
    using IM = std::integral_constant<int, MTM>; using IM1 = std::integral_constant<int, MTM - 1>;
    using IN = std::integral_constant<int, NTM>; using IN1 = std::integral_constant<int, NTM - 1>;
    // (the last K block can be short of whole groups: a wave whose X group is empty or smaller than MTM - 1 has no tiles there --
    //  the host plan keeps that from happening for groups that do hold output rows, see xty_plan)
    if (my_mt == MTM && my_nt == NTM) run(IM{}, IN{});
    else if (my_mt == MTM - 1 && my_nt == NTM) run(IM1{}, IN{});
    else if (my_mt == MTM && my_nt == NTM - 1) run(IM{}, IN1{});
    else if (my_mt == MTM - 1 && my_nt == NTM - 1) run(IM1{}, IN1{});
    else run(I0{}, I0{});
#undef GGNN_XT
}


// ---- X^T dY in 3-way split form (ggnn_split.hpp): both operands are activations, so both are split in flight ----------------------
// Same slabs (raw f32 rows by LDS-DMA, same pitches), same partial products and reduction as xty_kernel; what differs:
//   * the contraction step is 32 rows (v_mfma_f32_16x16x32_bf16): lane group g of an operand tile holds rows 4j + g, j = 0..7, of
//     the step -- for a fixed j the four groups read four consecutive rows: the bank pattern of the f32 kernel's 4-row steps;
//   * a tile's 8 values per lane are split into three bf16 planes in registers (44 vector instructions per tile and step) and a
//     tile PAIR takes 6 MFMAs of 16 clocks where the f32 kernel issues 8 of 32 clocks;
//   * 8 waves of 256 registers instead of 16 of 128: wave w owns X group w >> 1 (<= MTM tiles; its planes are split once per
//     step and stay resident, 48 registers) and TWO dY groups, (w >> 1) + 2 (w & 1) and the next one (mod 4) (<= 2 NTM tiles,
//     split tile by tile): <= 28 tile pairs = 112 accumulator registers.  Every (X group, dY group) pair exactly once; the two
//     waves of a SIMD (w, w + 4) carry 46 / 45 / 39 / 39 pairs at 13 x 13 tiles.
//   * the slab DMA goes out through inline assembly (the compiler does not see an LDS-DMA in flight, see dma_image_asm) and is
//     waited for explicitly before the barrier that publishes the slab.
#ifndef GGNN_XTY_INTER_MAX
#define GGNN_XTY_INTER_MAX 24
#endif
constexpr int kXtySplitWaves = 8;
constexpr int kXtySplitMaxI = 8;

template <bool GATHER, int ROWS, int MTM, int NTM>
__global__ __launch_bounds__(kXtySplitWaves * 64) void xty_split_kernel(XtyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float slab[];     // [2][ROWS][pitch_x] | [2][ROWS][pitch_y]
    constexpr int nw = kXtySplitWaves;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    int batch = 0;
    while (batch + 1 < a.nbatch && (int)blockIdx.y >= a.wg_off[batch + 1]) ++batch;
    const int split = (int)blockIdx.y - a.wg_off[batch], splits = a.wg_off[batch + 1] - a.wg_off[batch];
    const int rb = a.row_off[batch], re = a.row_off[batch + 1];
    int rows_per = (re - rb + splits - 1) / splits;
    rows_per = (rows_per + ROWS - 1) / ROWS * ROWS;
    const int r_beg = rb + split * rows_per;
    const int r_end = min(re, r_beg + rows_per);
    const int kcol0 = blockIdx.x * a.kb_tiles * 16;
    const int px = a.pitch_x, py = a.pitch_y;
    float* sx = slab;
    float* sy = slab + 2 * ROWS * px;
    float* out = a.part + (size_t)blockIdx.y * a.Kout * a.N;

    // this wave's tile groups
    int kt0, my_mt, nta0, cnta, ntb0, cntb;
    xty_group(a.kb_tiles, wave >> 1, kt0, my_mt);
    const int ga = ((wave >> 1) + 2 * (wave & 1)) & 3;
    xty_group(a.n_tiles, ga, nta0, cnta);
    xty_group(a.n_tiles, (ga + 1) & 3, ntb0, cntb);
    const int my_nt = cnta + cntb;

    // ---- per-lane decode of the DMA chunks (as xty_kernel, 8 waves) --------------------------------------------------------------
    constexpr int MAXI = kXtySplitMaxI;
    const int cprx = px / 4, cpry = py / 4;
    const int nix = ROWS * cprx / 64, niy = ROWS * cpry / 64;
    unsigned mx[MAXI], my[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int ix = j * nw + wave;
        mx[j] = 0x80000000u; my[j] = 0x80000000u;
        if (ix < nix) {
            const int c = ix * 64 + lane, row = c / cprx, q = c - row * cprx;
            int col = kcol0 + 4 * q;
            const bool ones = a.Kout > a.K && col == a.K && 4 * q < a.kb_tiles * 16;
            if (!(4 * q < a.kb_tiles * 16 && col < a.K)) col = 0;
            const int seg = col / a.Dseg, within = col - seg * a.Dseg;
            mx[j] = (unsigned)within | ((unsigned)seg << 16) | ((unsigned)row << 18) | (ones ? 0x40000000u : 0u);
        }
        if (ix < niy) {
            const int c = ix * 64 + lane, row = c / cpry, q = c - row * cpry;
            my[j] = (unsigned)(4 * q < a.N ? 4 * q : 0) | ((unsigned)row << 18);
        }
    }
    auto xrow = [&](int r0, unsigned m) __attribute__((always_inline)) -> int {
        int r = r0 + (int)((m >> 18) & 127u); r = r < r_end ? r : r_end - 1;
        return r;
    };
    int rid[GATHER ? MAXI : 1];
    auto fetch_rows = [&](int r0) __attribute__((always_inline)) {
        if constexpr (GATHER) {
#pragma unroll
            for (int j = 0; j < MAXI; ++j) rid[j] = (mx[j] >> 31) ? 0 : a.x_rows[xrow(r0 < r_end ? r0 : r_beg, mx[j])];
        }
    };
    auto sgpr_ptr = [](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
    };
    const float* X0 = sgpr_ptr(a.X[0]); const float* X1 = sgpr_ptr(a.X[1]);
    const float* X2 = sgpr_ptr(a.X[2]); const float* X3 = sgpr_ptr(a.X[3]);
    const int L0 = __builtin_amdgcn_readfirstlane(a.ldx[0]), L1 = __builtin_amdgcn_readfirstlane(a.ldx[1]);
    const int L2 = __builtin_amdgcn_readfirstlane(a.ldx[2]), L3 = __builtin_amdgcn_readfirstlane(a.ldx[3]);
    const float* Yb = sgpr_ptr(a.Y);
    const int ldy = __builtin_amdgcn_readfirstlane(a.ldy);
    // one 1-KiB LDS-DMA instruction with per-lane source addresses, through inline assembly (M0 = the wave's LDS destination)
    auto dma1 = [&](const float* src, float* dst) __attribute__((always_inline)) {
        const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void*)dst);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(l), "v"(src) : "memory");
    };
    auto issue = [&](int buf, int r0) __attribute__((always_inline)) {                             // uses rid[] (fetched one slab ahead)
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            if (!(mx[j] >> 31)) {
                const unsigned m = mx[j];
                const int seg = (int)((m >> 16) & 3u);
                const float* xb = seg == 0 ? X0 : (seg == 1 ? X1 : (seg == 2 ? X2 : X3));
                const int xl = seg == 0 ? L0 : (seg == 1 ? L1 : (seg == 2 ? L2 : L3));
                const int xr = GATHER ? rid[GATHER ? j : 0] : xrow(r0, m);
                const float* src = xb + (size_t)xr * xl + (m & 0xFFFFu);
                if (m & 0x40000000u) src = kXtyOnesChunk;
                if (r0 + (int)((m >> 18) & 127u) >= r_end) src = kXtyZeroChunk;
                dma1(src, sx + buf * ROWS * px + (size_t)(j * nw + wave) * 256);
            }
        }
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            if (!(my[j] >> 31)) {
                const unsigned m = my[j];
                int r = r0 + (int)((m >> 18) & 127u); r = r < r_end ? r : r_end - 1;
                dma1(Yb + (size_t)r * ldy + (m & 0xFFFFu), sy + buf * ROWS * py + (size_t)(j * nw + wave) * 256);
            }
        }
    };

    // the slab loop, instantiated per (X tiles, dY tiles) of a wave and selected once (see xty_kernel)
    auto run = [&](auto mt_c, auto nt_c) __attribute__((always_inline)) {
        constexpr int MT = decltype(mt_c)::value, NT = decltype(nt_c)::value;
        constexpr bool ON = MT > 0 && NT > 0;
        f32x4 acc[ON ? MT : 1][ON ? NT : 1];
#pragma unroll
        for (int mt = 0; mt < (ON ? MT : 1); ++mt)
#pragma unroll
            for (int nt = 0; nt < (ON ? NT : 1); ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // dY tile t of this wave: column offset within the slab row
        auto ycol = [&](int t) __attribute__((always_inline)) { return (t < cnta ? nta0 + t : ntb0 + (t - cnta)) * 16; };
        if (r_beg < r_end) {
            int buf = 0;
            fetch_rows(r_beg);
            issue(0, r_beg);
            fetch_rows(r_beg + ROWS);
            dma_wait();
            __syncthreads();
            for (int r0 = r_beg; r0 < r_end; r0 += ROWS) {
                const bool has_next = r0 + ROWS < r_end;
                if (has_next) { issue(buf ^ 1, r0 + ROWS); fetch_rows(r0 + 2 * ROWS); }
                if constexpr (ON) {
                    const float* bx = sx + buf * ROWS * px + kt0 * 16 + li;
                    const float* by = sy + buf * ROWS * py + li;
                    // 8 values of a tile column (rows 32 s + 4 j + kq) -> planes
                    auto tile_planes = [&](const float* col, int pitch, int s, u32x4& hi, u32x4& mid, u32x4& lo) __attribute__((always_inline)) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = col[(size_t)(32 * s + 4 * j + kq) * pitch];
                        unsigned h[4], m[4], l[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], h[q], m[q], l[q]);
                        hi = u32x4{h[0], h[1], h[2], h[3]}; mid = u32x4{m[0], m[1], m[2], m[3]}; lo = u32x4{l[0], l[1], l[2], l[3]};
                    };
#pragma unroll
                    for (int s = 0; s < ROWS / 32; ++s) {
                        u32x4 xh[MT], xm[MT], xl[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) tile_planes(bx + mt * 16, px, s, xh[mt], xm[mt], xl[mt]);
                        // dY tile nt+1 is read and split while the 6 MT MFMAs of tile nt issue: no fence between the two, the
                        // scheduler interleaves the vector instructions with the (long-latency) MFMAs; one fence per tile
                        u32x4 yh, ym, yl, zh, zm, zl;
                        tile_planes(by + ycol(0), py, s, yh, ym, yl);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (nt + 1 < NT) tile_planes(by + ycol(nt + 1), py, s, zh, zm, zl);
                            // six products per tile pair, product-major over the X tiles (consecutive MFMAs hit different accumulators)
#define GGNN_XS(XP, YP) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(XP[mt], YP, acc[mt][nt]);
                            GGNN_XS(xl, yh) GGNN_XS(xm, ym) GGNN_XS(xm, yh) GGNN_XS(xh, yl) GGNN_XS(xh, ym) GGNN_XS(xh, yh)
#undef GGNN_XS
                            yh = zh; ym = zm; yl = zl;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                dma_wait();
                __syncthreads();                                          // slab[buf] consumed by all waves, slab[buf^1] landed
                buf ^= 1;
            }
        }
        if constexpr (ON) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int k0 = kcol0 + (kt0 + mt) * 16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = ycol(nt) + li;
                    if (n < a.N) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = k0 + 4 * kq + e;
                            if (k < a.Kout) out[(size_t)k * a.N + n] = acc[mt][nt][e];
                        }
                    }
                }
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IM = std::integral_constant<int, MTM>; using IM1 = std::integral_constant<int, MTM - 1>;
    using IN = std::integral_constant<int, 2 * NTM>; using IN1 = std::integral_constant<int, 2 * NTM - 1>;
    using IN2 = std::integral_constant<int, 2 * NTM - 2>;
    if (my_mt == MTM && my_nt == 2 * NTM) run(IM{}, IN{});
    else if (my_mt == MTM && my_nt == 2 * NTM - 1) run(IM{}, IN1{});
    else if (my_mt == MTM && my_nt == 2 * NTM - 2) run(IM{}, IN2{});
    else if (my_mt == MTM - 1 && my_nt == 2 * NTM) run(IM1{}, IN{});
    else if (my_mt == MTM - 1 && my_nt == 2 * NTM - 1) run(IM1{}, IN1{});
    else if (my_mt == MTM - 1 && my_nt == 2 * NTM - 2) run(IM1{}, IN2{});
    else run(I0{}, I0{});
}

// ---- X^T dY in split form with the operands split ONCE, into LDS planes (round 4) -------------------------------------------------
// xty_split_kernel above splits an operand tile in EVERY wave that multiplies it (484 vector instructions and 88 ds_read_b32 per
// wave and 32-row step next to 168 MFMAs; on this chip the vector and matrix instructions of a SIMD's two waves do not overlap --
// DESIGN.md K3 -- so the splits were ~40 % of the loop).  Here a step's 32 rows x (X columns + dY columns) are fetched by ONE thread
// per (8-row group, 4-column quad): eight 16-byte loads (coalesced: consecutive lanes = consecutive quads of a row), a step ahead,
// straight into registers -- no f32 slab in LDS --, split once (44 vector instructions per column) and written to LDS as the
// three bf16 planes in MFMA operand layout
//     plane[p][g][column slot][8 x bf16]      (g = 8-row group of the step: lane (i, g) of a 16-column tile reads ONE ds_read_b128)
// double-buffered: 2 x 192 (XC + YC) bytes (XC = 16 kb_tiles, YC = 16 n_tiles: 156 KiB at 13 + 13 tiles).  Column c of tile t sits in
// slot 16 t + ((c + t) & 15): a rotation per tile, so that the 8 lanes of a ds_write_b128 phase -- quads of two neighbouring tiles,
// same column within the quad -- hit 32 distinct banks; a tile read still covers its 16 slots once.  A quad is data (pointer to its
// first row, byte stride), the chunk {1, 0, 0, 0} with stride 0 (the ones column K of Kout == K + 1: the bias gradient) or the zero
// chunk (padding columns; every quad for rows past the range) -- K, N and the segment width are multiples of 4: no mixed quads, no
// per-kind control flow.  The next step's columns are cut into the MFMA stream (one per tile or two) for waves with <= 24 tile pairs.
// Tile groups, partial products and the reduction are xty_split_kernel's (same XtyArgs / plan); within a 32-row step the rows sit
// in other MFMA k slots, so results agree with it to f32 rounding, not bit for bit.
// Ungathered X only (the training step's GRU weight gradients; the row-gathered edge-weight products gain nothing from this form:
// tools/attic/xty_planes_gather.patch.txt).
template <int MTM, int NTM>
__global__ __launch_bounds__(kXtySplitWaves * 64) void xty_planes_kernel(XtyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float slab[];     // [2][3 planes][4 groups][XC + YC] x 16 bytes
    constexpr int nw = kXtySplitWaves, NTH = nw * 64, ROWS = 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;
    int batch = 0;
    while (batch + 1 < a.nbatch && (int)blockIdx.y >= a.wg_off[batch + 1]) ++batch;
    const int split = (int)blockIdx.y - a.wg_off[batch], splits = a.wg_off[batch + 1] - a.wg_off[batch];
    const int rb = a.row_off[batch], re = a.row_off[batch + 1];
    int rows_per = (re - rb + splits - 1) / splits;
    rows_per = (rows_per + ROWS - 1) / ROWS * ROWS;
    const int r_beg = rb + split * rows_per;
    const int r_end = min(re, r_beg + rows_per);
    const int kcol0 = blockIdx.x * a.kb_tiles * 16;
    const int XC = a.kb_tiles * 16, YC = a.n_tiles * 16, CT = XC + YC;      // CT <= NTH (one quad per thread and 8-row group)
    const unsigned plane_b = 4u * (unsigned)CT * 16u;                  // bytes of one plane (4 groups)
    const unsigned buf_b = 3u * plane_b;
    float* out = a.part + (size_t)blockIdx.y * a.Kout * a.N;

    // this wave's tile groups (xty_split_kernel's)
    int kt0, my_mt, nta0, cnta, ntb0, cntb;
    xty_group(a.kb_tiles, wave >> 1, kt0, my_mt);
    const int ga = ((wave >> 1) + 2 * (wave & 1)) & 3;
    xty_group(a.n_tiles, ga, nta0, cnta);
    xty_group(a.n_tiles, (ga + 1) & 3, ntb0, cntb);
    const int my_nt = cnta + cntb;

    // ---- producer side: thread t < CT owns quad (t % (CT / 4)) of 8-row group t / (CT / 4) ------------------------------------------
    const char* const zero_p = reinterpret_cast<const char*>(kXtyZeroChunk);
    const int QT = CT >> 2;
    const bool uon = tid < 4 * QT;
    const int ug = uon ? tid / QT : 0;                                  // 8-row group
    const int uc = uon ? 4 * (tid - ug * QT) : 0;                       // first column of the quad, in [0, CT)
    // slots of the quad's four columns: region base + 16 tile + ((column in tile + tile) & 15)
    const int ureg = uc < XC ? 0 : XC, utile = (uc - ureg) >> 4;
    const unsigned udbase = ((unsigned)ug * (unsigned)CT + (unsigned)ureg + 16u * (unsigned)utile) * 16u;
    const unsigned urot = (unsigned)(((uc - ureg) & 15) + utile);       // (+ j, & 15: slot within the tile)
    const char* urow = zero_p; unsigned ustride = 0u;
    if (uon) {
        const size_t row0 = (size_t)(r_beg + 8 * ug);
        if (uc < XC) {
            const int col = kcol0 + uc;
            if (col < a.K) {
                const int seg = col / a.Dseg;
                ustride = (unsigned)a.ldx[seg] * 4u;
                urow = reinterpret_cast<const char*>(a.X[seg] + (col - seg * a.Dseg)) + row0 * ustride;
            } else if (col == a.K && a.Kout > a.K) urow = reinterpret_cast<const char*>(kXtyOnesChunk);
        } else if (uc - XC < a.N) {
            ustride = (unsigned)a.ldy * 4u;
            urow = reinterpret_cast<const char*>(a.Y + (uc - XC)) + row0 * ustride;
        }
    }
    const bool wave_on = wave * 64 < 4 * QT;                            // (wave-uniform: this wave owns quads)
    f32x4 stage[8];                                                     // the quad's 8 rows of a step, fetched a step ahead
    // FULL: all 32 rows of the step lie inside the range; else rows >= r_end read the zero chunk (every operand: 0 x 0)
    auto fetch = [&](int r0, auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        if (wave_on) {
            const char* q = urow;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const char* src = q;
                if constexpr (!FULL) { if (r0 + 8 * ug + j >= r_end) src = zero_p; }
                stage[j] = *(const __attribute__((address_space(1))) f32x4*)src;       // (global_load_dwordx4, not a flat access)
                q += ustride;
            }
            urow += (size_t)32 * ustride;
        }
    };
    auto fetch_step = [&](int r0) __attribute__((always_inline)) {
        if (r0 + ROWS <= r_end) fetch(r0, std::true_type{}); else fetch(r0, std::false_type{});
    };
    // column k of the quad: split its eight staged values, write its planes of slab[buf]
    auto put_col = [&](int buf, auto k_c) __attribute__((always_inline)) {
        constexpr int k = decltype(k_c)::value;
        if (wave_on) {
            char* base = reinterpret_cast<char*>(slab) + (size_t)buf * buf_b;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) split_pair(stage[2 * q][k], stage[2 * q + 1][k], h[q], m[q], l[q]);
            if (uon) {
                const unsigned d = udbase + ((urot + (unsigned)k) & 15u) * 16u;
                *reinterpret_cast<u32x4*>(base + d) = u32x4{h[0], h[1], h[2], h[3]};
                *reinterpret_cast<u32x4*>(base + plane_b + d) = u32x4{m[0], m[1], m[2], m[3]};
                *reinterpret_cast<u32x4*>(base + 2u * plane_b + d) = u32x4{l[0], l[1], l[2], l[3]};
            }
        }
    };
    auto put = [&](int buf) __attribute__((always_inline)) {
        put_col(buf, std::integral_constant<int, 0>{}); put_col(buf, std::integral_constant<int, 1>{});
        put_col(buf, std::integral_constant<int, 2>{}); put_col(buf, std::integral_constant<int, 3>{});
    };
    // column k of the NEXT step's quad (behind the last one: the fetch of the step after it), cut into the MFMA stream below -- in one
    // wave's stream ~1 vector instruction per MFMA issues in the MFMAs' shadow (tools/issue_probe.hip)
    auto advance_col = [&](int buf, int r0, bool has_next, auto k_c) __attribute__((always_inline)) {
        if (has_next) {
            put_col(buf ^ 1, k_c);
            if (decltype(k_c)::value == 3 && r0 + 2 * ROWS < r_end) fetch_step(r0 + 2 * ROWS);
        }
    };

    auto run = [&](auto mt_c, auto nt_c) __attribute__((always_inline)) {
        constexpr int MT = decltype(mt_c)::value, NT = decltype(nt_c)::value;
        constexpr bool ON = MT > 0 && NT > 0;
        f32x4 acc[ON ? MT : 1][ON ? NT : 1];
#pragma unroll
        for (int mt = 0; mt < (ON ? MT : 1); ++mt)
#pragma unroll
            for (int nt = 0; nt < (ON ? NT : 1); ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto ytile = [&](int t) __attribute__((always_inline)) { return t < cnta ? nta0 + t : ntb0 + (t - cnta); };
        // the MFMAs of one step over slab[buf]; INTER: the next step's columns cut in (one per PU tiles)
        auto mma_step = [&](int buf, int r0, bool has_next, auto inter_c) __attribute__((always_inline)) {
            constexpr bool INTER = decltype(inter_c)::value;
            if constexpr (ON) {
                const char* pb = reinterpret_cast<const char*>(slab) + (size_t)buf * buf_b + (unsigned)kq * (unsigned)CT * 16u;
                // tile T of a region (X: base 0, dY: base XC): lane li reads slot 16 T + ((li + T) & 15)
                auto planes = [&](int reg, int T, u32x4& hi, u32x4& mid, u32x4& lo) __attribute__((always_inline)) {
                    const char* q = pb + (unsigned)(reg + 16 * T + ((li + T) & 15)) * 16u;
                    hi = *reinterpret_cast<const u32x4*>(q); mid = *reinterpret_cast<const u32x4*>(q + plane_b);
                    lo = *reinterpret_cast<const u32x4*>(q + 2u * plane_b);
                };
                u32x4 xh[MT], xm[MT], xl[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) planes(0, kt0 + mt, xh[mt], xm[mt], xl[mt]);
                u32x4 yh, ym, yl, zh, zm, zl;
                planes(XC, ytile(0), yh, ym, yl);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (nt + 1 < NT) planes(XC, ytile(nt + 1), zh, zm, zl);
                    // six products per tile pair, product-major over the X tiles (consecutive MFMAs hit different accumulators)
#define GGNN_XS(XP, YP) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_bf16(XP[mt], YP, acc[mt][nt]);
                    GGNN_XS(xl, yh) GGNN_XS(xm, ym) GGNN_XS(xm, yh) GGNN_XS(xh, yl) GGNN_XS(xh, ym) GGNN_XS(xh, yh)
#undef GGNN_XS
                    yh = zh; ym = zm; yl = zl;
                    if constexpr (INTER) {
                        constexpr int PU = NT >= 8 ? 2 : 1;              // tiles per column slice
                        if (nt % PU == 0 && nt / PU < 4) {
                            if (nt / PU == 0) advance_col(buf, r0, has_next, std::integral_constant<int, 0>{});
                            if (nt / PU == 1) advance_col(buf, r0, has_next, std::integral_constant<int, 1>{});
                            if (nt / PU == 2) advance_col(buf, r0, has_next, std::integral_constant<int, 2>{});
                            if (nt / PU == 3) advance_col(buf, r0, has_next, std::integral_constant<int, 3>{});
                        }
                    }
                }
                if constexpr (INTER && NT < 4) {                         // (fewer tiles than columns: the rest behind the last tile)
                    if (NT <= 1) advance_col(buf, r0, has_next, std::integral_constant<int, 1>{});
                    if (NT <= 2) advance_col(buf, r0, has_next, std::integral_constant<int, 2>{});
                    if (NT <= 3) advance_col(buf, r0, has_next, std::integral_constant<int, 3>{});
                }
            }
        };
        if (r_beg < r_end) {
            int buf = 0;
            fetch_step(r_beg);
            put(0);
            if (r_beg + ROWS < r_end) fetch_step(r_beg + ROWS);
            __syncthreads();
            constexpr bool INTER = ON && MT * NT <= GGNN_XTY_INTER_MAX;   // (the widest waves have no registers for the staged values beside their tiles)
            for (int r0 = r_beg; r0 < r_end; r0 += ROWS) {
                const bool has_next = r0 + ROWS < r_end;
                // the next step's planes (its values were fetched a step ago) and the fetch of the step after it: a wave with tiles
                // to spare spreads them over its MFMA stream (advance_col), the others do them here
                if constexpr (!INTER) { if (has_next) { put(buf ^ 1); if (r0 + 2 * ROWS < r_end) fetch_step(r0 + 2 * ROWS); } }
                mma_step(buf, r0, has_next, std::integral_constant<bool, INTER>{});
                __syncthreads();                           // slab[buf] consumed by all waves, slab[buf ^ 1] written by all
                buf ^= 1;
            }
        }
        if constexpr (ON) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int k0 = kcol0 + (kt0 + mt) * 16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = ytile(nt) * 16 + li;
                    if (n < a.N) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = k0 + 4 * kq + e;
                            if (k < a.Kout) out[(size_t)k * a.N + n] = acc[mt][nt][e];
                        }
                    }
                }
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IM = std::integral_constant<int, MTM>; using IM1 = std::integral_constant<int, MTM - 1>;
    using IN = std::integral_constant<int, 2 * NTM>; using IN1 = std::integral_constant<int, 2 * NTM - 1>;
    using IN2 = std::integral_constant<int, 2 * NTM - 2>;
    if (my_mt == MTM && my_nt == 2 * NTM) run(IM{}, IN{});
    else if (my_mt == MTM && my_nt == 2 * NTM - 1) run(IM{}, IN1{});
    else if (my_mt == MTM && my_nt == 2 * NTM - 2) run(IM{}, IN2{});
    else if (my_mt == MTM - 1 && my_nt == 2 * NTM) run(IM1{}, IN{});
    else if (my_mt == MTM - 1 && my_nt == 2 * NTM - 1) run(IM1{}, IN1{});
    else if (my_mt == MTM - 1 && my_nt == 2 * NTM - 2) run(IM1{}, IN2{});
    else run(I0{}, I0{});
}

// C[b][i] = sum over the workgroup rows of batch b of part[row][i], in row order.  With a separate bias destination the K weight rows
// go to C [nbatch][K][N] and the ones row to Cb [nbatch][N]; accumulate: the sums are ADDED to what the destinations hold (the
// gradient buffers of the training step: one launch less per product, and no torch add on the side stream).
struct XtyReduceArgs { int wg_off[kXtyMaxBatch + 1]; };
// A block = 32 consecutive output elements x the 8 chains: thread (c, q) sums the partials of the workgroup rows p == q (mod 8) of
// its element in row order, the chains are combined pairwise through LDS -- the same sums in the same order as one thread walking
// eight interleaved chains (the first form of this kernel), but 8x the threads: the reduction of a 201 x 200 product over 256
// workgroup rows is 41 MB read by 1256 blocks instead of 157 (it had become the largest single line of the training step's profile:
// 24 launches of 31-59 us next to the fused GRU backward, for 11 us of memory traffic).
__global__ __launch_bounds__(256) void xty_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, float* __restrict__ Cb,
                                                         int KN, int KwN, int nbatch, int accumulate, XtyReduceArgs ra) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + c;
    const bool on = i < (long long)KN * nbatch;
    float s = 0.f;
    int b = 0, j = 0;
    if (on) {
        b = (int)(i / KN); j = (int)(i - (long long)b * KN);
        const int S = ra.wg_off[b + 1] - ra.wg_off[b];
        const float* src = part + (size_t)ra.wg_off[b] * KN + j;
        int p = q;
        for (; p + 24 < S; p += 32) {                      // four loads in flight per thread; the adds stay in row order
            const float v0 = src[(size_t)p * KN], v1 = src[(size_t)(p + 8) * KN], v2 = src[(size_t)(p + 16) * KN], v3 = src[(size_t)(p + 24) * KN];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; p < S; p += 8) s += src[(size_t)p * KN];
    }
    red[q][c] = s;
    __syncthreads();
    if (q == 0 && on) {
        const float v = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
        float* dst = (Cb && j >= KwN) ? Cb + (size_t)b * (KN - KwN) + (j - KwN) : C + (size_t)b * (Cb ? KwN : KN) + j;
        *dst = accumulate ? *dst + v : v;
    }
}

struct XtyPlan { int rows, kb_tiles, n_tiles, kblocks, px, py, wg_rows; size_t lds; int wg_off[kXtyMaxBatch + 1]; };

// Workgroup rows (one 16-wave workgroup per CU and K block) are dealt to the batches in proportion to their row counts, at least
// one per non-empty batch and at most one per 4 slabs of rows; an empty batch gets none (its product is the empty sum, 0).
static XtyPlan xty_plan(const int* row_off, int nbatch, int Kout, int N) {
    XtyPlan p{};
    const int ktiles = (Kout + 15) / 16;
    p.kblocks = (ktiles + 15) / 16;
    p.kb_tiles = (ktiles + p.kblocks - 1) / p.kblocks;
    p.n_tiles = (N + 15) / 16;
    p.px = xty_pitch(p.kb_tiles * 16); p.py = xty_pitch(p.n_tiles * 16);
    // 64-row slabs when two of them fit the 160 KiB of LDS (half the barriers, and a slab's MFMA time then covers the latency of the
    // next slab's DMA also for the narrow edge-weight products), else 32
    const int mtm = (p.kb_tiles + 3) / 4, ntm = (p.n_tiles + 3) / 4;
    p.rows = (size_t)2 * 64 * (p.px + p.py) * sizeof(float) <= (size_t)160 * 1024 && mtm + ntm <= 6 ? 64 : 32;
    static const int rows_env = [] { const char* e = getenv("GGNN_XTY_ROWS"); return e ? atoi(e) : 0; }();   // (experiments)
    if (rows_env == 32) p.rows = 32;
    p.lds = (size_t)2 * p.rows * (p.px + p.py) * sizeof(float);
    long long total = 0;
    for (int b = 0; b < nbatch; ++b) total += row_off[b + 1] - row_off[b];
    int target = num_cus() / p.kblocks;
    if (target < 1) target = 1;
    p.wg_off[0] = 0;
    for (int b = 0; b < nbatch; ++b) {
        const long long m = row_off[b + 1] - row_off[b];
        long long w = total > 0 ? (long long)target * m / total : 0;
        const long long cap = (m + 4 * p.rows - 1) / (4 * p.rows);
        if (w > cap) w = cap;
        if (w < 1 && m > 0) w = 1;
        p.wg_off[b + 1] = p.wg_off[b] + (int)w;
    }
    p.wg_rows = p.wg_off[nbatch];
    return p;
}

template <bool GATHER, int ROWS, int MTM, int NTM>
static int launch_xty(const XtyArgs& a, const XtyPlan& p, float* C, float* Cb, int accumulate, hipStream_t st) {
    static std::atomic<unsigned long long> lds_ok{0};
    if (p.wg_rows > 0) {
        if (p.lds > 64 * 1024) GGNN_CHECK_HIP((allow_dynamic_lds(&xty_kernel<GATHER, ROWS, MTM, NTM>, p.lds, lds_ok)));
        hipLaunchKernelGGL((xty_kernel<GATHER, ROWS, MTM, NTM>), dim3(p.kblocks, p.wg_rows), dim3(kXtyWaves * 64), p.lds, st, a);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    XtyReduceArgs ra;
    for (int b = 0; b <= kXtyMaxBatch; ++b) ra.wg_off[b] = a.wg_off[b <= a.nbatch ? b : a.nbatch];
    const long long total = (long long)a.Kout * a.N * a.nbatch;
    hipLaunchKernelGGL(xty_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const float*)a.part, C, Cb, a.Kout * a.N,
                       a.K * a.N, a.nbatch, accumulate, ra);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <int MTM, int NTM>
static int launch_xty_planes(const XtyArgs& a, const XtyPlan& p, float* C, float* Cb, int accumulate, hipStream_t st) {
    static std::atomic<unsigned long long> lds_ok{0};
    const size_t lds = (size_t)2 * 192 * (size_t)(16 * (p.kb_tiles + p.n_tiles));
    if (p.wg_rows > 0) {
        if (lds > 64 * 1024) GGNN_CHECK_HIP((allow_dynamic_lds(&xty_planes_kernel<MTM, NTM>, lds, lds_ok)));
        hipLaunchKernelGGL((xty_planes_kernel<MTM, NTM>), dim3(p.kblocks, p.wg_rows), dim3(kXtySplitWaves * 64), lds, st, a);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    XtyReduceArgs ra;
    for (int b = 0; b <= kXtyMaxBatch; ++b) ra.wg_off[b] = a.wg_off[b <= a.nbatch ? b : a.nbatch];
    const long long total = (long long)a.Kout * a.N * a.nbatch;
    hipLaunchKernelGGL(xty_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const float*)a.part, C, Cb, a.Kout * a.N,
                       a.K * a.N, a.nbatch, accumulate, ra);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

template <bool GATHER, int ROWS, int MTM, int NTM>
static int launch_xty_split(const XtyArgs& a, const XtyPlan& p, float* C, float* Cb, int accumulate, hipStream_t st) {
    static std::atomic<unsigned long long> lds_ok{0};
    if (p.wg_rows > 0) {
        if (p.lds > 64 * 1024) GGNN_CHECK_HIP((allow_dynamic_lds(&xty_split_kernel<GATHER, ROWS, MTM, NTM>, p.lds, lds_ok)));
        hipLaunchKernelGGL((xty_split_kernel<GATHER, ROWS, MTM, NTM>), dim3(p.kblocks, p.wg_rows), dim3(kXtySplitWaves * 64), p.lds, st, a);
        GGNN_CHECK_HIP(hipGetLastError());
    }
    XtyReduceArgs ra;
    for (int b = 0; b <= kXtyMaxBatch; ++b) ra.wg_off[b] = a.wg_off[b <= a.nbatch ? b : a.nbatch];
    const long long total = (long long)a.Kout * a.N * a.nbatch;
    hipLaunchKernelGGL(xty_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, (const float*)a.part, C, Cb, a.Kout * a.N,
                       a.K * a.N, a.nbatch, accumulate, ra);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// ---- deterministic column sums -----------------------------------------------------------------------------------------
constexpr int kColsumBlocks = 256;

// block b sums rows b, b + gridDim, ... for all columns (thread = float4 column group x row lane); partial [blocks][N]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ Y, int ldy, int M, int N, float* __restrict__ part) {
    extern __shared__ float red[];                                     // [rows_per_block_iter][N]
    const int N4 = N >> 2;
    const int lanes_per_row = N4;                                       // one float4 per thread per row
    const int rows_par = 256 / lanes_per_row > 0 ? 256 / lanes_per_row : 1;
    const int rl = threadIdx.x / lanes_per_row, c4 = threadIdx.x % lanes_per_row;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par) {
        for (int r = blockIdx.x * rows_par + rl; r < M; r += gridDim.x * rows_par)
            s += *reinterpret_cast<const f32x4*>(Y + (size_t)r * ldy + 4 * c4);
        *reinterpret_cast<f32x4*>(red + rl * N + 4 * c4) = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        float t = 0.f;
        for (int q = 0; q < rows_par; ++q) t += red[q * N + i];
        part[(size_t)blockIdx.x * N + i] = t;
    }
}

__global__ void colsum_final_kernel(const float* __restrict__ part, int nblocks, int N, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // eight independent chains, fixed combination order
    int b = 0;
    for (; b + 8 <= nblocks; b += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += part[(size_t)(b + q) * N + i];
    }
    for (int q = 0; b < nblocks; ++b, ++q) s[q] += part[(size_t)b * N + i];
    out[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

}  // namespace ggnn

using namespace ggnn;

extern "C" size_t ggnn_xty_workspace_bytes(int M_max, int K, int N, int nbatch) {
    if (M_max <= 0 || K <= 0 || N <= 0 || nbatch <= 0) return 256;
    // workgroup rows: <= one per CU and K block, plus the one-per-non-empty-batch minimum (sized for the ones-row form)
    return (size_t)(num_cus() + nbatch) * (K + 1) * N * sizeof(float) + 256;
}

extern "C" int ggnn_xty_f32(const float* const* x_segs, int nseg, int Dseg, const int32_t* ldx, const int32_t* x_rows,
                            const float* Y, int ldy, float* C, int K, int N, int ones_row, const int32_t* row_off, int nbatch,
                            void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    return ggnn_xty_acc_f32(x_segs, nseg, Dseg, ldx, x_rows, Y, ldy, C, nullptr, 0, K, N, ones_row, row_off, nbatch, ws, ws_bytes, stream);
}

extern "C" int ggnn_xty_acc_f32(const float* const* x_segs, int nseg, int Dseg, const int32_t* ldx, const int32_t* x_rows,
                                const float* Y, int ldy, float* C, float* Cb, int accumulate, int K, int N, int ones_row,
                                const int32_t* row_off, int nbatch, void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    const int Kout = ones_row ? K + 1 : K;
    GGNN_CHECK_ARG(!Cb || ones_row, "a bias destination needs the ones row");
    GGNN_CHECK_ARG(nseg >= 1 && nseg <= 4 && Dseg > 0 && Dseg % 4 == 0 && K == nseg * Dseg, "X is nseg <= 4 segments of Dseg columns (K = %d, nseg = %d, Dseg = %d)", K, nseg, Dseg);
    GGNN_CHECK_ARG(N > 0 && N % 4 == 0 && N <= 256, "N = %d must be a multiple of 4, <= 256", N);
    GGNN_CHECK_ARG(nbatch >= 1 && nbatch <= kXtyMaxBatch && row_off && C && ldx, "bad batch description");
    GGNN_CHECK_ARG(ldy >= N && ldy % 4 == 0, "ldy %d", ldy);
    hipStream_t st = (hipStream_t)stream;
    XtyArgs a{};
    int m_max = 0;
    for (int b = 0; b <= nbatch; ++b) {
        a.row_off[b] = row_off[b];
        if (b) { GGNN_CHECK_ARG(row_off[b] >= row_off[b - 1], "row_off not monotone"); m_max = row_off[b] - row_off[b - 1] > m_max ? row_off[b] - row_off[b - 1] : m_max; }
    }
    if (m_max == 0) {                                   // empty sums
        if (!accumulate) {
            GGNN_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)nbatch * (Cb ? K : Kout) * N * sizeof(float), st));
            if (Cb) GGNN_CHECK_HIP(hipMemsetAsync(Cb, 0, (size_t)nbatch * N * sizeof(float), st));
        }
        return GGNN_OK;
    }
    GGNN_CHECK_ARG(x_segs && Y && ws && aligned16(Y) && aligned16(ws), "null or misaligned pointer");
    for (int s = 0; s < nseg; ++s) {
        GGNN_CHECK_ARG(x_segs[s] && aligned16(x_segs[s]) && ldx[s] >= Dseg && ldx[s] % 4 == 0, "X segment %d null, misaligned or bad stride", s);
        a.X[s] = x_segs[s]; a.ldx[s] = ldx[s];
    }
    if (ws_bytes < ggnn_xty_workspace_bytes(m_max, K, N, nbatch)) return fail(GGNN_E_WORKSPACE, "xty workspace too small");
    const XtyPlan p = xty_plan(row_off, nbatch, Kout, N);
    a.Kout = Kout;
    a.nseg = nseg; a.Dseg = Dseg; a.x_rows = x_rows; a.Y = Y; a.ldy = ldy; a.part = static_cast<float*>(ws);
    a.K = K; a.N = N; a.nbatch = nbatch; a.kb_tiles = p.kb_tiles; a.n_tiles = p.n_tiles;
    a.pitch_x = p.px; a.pitch_y = p.py;
    for (int b = 0; b <= kXtyMaxBatch; ++b) a.wg_off[b] = p.wg_off[b <= nbatch ? b : nbatch];
    { const char* e = getenv("GGNN_XTY_TPTR"); a.tdbg = e ? (unsigned long long*)strtoull(e, nullptr, 10) : nullptr; }
    if ((p.rows * p.px / 256 + kXtyWaves - 1) / kXtyWaves > kXtyMaxI || (p.rows * p.py / 256 + kXtyWaves - 1) / kXtyWaves > kXtyMaxI)
        return fail(GGNN_E_UNSUPPORTED, "xty: slab too wide (K=%d N=%d)", K, N);
    const int mtm = (p.kb_tiles + 3) / 4, ntm = (p.n_tiles + 3) / 4;
    const bool g = x_rows != nullptr;
    // split form (K0): the shapes of the training step at D = 100 (the others stay on the f32 kernel); the tile groups of a
    // wave must come out as (MTM or MTM-1) x (2 NTM .. 2 NTM - 2) tiles: true whenever no group is empty
    static const bool xty_split = [] { const char* e = getenv("GGNN_XTY_SPLIT"); return !e || atoi(e) != 0; }();
    if (split_matrix_path() && xty_split && p.kb_tiles >= 4 && p.n_tiles >= 4 && (p.rows * p.px / 256 + kXtySplitWaves - 1) / kXtySplitWaves <= kXtySplitMaxI &&
        (p.rows * p.py / 256 + kXtySplitWaves - 1) / kXtySplitWaves <= kXtySplitMaxI) {
        // operands split once into LDS planes (xty_planes_kernel; GGNN_XTY_PLANES=0: the per-wave split below): ungathered X, both
        // plane buffers within the 160 KiB of LDS, every unit of a step owned by a thread
        static const bool xty_planes = [] { const char* e = getenv("GGNN_XTY_PLANES"); return !e || atoi(e) != 0; }();
        if (xty_planes && !g && (size_t)2 * 192 * 16 * (p.kb_tiles + p.n_tiles) <= (size_t)160 * 1024 &&
            16 * (p.kb_tiles + p.n_tiles) <= kXtySplitWaves * 64) {
#define GGNN_XTYP_CASE(M, Nn) if (mtm == M && ntm == Nn) return launch_xty_planes<M, Nn>(a, p, C, Cb, accumulate, st);
            GGNN_XTYP_CASE(4, 4) GGNN_XTYP_CASE(3, 4) GGNN_XTYP_CASE(4, 2) GGNN_XTYP_CASE(3, 2)
#undef GGNN_XTYP_CASE
        }
#define GGNN_XTYS_CASE(G, R, M, Nn) if (g == G && p.rows == R && mtm == M && ntm == Nn) return launch_xty_split<G, R, M, Nn>(a, p, C, Cb, accumulate, st);
        GGNN_XTYS_CASE(false, 32, 4, 4) GGNN_XTYS_CASE(false, 32, 3, 4) GGNN_XTYS_CASE(false, 64, 4, 2) GGNN_XTYS_CASE(false, 64, 3, 2)
        // (the row-gathered edge-weight products, 2 x 2 tile groups per wave, measure slower in split form -- 52 vs 48 us: with so
        //  few tile pairs per operand tile the split work is not amortised -- and stay on the f32 kernel)
#undef GGNN_XTYS_CASE
    }
#define GGNN_XTY_CASE(G, R, M, Nn) if (g == G && p.rows == R && mtm == M && ntm == Nn) return launch_xty<G, R, M, Nn>(a, p, C, Cb, accumulate, st);
#define GGNN_XTY_ROW32(M) GGNN_XTY_CASE(false, 32, M, 1) GGNN_XTY_CASE(false, 32, M, 2) GGNN_XTY_CASE(false, 32, M, 3) GGNN_XTY_CASE(false, 32, M, 4)
    GGNN_XTY_ROW32(1) GGNN_XTY_ROW32(2) GGNN_XTY_ROW32(3) GGNN_XTY_ROW32(4)
    // 64-row slabs exist for the group shapes whose two slabs can fit the LDS (xty_plan picks them when they do)
    GGNN_XTY_CASE(false, 64, 1, 1) GGNN_XTY_CASE(false, 64, 1, 2) GGNN_XTY_CASE(false, 64, 1, 3) GGNN_XTY_CASE(false, 64, 1, 4)
    GGNN_XTY_CASE(false, 64, 2, 1) GGNN_XTY_CASE(false, 64, 2, 2) GGNN_XTY_CASE(false, 64, 2, 3) GGNN_XTY_CASE(false, 64, 2, 4)
    GGNN_XTY_CASE(false, 64, 3, 1) GGNN_XTY_CASE(false, 64, 3, 2) GGNN_XTY_CASE(false, 64, 3, 3)
    GGNN_XTY_CASE(false, 64, 4, 1) GGNN_XTY_CASE(false, 64, 4, 2)
    // row-gathered X: one segment of <= 128 columns (edge-weight gradients)
    GGNN_XTY_CASE(true, 64, 1, 1) GGNN_XTY_CASE(true, 64, 1, 2) GGNN_XTY_CASE(true, 64, 2, 1) GGNN_XTY_CASE(true, 64, 2, 2)
#undef GGNN_XTY_ROW32
#undef GGNN_XTY_CASE
    return fail(GGNN_E_UNSUPPORTED, "xty: no kernel for %d x %d tile groups (K=%d N=%d)", mtm, ntm, K, N);
}

extern "C" size_t ggnn_colsum_workspace_bytes(int N) { return (size_t)kColsumBlocks * (N > 0 ? N : 1) * sizeof(float) + 256; }

extern "C" int ggnn_colsum_f32(const float* Y, int ldy, int M, int N, float* out, void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(M >= 0 && N > 0 && N % 4 == 0 && N <= 1024 && ldy >= N && ldy % 4 == 0, "bad sizes M=%d N=%d ldy=%d", M, N, ldy);
    GGNN_CHECK_ARG(out && ws, "null pointer");
    if (ws_bytes < ggnn_colsum_workspace_bytes(N)) return fail(GGNN_E_WORKSPACE, "colsum workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) { GGNN_CHECK_HIP(hipMemsetAsync(out, 0, N * sizeof(float), st)); return GGNN_OK; }
    GGNN_CHECK_ARG(Y && aligned16(Y), "null or misaligned pointer");
    const int rows_par = 256 / (N / 4) > 0 ? 256 / (N / 4) : 1;
    GGNN_CHECK_ARG(N / 4 <= 256, "N too large");
    int nb = (M + rows_par - 1) / rows_par;
    if (nb > kColsumBlocks) nb = kColsumBlocks;
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), (size_t)rows_par * N * sizeof(float), st, Y, ldy, M, N, part);
    GGNN_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, st, (const float*)part, nb, N, out);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
