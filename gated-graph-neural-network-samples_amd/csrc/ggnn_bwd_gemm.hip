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
// "4 consecutive rows x 16 consecutive columns" fragments.  Row slabs (32 rows) of X and dY are brought into LDS by LDS-DMA
// (global_load_lds, no staging registers, two slabs in flight) in their global row-major layout, with a row pitch == 16 or 48
// (mod 64 floats) so that the four rows a ds_read_b32 operand fetch touches sit in disjoint banks.  Wave w of a workgroup owns
// MT 16-column tiles of X (rows of dW) x ALL N columns (NT tiles): MT + NT operand reads feed MT*NT MFMAs per 4-row step.
#include "ggnn_gemm.hpp"
#include "ggnn_stage.hpp"

namespace ggnn {

constexpr int kXtyRows = 32;                   // rows per LDS slab
constexpr int kXtyMaxBatch = 64;

struct XtyArgs {
    const float* X[4]; int ldx[4]; int nseg; int Dseg;     // X(v, k) = X[k / Dseg][row(v) * ldx + k % Dseg]
    const int* x_rows;                                      // optional: row(v) = x_rows[v]
    const float* Y; int ldy;                                // dY(v, n)
    float* part;                                            // partial products [batch][split][K][N]
    int K, N, nbatch, splits;
    int Kout;                                               // K, or K + 1: row K of the product is 1^T dY (column sums = bias gradient)
    int row_off[kXtyMaxBatch + 1];                          // batch b owns rows row_off[b] .. row_off[b+1]-1
    int kb_tiles;                                           // 16-column tiles of X per workgroup
    int pitch_x, pitch_y;                                   // LDS row pitches in floats
};

static inline int xty_pitch(int width) {                    // smallest pitch >= width, a multiple of 16, == 16 or 48 (mod 64)
    int p = (width + 15) / 16 * 16;
    while (p % 64 != 16 && p % 64 != 48) p += 16;
    return p;
}

// Slab loader: a slab is a linear run of 16-byte chunks (row-major, `cpr` chunks per row); DMA instruction j of a wave moves
// chunks [64*(j*nw + wave), +64).  chunk -> (row in slab, segment, column in segment) is fixed per lane, so it is decoded once
// and kept packed: bits 0-15 column within the segment, 16-17 segment, 18-23 row in slab, 31 = padding chunk.
__host__ __device__ constexpr int xty_max_instr(int mt) { return mt >= 2 ? 8 : 4; }   // decoded DMA instructions per wave and slab

template <int MT, int NT, bool GATHER>
__global__ __launch_bounds__(MT >= 2 ? 512 : 832) void xty_kernel(XtyArgs a) {
    extern __shared__ __attribute__((aligned(16))) float slab[];     // [2][kXtyRows][pitch_x] | [2][kXtyRows][pitch_y]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int batch = blockIdx.z, split = blockIdx.y;
    const int rb = a.row_off[batch], re = a.row_off[batch + 1];
    int rows_per = (re - rb + a.splits - 1) / a.splits;
    rows_per = (rows_per + kXtyRows - 1) / kXtyRows * kXtyRows;
    const int r_beg = rb + split * rows_per;
    const int r_end = min(re, r_beg + rows_per);
    const int kcol0 = blockIdx.x * a.kb_tiles * 16;                  // first X column of this workgroup
    const int px = a.pitch_x, py = a.pitch_y;
    float* sx = slab;
    float* sy = slab + 2 * kXtyRows * px;
    float* out = a.part + ((size_t)batch * a.splits + split) * a.Kout * a.N;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (r_beg < r_end) {
        // ---- per-lane decode of the DMA chunks -----------------------------------------------------------------------
        constexpr int MAXI = xty_max_instr(MT);
        const int cprx = px / 4, cpry = py / 4;
        const int nix = kXtyRows * cprx / 64, niy = kXtyRows * cpry / 64;      // 1-KiB instructions per slab
        unsigned mx[MAXI], my[MAXI];
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            const int ix = j * nw + wave;
            mx[j] = 0x80000000u; my[j] = 0x80000000u;
            if (ix < nix) {
                const int c = ix * 64 + lane, row = c / cprx, q = c - row * cprx;
                int col = kcol0 + 4 * q;
                if (!(4 * q < a.kb_tiles * 16 && col < a.K)) col = 0;            // padding chunk: any valid address
                const int seg = col / a.Dseg, within = col - seg * a.Dseg;
                mx[j] = (unsigned)within | ((unsigned)seg << 16) | ((unsigned)row << 18);
            }
            if (ix < niy) {
                const int c = ix * 64 + lane, row = c / cpry, q = c - row * cpry;
                my[j] = (unsigned)(4 * q < a.N ? 4 * q : 0) | ((unsigned)row << 18);
            }
        }
        // X row of slab row r (row gather for the edge-weight gradients); clamped rows are masked out in the products
        auto xrow = [&](int r0, unsigned m) -> int {
            int r = r0 + (int)((m >> 18) & 63u); r = r < r_end ? r : r_end - 1;
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
                    float* dst = sx + buf * kXtyRows * px + (size_t)(j * nw + wave) * 256;      // 1 KiB = 256 floats per instruction
                    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < MAXI; ++j) {
                if (!(my[j] >> 31)) {
                    const unsigned m = my[j];
                    int r = r0 + (int)((m >> 18) & 63u); r = r < r_end ? r : r_end - 1;
                    const float* src = a.Y + (size_t)r * a.ldy + (m & 0xFFFFu);
                    float* dst = sy + buf * kXtyRows * py + (size_t)(j * nw + wave) * 256;
                    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
                }
            }
        };

        const int ktile0 = wave * MT;                                 // this wave's first X tile inside the workgroup's block
        const bool wave_on = ktile0 < a.kb_tiles && kcol0 + ktile0 * 16 < a.Kout;
        int buf = 0;
        fetch_rows(r_beg);
        issue(0, r_beg);
        fetch_rows(r_beg + kXtyRows);
        __syncthreads();                                              // (vmcnt(0) + barrier: slab 0 has landed)
        for (int r0 = r_beg; r0 < r_end; r0 += kXtyRows) {
            if (r0 + kXtyRows < r_end) { issue(buf ^ 1, r0 + kXtyRows); fetch_rows(r0 + 2 * kXtyRows); }
            if (wave_on) {
                const float* bx = sx + buf * kXtyRows * px + ktile0 * 16 + li;
                const float* by = sy + buf * kXtyRows * py + li;
                const int nvalid = r_end - r0;                        // rows of this slab that exist (>= kXtyRows: all)
                float xa[2][MT], yb[2][NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xa[0][mt] = bx[kq * px + mt * 16];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) yb[0][nt] = by[kq * py + nt * 16];
#pragma unroll
                for (int s = 0; s < kXtyRows / 4; ++s) {
                    if (s + 1 < kXtyRows / 4) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) xa[(s + 1) & 1][mt] = bx[(4 * (s + 1) + kq) * px + mt * 16];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) yb[(s + 1) & 1][nt] = by[(4 * (s + 1) + kq) * py + nt * 16];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bool row_ok = 4 * s + kq < nvalid;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        float xv = xa[s & 1][mt];
                        // the "ones" column K of X (Kout == K + 1): its row of the product is the column sum of dY
                        if (a.Kout > a.K && kcol0 + (ktile0 + mt) * 16 + li == a.K) xv = 1.0f;
                        xv = row_ok ? xv : 0.f;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, yb[s & 1][nt], acc[mt][nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                          // slab[buf] consumed by all waves, slab[buf^1] landed
            buf ^= 1;
        }
    }
    // accumulator tile (mt, nt): lane (li, kq) holds dW[k0 + 4*kq + e][16*nt + li], e = 0..3
    const int ktile0 = wave * MT;
    if (ktile0 < a.kb_tiles) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int k0 = kcol0 + (ktile0 + mt) * 16;
            if ((ktile0 + mt) < a.kb_tiles) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = 16 * nt + li;
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
    }
}

// C[b][i] = sum over the splits of part[b][s][i], in split order
__global__ void xty_reduce_kernel(const float* __restrict__ part, float* __restrict__ C, int KN, int S, int nbatch) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)KN * nbatch) return;
    const int b = (int)(i / KN), j = (int)(i - (long long)b * KN);
    // eight independent chains (split p goes to chain p % 8), combined pairwise: a fixed order with 1/8 of the dependent loads
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = part + (size_t)b * S * KN + j;
    int p = 0;
    for (; p + 8 <= S; p += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] += src[(size_t)(p + q) * KN];
    }
    for (int q = 0; p < S; ++p, ++q) s[q] += src[(size_t)p * KN];
    C[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

struct XtyPlan { int mt, nt, nw, kb_tiles, kblocks, splits; size_t lds; };

static XtyPlan xty_plan(int M_max, int Kout, int N, int nbatch) {
    XtyPlan p{};
    const int ktiles = (Kout + 15) / 16;
    p.nt = N <= 112 ? 7 : 13;
    p.mt = N <= 112 ? 2 : 1;
    const int max_w = p.mt >= 2 ? 8 : 13;                            // waves per workgroup (register budget: launch bounds)
    int tiles_per_wg = max_w * p.mt;
    p.kblocks = (ktiles + tiles_per_wg - 1) / tiles_per_wg;
    int px, py;
    for (;; ++p.kblocks) {                                          // more (narrower) K blocks until a wave's share of the slab
        p.kb_tiles = (ktiles + p.kblocks - 1) / p.kblocks;          // DMA fits its decoded instructions
        p.nw = (p.kb_tiles + p.mt - 1) / p.mt;
        px = xty_pitch(p.kb_tiles * 16); py = xty_pitch(p.nt * 16);
        const int ix = kXtyRows * px * 4 / 1024, iy = kXtyRows * py * 4 / 1024;
        if (((ix + p.nw - 1) / p.nw <= xty_max_instr(p.mt) && (iy + p.nw - 1) / p.nw <= xty_max_instr(p.mt)) || p.kb_tiles == 1) break;
    }
    // every DMA instruction moves a whole KiB: kXtyRows * pitch * 4 is a multiple of 1024 because pitch % 16 == 0 and 32 rows
    p.lds = (size_t)2 * kXtyRows * (px + py) * sizeof(float);
    // workgroups per CU: as many as registers (mt = 2: 3 waves per SIMD = 12 waves; mt = 1: 16 waves) and LDS allow
    int per_cu = (p.mt >= 2 ? 12 : 16) / p.nw;
    const int by_lds = (int)((size_t)160 * 1024 / p.lds);
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    static const int per_cu_env = [] { const char* e = getenv("GGNN_XTY_PER_CU"); return e ? atoi(e) : 0; }();   // (experiments)
    if (per_cu_env > 0) per_cu = per_cu_env;
    int target = per_cu * num_cus() / (p.kblocks * nbatch);
    if (target < 1) target = 1;
    const int max_s = (M_max + 4 * kXtyRows - 1) / (4 * kXtyRows);    // at least 4 slabs per split
    p.splits = target < max_s ? target : (max_s > 0 ? max_s : 1);
    return p;
}

template <int MT, int NT, bool GATHER>
static int launch_xty(const XtyArgs& a, const XtyPlan& p, float* C, hipStream_t st) {
    static std::atomic<unsigned long long> lds_ok{0};
    if (p.lds > 64 * 1024) GGNN_CHECK_HIP(allow_dynamic_lds(&xty_kernel<MT, NT, GATHER>, p.lds, lds_ok));
    hipLaunchKernelGGL((xty_kernel<MT, NT, GATHER>), dim3(p.kblocks, p.splits, a.nbatch), dim3(p.nw * 64), p.lds, st, a);
    GGNN_CHECK_HIP(hipGetLastError());
    const long long total = (long long)a.Kout * a.N * a.nbatch;
    hipLaunchKernelGGL(xty_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)a.part, C, a.Kout * a.N,
                       p.splits, a.nbatch);
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
    const XtyPlan p = xty_plan(M_max, K + 1, N, nbatch);             // (sized for the ones-row form)
    return (size_t)p.splits * nbatch * (K + 1) * N * sizeof(float) + 256;
}

extern "C" int ggnn_xty_f32(const float* const* x_segs, int nseg, int Dseg, const int32_t* ldx, const int32_t* x_rows,
                            const float* Y, int ldy, float* C, int K, int N, int ones_row, const int32_t* row_off, int nbatch,
                            void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    const int Kout = ones_row ? K + 1 : K;
    GGNN_CHECK_ARG(nseg >= 1 && nseg <= 4 && Dseg > 0 && Dseg % 4 == 0 && K == nseg * Dseg, "X is nseg <= 4 segments of Dseg columns (K = %d, nseg = %d, Dseg = %d)", K, nseg, Dseg);
    GGNN_CHECK_ARG(N > 0 && N % 4 == 0 && N <= 208, "N = %d must be a multiple of 4, <= 208", N);
    GGNN_CHECK_ARG(nbatch >= 1 && nbatch <= kXtyMaxBatch && row_off && C && ldx, "bad batch description");
    GGNN_CHECK_ARG(ldy >= N && ldy % 4 == 0, "ldy %d", ldy);
    hipStream_t st = (hipStream_t)stream;
    XtyArgs a{};
    int m_max = 0;
    for (int b = 0; b <= nbatch; ++b) {
        a.row_off[b] = row_off[b];
        if (b) { GGNN_CHECK_ARG(row_off[b] >= row_off[b - 1], "row_off not monotone"); m_max = row_off[b] - row_off[b - 1] > m_max ? row_off[b] - row_off[b - 1] : m_max; }
    }
    if (m_max == 0) {
        GGNN_CHECK_HIP(hipMemsetAsync(C, 0, (size_t)nbatch * Kout * N * sizeof(float), st));
        return GGNN_OK;
    }
    GGNN_CHECK_ARG(x_segs && Y && ws && aligned16(Y) && aligned16(ws), "null or misaligned pointer");
    for (int s = 0; s < nseg; ++s) {
        GGNN_CHECK_ARG(x_segs[s] && aligned16(x_segs[s]) && ldx[s] >= Dseg && ldx[s] % 4 == 0, "X segment %d null, misaligned or bad stride", s);
        a.X[s] = x_segs[s]; a.ldx[s] = ldx[s];
    }
    if (ws_bytes < ggnn_xty_workspace_bytes(m_max, K, N, nbatch)) return fail(GGNN_E_WORKSPACE, "xty workspace too small");
    const XtyPlan p = xty_plan(m_max, Kout, N, nbatch);
    a.Kout = Kout;
    a.nseg = nseg; a.Dseg = Dseg; a.x_rows = x_rows; a.Y = Y; a.ldy = ldy; a.part = static_cast<float*>(ws);
    a.K = K; a.N = N; a.nbatch = nbatch; a.splits = p.splits; a.kb_tiles = p.kb_tiles;
    a.pitch_x = xty_pitch(p.kb_tiles * 16); a.pitch_y = xty_pitch(p.nt * 16);
    const int px_kib = kXtyRows * a.pitch_x * 4 / 1024, py_kib = kXtyRows * a.pitch_y * 4 / 1024;
    if ((px_kib + p.nw - 1) / p.nw > xty_max_instr(p.mt) || (py_kib + p.nw - 1) / p.nw > xty_max_instr(p.mt))
        return fail(GGNN_E_UNSUPPORTED, "xty: slab too wide for %d waves (K=%d N=%d)", p.nw, K, N);
    if (p.nt == 7) return x_rows ? launch_xty<2, 7, true>(a, p, C, st) : launch_xty<2, 7, false>(a, p, C, st);
    return x_rows ? launch_xty<1, 13, true>(a, p, C, st) : launch_xty<1, 13, false>(a, p, C, st);
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
