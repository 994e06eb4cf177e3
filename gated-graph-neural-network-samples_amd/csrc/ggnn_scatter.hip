// K2: edge-indexed gather + segment sum (tf.unsorted_segment_sum of the per-edge messages,
// chem_tensorflow_sparse.py:160-162,168,198-209) and its index prep (:120-129).
//
// HBM-bound.  Design (DESIGN.md "K2"):
//   * the M messages are bucketed by TARGET once per batch with a stable radix sort (rocPRIM,
//     called directly): inside a node the slots keep the reference's accumulation order (type
//     ascending, then list order), the sum needs no atomics and is bit-reproducible.
//   * one sub-wave (16/32/64 lanes, 16 B per lane) owns one target node: it loads up to LPR slot
//     indices with ONE coalesced read, broadcasts them with __shfl, and streams the gathered
//     D-float source rows (each a contiguous, 16-byte aligned 4*D-byte read) with 4 rows in flight
//     per lane, accumulating in registers in slot order.
//   * bias (:202-204), mean normalisation (:206-209) and the single coalesced row store are the
//     epilogue; [M,D] messages are never materialised (the reference materialises them twice,
//     :161-168).
#include "ggnn_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace ggnn {

constexpr int kMaxTypes = 64;
struct TypeOffsets { long long off[kMaxTypes + 1]; int T; };

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// ---- index prep -----------------------------------------------------------------------------------
// by_source == 0: key = dst           (V segments),   gathered row = src*T + type   (forward)
// by_source == 1: key = src*T + type  (V*T segments), gathered row = dst            (backward: the transpose)
__device__ __forceinline__ int type_of(const TypeOffsets& to, long long m) {
    int t = 0;
    while (t + 1 < to.T && m >= to.off[t + 1]) ++t;
    return t;
}

__global__ void csr_prep_kernel(const int* __restrict__ adj, TypeOffsets to, long long M, int V, int by_source,
                                int* keys, int* vals, int* err_flag) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int src = adj[2 * m], dst = adj[2 * m + 1];
    if ((unsigned)src >= (unsigned)V || (unsigned)dst >= (unsigned)V) {
        if (err_flag) *err_flag = 1;
    }
    const int s = min(max(src, 0), V - 1), d = min(max(dst, 0), V - 1);
    keys[m] = by_source ? s * to.T + type_of(to, m) : d;
    vals[m] = (int)m;
}

__global__ void csr_finalize_kernel(const int* __restrict__ adj, const int* __restrict__ vals_sorted, TypeOffsets to,
                                    long long M, int V, int by_source, int* gather_row, int* msg_perm) {
    const long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= M) return;
    const int m = vals_sorted[slot];
    const int src = min(max(adj[2 * (long long)m], 0), V - 1);
    const int dst = min(max(adj[2 * (long long)m + 1], 0), V - 1);
    gather_row[slot] = by_source ? dst : src * to.T + type_of(to, m);
    if (msg_perm) msg_perm[slot] = m;
}

__global__ void csr_rowptr_kernel(const int* __restrict__ keys_sorted, long long M, int V, int* row_ptr) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > V) return;
    long long lo = 0, hi = M;          // first slot with key >= v
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    row_ptr[v] = (int)lo;
}

static size_t cub_temp_bytes(long long M, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                    (unsigned int)M, 0u, (unsigned int)end_bit, (hipStream_t)0);
    return bytes;
}

static int key_bits(int V) {
    int bits = 1;
    while (bits < 31 && (1LL << bits) < (long long)V) ++bits;
    return bits;
}

// ---- gather + segment sum ---------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void gather_segment_sum_kernel(
        const float* __restrict__ H, const int* __restrict__ row_ptr, const int* __restrict__ gidx,
        const float* __restrict__ nin, const float* __restrict__ bias, int use_avg, float* __restrict__ out,
        int V, int D, int T, int accumulate) {
    constexpr int NODES = 256 / LPR;
    const int l = threadIdx.x % LPR;
    int v = blockIdx.x * NODES + threadIdx.x / LPR;
    const bool live = v < V;
    v = live ? v : V - 1;                       // keep the sub-wave convergent for __shfl
    const int beg = row_ptr[v], end = live ? row_ptr[v + 1] : beg;
    const int D4 = D >> 2;

    float deg = 0.f;
    if (use_avg && nin)
        for (int t = 0; t < T; ++t) deg += nin[(size_t)v * T + t];
    const float den = deg + 1e-7f;              // utils.py:8 SMALL_NUMBER, fp32 like the reference

    for (int c0 = 0; c0 < D4; c0 += LPR) {      // one pass for D <= 4*LPR
        const int c4 = c0 + l;
        const bool col_ok = c4 < D4;
        const float* hcol = H + 4 * (col_ok ? c4 : 0);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e0 = beg; e0 < end; e0 += LPR) {
            const int cnt = min(LPR, end - e0);
            const int my = (l < cnt) ? gidx[e0 + l] : 0;     // one coalesced read of <= LPR slot indices
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {                   // 4 gathered rows in flight per lane
                const int i0 = __shfl(my, j, LPR), i1 = __shfl(my, j + 1, LPR);
                const int i2 = __shfl(my, j + 2, LPR), i3 = __shfl(my, j + 3, LPR);
                const f32x4 r0 = *reinterpret_cast<const f32x4*>(hcol + (size_t)i0 * D);
                const f32x4 r1 = *reinterpret_cast<const f32x4*>(hcol + (size_t)i1 * D);
                const f32x4 r2 = *reinterpret_cast<const f32x4*>(hcol + (size_t)i2 * D);
                const f32x4 r3 = *reinterpret_cast<const f32x4*>(hcol + (size_t)i3 * D);
                acc += r0; acc += r1; acc += r2; acc += r3;  // slot order = reference accumulation order
            }
            for (; j < cnt; ++j) {
                const int i0 = __shfl(my, j, LPR);
                acc += *reinterpret_cast<const f32x4*>(hcol + (size_t)i0 * D);
            }
        }
        if (col_ok && live) {
            if (bias) {                                       // :202-204  incoming += nin @ edge_biases
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                for (int t = 0; t < T; ++t)
                    b += nin[(size_t)v * T + t] * *reinterpret_cast<const f32x4*>(bias + (size_t)t * D + 4 * c4);
                acc += b;
            }
            if (use_avg) acc = acc / den;                     // :206-209
            if (accumulate) acc += *reinterpret_cast<const f32x4*>(out + (size_t)v * D + 4 * c4);
            *reinterpret_cast<f32x4*>(out + (size_t)v * D + 4 * c4) = acc;
        }
    }
}

// Flat variant: one lane per (node, float4 column) with no idle lanes (D/4 = 25 lanes per node do not divide a wave:
// the sub-wave form keeps 50 of 64 lanes busy).  Lanes of one node read the same row_ptr / slot indices (served as
// broadcasts by the vector cache); no cross-lane traffic.  Slots are walked 2 at a time.
__global__ __launch_bounds__(256) void gather_segment_sum_flat_kernel(
        const float* __restrict__ H, const int* __restrict__ row_ptr, const int* __restrict__ gidx,
        const float* __restrict__ nin, const float* __restrict__ bias, int use_avg, float* __restrict__ out,
        long long total4, int D, int T, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int D4 = D >> 2;
    const int v = (int)(i / D4);
    const int c4 = (int)(i - (long long)v * D4);
    const int beg = row_ptr[v], end = row_ptr[v + 1];
    const float* hcol = H + 4 * c4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int e = beg;
    for (; e + 2 <= end; e += 2) {
        const int i0 = gidx[e], i1 = gidx[e + 1];
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(hcol + (size_t)i0 * D);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(hcol + (size_t)i1 * D);
        acc += r0; acc += r1;
    }
    if (e < end) acc += *reinterpret_cast<const f32x4*>(hcol + (size_t)gidx[e] * D);
    if (bias || use_avg) {
        float deg = 0.f;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < T; ++t) {
            const float n = nin[(size_t)v * T + t];
            deg += n;
            if (bias) b += n * *reinterpret_cast<const f32x4*>(bias + (size_t)t * D + 4 * c4);
        }
        if (bias) acc += b;
        if (use_avg) acc = acc / (deg + 1e-7f);
    }
    if (accumulate) acc += *reinterpret_cast<const f32x4*>(out + (size_t)v * D + 4 * c4);
    *reinterpret_cast<f32x4*>(out + (size_t)v * D + 4 * c4) = acc;
}

// Slot heads: the gather rows of the first FOUR message slots of every node in one 16-byte record (-1 = no such slot).
// The CSR walk is a chain of three dependent loads per lane (row_ptr -> slot index -> source row); with the head record the
// slot indices of a node arrive with ONE load that does not depend on row_ptr, so a lane has its (up to) four source rows in
// flight after a single round trip -- at QM9 shapes (no atom has more than 4 bonds) the slot list is never touched at all.
__global__ void slot_heads_kernel(const int* __restrict__ row_ptr, const int* __restrict__ gidx, int V, int4* __restrict__ heads) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int beg = row_ptr[v], n = row_ptr[v + 1] - beg;
    int4 h;
    h.x = n > 0 ? gidx[beg] : -1; h.y = n > 1 ? gidx[beg + 1] : -1;
    h.z = n > 2 ? gidx[beg + 2] : -1; h.w = n > 3 ? gidx[beg + 3] : -1;
    heads[v] = h;
}

// One lane per (node, float4 column), slot heads: same sums, same order, same epilogue as gather_segment_sum_flat_kernel.
// A thread works on ITEMS (node, column) items, 256 apart, in three phases -- all head records, then all (up to 4 ITEMS) source
// rows, then the adds and stores -- so that it has ITEMS times the loads in flight per dependent level and the launch has
// 1 / ITEMS of the workgroups: with one item per thread the 25 us launch at QM9 shape spent a third of its time ramping up and
// draining 9,766 tiny workgroups.
template <int ITEMS>
__global__ __launch_bounds__(256) void gather_segment_sum_heads_kernel(
        const float* __restrict__ H, const int* __restrict__ row_ptr, const int* __restrict__ gidx, const int4* __restrict__ heads,
        const float* __restrict__ nin, const float* __restrict__ bias, int use_avg, float* __restrict__ out,
        long long total4, int D, int T, int accumulate) {
    const int D4 = D >> 2;
    const long long i0 = ((long long)blockIdx.x * ITEMS) * 256 + threadIdx.x;
    int v[ITEMS], c4[ITEMS];
    bool on[ITEMS];
    int4 hd[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long i = i0 + (long long)k * 256;
        on[k] = i < total4;
        const long long ic = on[k] ? i : 0;
        v[k] = (int)(ic / D4);
        c4[k] = (int)(ic - (long long)v[k] * D4);
        hd[k] = heads[v[k]];
        if (!on[k]) hd[k] = make_int4(-1, -1, -1, -1);
    }
    f32x4 r[ITEMS][4];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const float* hcol = H + 4 * c4[k];
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        r[k][0] = hd[k].x >= 0 ? *reinterpret_cast<const f32x4*>(hcol + (size_t)hd[k].x * D) : z;
        r[k][1] = hd[k].y >= 0 ? *reinterpret_cast<const f32x4*>(hcol + (size_t)hd[k].y * D) : z;
        r[k][2] = hd[k].z >= 0 ? *reinterpret_cast<const f32x4*>(hcol + (size_t)hd[k].z * D) : z;
        r[k][3] = hd[k].w >= 0 ? *reinterpret_cast<const f32x4*>(hcol + (size_t)hd[k].w * D) : z;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (!on[k]) continue;
        const float* hcol = H + 4 * c4[k];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc += r[k][0];                                             // slot order = reference accumulation order
        if (hd[k].y >= 0) acc += r[k][1];
        if (hd[k].z >= 0) acc += r[k][2];
        if (hd[k].w >= 0) {
            acc += r[k][3];
            const int beg = row_ptr[v[k]], end = row_ptr[v[k] + 1];     // (only a node with more than four slots walks the slot list)
            for (int e = beg + 4; e < end; ++e) acc += *reinterpret_cast<const f32x4*>(hcol + (size_t)gidx[e] * D);
        }
        if (bias || use_avg) {
            float deg = 0.f;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < T; ++t) {
                const float n = nin[(size_t)v[k] * T + t];
                deg += n;
                if (bias) b += n * *reinterpret_cast<const f32x4*>(bias + (size_t)t * D + 4 * c4[k]);
            }
            if (bias) acc += b;
            if (use_avg) acc = acc / (deg + 1e-7f);
        }
        float* o = out + (size_t)v[k] * D + 4 * c4[k];
        if (accumulate) acc += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = acc;
    }
}

// ---- propagation attention (chem_tensorflow_sparse.py:147-149, 170-196) fused into the segment sum ---------
// score_m = <h[src_m], h[tgt_m]> * factor[type_m]; a_m = softmax over the messages INTO each target (max-shifted,
// denominator + 1e-7); incoming[v] = sum_m a_m * msg_m.  One sub-wave per target: pass 1 finds the max score,
// pass 2 recomputes the scores, and accumulates e_m * msg_m and sum e_m; the division by (sum + 1e-7) is applied
// once to the accumulated row (== scaling every message first, up to fp32 rounding order).
template <int LPR>
__global__ __launch_bounds__(256) void gather_segment_sum_attn_kernel(
        const float* __restrict__ H, const float* __restrict__ h, const int* __restrict__ row_ptr,
        const int* __restrict__ gidx, const float* __restrict__ factors, const float* __restrict__ nin,
        const float* __restrict__ bias, int use_avg, float* __restrict__ out, int V, int D, int T) {
    constexpr int NODES = 256 / LPR;
    const int l = threadIdx.x % LPR;
    int v = blockIdx.x * NODES + threadIdx.x / LPR;
    const bool live = v < V;
    v = live ? v : V - 1;
    const int beg = row_ptr[v], end = live ? row_ptr[v + 1] : beg;
    const int D4 = D >> 2;                                  // D4 <= LPR (checked by the launcher)
    const bool col_ok = l < D4;
    const int c4 = col_ok ? l : 0;
    const f32x4 hv = col_ok ? *reinterpret_cast<const f32x4*>(h + (size_t)v * D + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};

    auto score = [&](int g) {
        const int src = g / T, t = g - src * T;
        f32x4 hs = {0.f, 0.f, 0.f, 0.f};
        if (col_ok) hs = *reinterpret_cast<const f32x4*>(h + (size_t)src * D + 4 * c4);
        float part = hs.x * hv.x + hs.y * hv.y + hs.z * hv.z + hs.w * hv.w;
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, LPR);
        return part * factors[t];
    };
    float m = -3.402823466e+38f;
    for (int e = beg; e < end; ++e) m = fmaxf(m, score(gidx[e]));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float S = 0.f;
    for (int e = beg; e < end; ++e) {
        const int g = gidx[e];
        const float w = expf(score(g) - m);
        S += w;
        if (col_ok) acc += w * *reinterpret_cast<const f32x4*>(H + (size_t)g * D + 4 * c4);
    }
    if (col_ok && live) {
        acc = acc / (S + 1e-7f);                            // :194
        float deg = 0.f;
        if ((use_avg || bias) && nin)
            for (int t = 0; t < T; ++t) deg += nin[(size_t)v * T + t];
        if (bias) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < T; ++t)
                b += nin[(size_t)v * T + t] * *reinterpret_cast<const f32x4*>(bias + (size_t)t * D + 4 * c4);
            acc += b;
        }
        if (use_avg) acc = acc / (deg + 1e-7f);
        *reinterpret_cast<f32x4*>(out + (size_t)v * D + 4 * c4) = acc;
    }
}

// ---- backward of the propagation attention (what TF autodiff derives from chem_tensorflow_sparse.py:170-196) ----------------
// Forward, per target v over its message slots e (g_e = src_e*T + t_e):  p_e = <h[src_e], h[v]>,  s_e = p_e f[t_e],
//   a_e = exp(s_e - max) / (sum_k exp(s_k - max) + 1e-7),   att[v] = sum_e a_e H[g_e].
// Given d = dL/d att [V,D]:   da_e = <H[g_e], d[v]>,   ds_e = a_e (da_e - sum_k a_k da_k)
//   target side (this kernel):   dh[v] (+)= sum_e ds_e f[t_e] h[src_e]
//   per message (written by ORIGINAL message id m = msg_perm[slot], the order the by-source index refers to):
//       coef_a[m] = a_e            -> dH[g_e]   += a_e d[v]                 (weighted transpose gather, below)
//       coef_s[m] = ds_e f[t_e]    -> dh[src_e] += ds_e f[t_e] h[v]         (weighted gather by source node, below)
//       dfac[m]   = ds_e p_e       -> df[t]      = sum over the messages of type t  (range reduction, below)
// One sub-wave per target, the scores recomputed in every pass (no [M] temporaries beyond the three outputs).
template <int LPR>
__global__ __launch_bounds__(256) void attn_bwd_target_kernel(
        const float* __restrict__ H, const float* __restrict__ h, const float* __restrict__ d, const int* __restrict__ row_ptr,
        const int* __restrict__ gidx, const int* __restrict__ msg_perm, const float* __restrict__ factors,
        float* __restrict__ coef_a, float* __restrict__ coef_s, float* __restrict__ dfac, float* __restrict__ dh, int accumulate,
        int V, int D, int T) {
    constexpr int NODES = 256 / LPR;
    const int l = threadIdx.x % LPR;
    int v = blockIdx.x * NODES + threadIdx.x / LPR;
    const bool live = v < V;
    v = live ? v : V - 1;
    const int beg = row_ptr[v], end = live ? row_ptr[v + 1] : beg;
    const int D4 = D >> 2;
    const bool col_ok = l < D4;
    const int c4 = col_ok ? l : 0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 hv = col_ok ? *reinterpret_cast<const f32x4*>(h + (size_t)v * D + 4 * c4) : zero;
    const f32x4 dv = col_ok ? *reinterpret_cast<const f32x4*>(d + (size_t)v * D + 4 * c4) : zero;
    auto dot = [&](f32x4 a, f32x4 b) {
        float part = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
#pragma unroll
        for (int off = LPR / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, LPR);
        return part;
    };
    auto hsrc = [&](int g) { return col_ok ? *reinterpret_cast<const f32x4*>(h + (size_t)(g / T) * D + 4 * c4) : zero; };
    float m = -3.402823466e+38f;
    for (int e = beg; e < end; ++e) { const int g = gidx[e]; m = fmaxf(m, dot(hsrc(g), hv) * factors[g % T]); }
    float S = 0.f;
    for (int e = beg; e < end; ++e) { const int g = gidx[e]; S += expf(dot(hsrc(g), hv) * factors[g % T] - m); }
    const float inv = 1.0f / (S + 1e-7f);
    float t1 = 0.f;
    for (int e = beg; e < end; ++e) {
        const int g = gidx[e];
        const float a = expf(dot(hsrc(g), hv) * factors[g % T] - m) * inv;
        const f32x4 Hg = col_ok ? *reinterpret_cast<const f32x4*>(H + (size_t)g * D + 4 * c4) : zero;
        t1 += a * dot(Hg, dv);
    }
    f32x4 acc = zero;
    for (int e = beg; e < end; ++e) {
        const int g = gidx[e];
        const f32x4 hs = hsrc(g);
        const float f = factors[g % T];
        const float p = dot(hs, hv);
        const float a = expf(p * f - m) * inv;
        const f32x4 Hg = col_ok ? *reinterpret_cast<const f32x4*>(H + (size_t)g * D + 4 * c4) : zero;
        const float ds = a * (dot(Hg, dv) - t1);
        acc += (ds * f) * hs;
        if (l == 0) {
            const int mid = msg_perm[e];
            coef_a[mid] = a; coef_s[mid] = ds * f; dfac[mid] = ds * p;
        }
    }
    if (col_ok && live) {
        float* o = dh + (size_t)v * D + 4 * c4;
        if (accumulate) acc += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = acc;
    }
}

// out[seg] (+)= sum over the slots of seg of w[wid[slot]] * rows[gidx[slot]]   (slot order; one sub-wave per segment)
template <int LPR>
__global__ __launch_bounds__(256) void weighted_segment_sum_kernel(
        const float* __restrict__ rows, const int* __restrict__ row_ptr, const int* __restrict__ gidx, const int* __restrict__ wid,
        const float* __restrict__ w, float* __restrict__ out, int accumulate, int nseg, int D) {
    constexpr int SEGS = 256 / LPR;
    const int l = threadIdx.x % LPR;
    const int sgm = blockIdx.x * SEGS + threadIdx.x / LPR;
    if (sgm >= nseg) return;
    const int beg = row_ptr[sgm], end = row_ptr[sgm + 1];
    const int D4 = D >> 2;
    for (int c4 = l; c4 < D4; c4 += LPR) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int e = beg; e < end; ++e)
            acc += w[wid[e]] * *reinterpret_cast<const f32x4*>(rows + (size_t)gidx[e] * D + 4 * c4);
        float* o = out + (size_t)sgm * D + 4 * c4;
        if (accumulate) acc += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = acc;
    }
}

// out[b] = sum of values[off[b] .. off[b+1])   (one block per range, fixed-order tree: deterministic)
struct RangeOffsets { long long off[kMaxTypes + 1]; };
__global__ __launch_bounds__(256) void range_sum_kernel(const float* __restrict__ values, RangeOffsets ro, float* __restrict__ out) {
    __shared__ float red[256];
    const long long beg = ro.off[blockIdx.x], end = ro.off[blockIdx.x + 1];
    float s = 0.f;
    for (long long i = beg + threadIdx.x; i < end; i += 256) s += values[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

__global__ void unsorted_segment_sum_kernel(const float* __restrict__ data, const int* __restrict__ ids,
                                            float* out, long long total, int D, int num_segments) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long long m = e / D;
    const int d = (int)(e - m * D);
    const int id = ids[m];
    if ((unsigned)id < (unsigned)num_segments) unsafeAtomicAdd(out + (size_t)id * D + d, data[e]);
}

}  // namespace ggnn

using namespace ggnn;

extern "C" size_t ggnn_csr_workspace_bytes(int64_t M, int V) {
    if (M <= 0 || V <= 0) return 256;
    // sized for the larger (by-source, V*T <= 2^31 keys) variant: 31 key bits
    return 4 * align256((size_t)M * sizeof(int)) + align256(cub_temp_bytes(M, 31)) + 256;
}

static int build_csr(const int32_t* adj, const int64_t* type_off, int T, int V, int64_t M, int by_source,
                     int32_t* row_ptr, int32_t* gather_row, int32_t* msg_perm, int32_t* err_flag, void* ws,
                     size_t ws_bytes, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(T > 0 && T <= kMaxTypes, "num_edge_types %d outside 1..%d", T, kMaxTypes);
    GGNN_CHECK_ARG(V >= 0 && M >= 0 && M < (1LL << 31), "bad sizes V=%d M=%lld", V, (long long)M);
    GGNN_CHECK_ARG((long long)V * T < (1LL << 31), "V*T overflows int32");
    GGNN_CHECK_ARG(type_off && row_ptr, "null pointer");
    GGNN_CHECK_ARG(type_off[0] == 0 && type_off[T] == M, "type_off must start at 0 and end at M");
    for (int t = 0; t < T; ++t) GGNN_CHECK_ARG(type_off[t] <= type_off[t + 1], "type_off not monotone");
    hipStream_t st = (hipStream_t)stream;
    const int nseg = by_source ? V * T : V;
    if (M == 0 || V == 0) {
        GGNN_CHECK_HIP(hipMemsetAsync(row_ptr, 0, sizeof(int) * ((size_t)nseg + 1), st));
        return GGNN_OK;
    }
    GGNN_CHECK_ARG(adj && gather_row && ws, "null pointer");
    if (ws_bytes < ggnn_csr_workspace_bytes(M, V))
        return fail(GGNN_E_WORKSPACE, "CSR workspace too small: %zu < %zu", ws_bytes, ggnn_csr_workspace_bytes(M, V));
    const size_t arr = align256((size_t)M * sizeof(int));
    char* p = static_cast<char*>(ws);
    p = reinterpret_cast<char*>(align256(reinterpret_cast<size_t>(p)));
    int* keys_in = reinterpret_cast<int*>(p);
    int* vals_in = reinterpret_cast<int*>(p + arr);
    int* keys_out = reinterpret_cast<int*>(p + 2 * arr);
    int* vals_out = reinterpret_cast<int*>(p + 3 * arr);
    void* cub_ws = p + 4 * arr;
    const int bits = key_bits(nseg);
    size_t cub_bytes = cub_temp_bytes(M, bits);
    TypeOffsets to;
    to.T = T;
    for (int t = 0; t <= T; ++t) to.off[t] = type_off[t];

    const int threads = 256;
    const unsigned blocks_m = (unsigned)((M + threads - 1) / threads);
    hipLaunchKernelGGL(csr_prep_kernel, dim3(blocks_m), dim3(threads), 0, st, adj, to, (long long)M, V, by_source,
                       keys_in, vals_in, err_flag);
    GGNN_CHECK_HIP(hipGetLastError());
    GGNN_CHECK_HIP(rocprim::radix_sort_pairs(cub_ws, cub_bytes, (const int*)keys_in, keys_out, (const int*)vals_in, vals_out,
                                             (unsigned int)M, 0u, (unsigned int)bits, st));
    hipLaunchKernelGGL(csr_finalize_kernel, dim3(blocks_m), dim3(threads), 0, st, adj, (const int*)vals_out, to,
                       (long long)M, V, by_source, gather_row, msg_perm);
    GGNN_CHECK_HIP(hipGetLastError());
    const unsigned blocks_v = (unsigned)(((long long)nseg + 1 + threads - 1) / threads);
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3(blocks_v), dim3(threads), 0, st, (const int*)keys_out, (long long)M,
                       nseg, row_ptr);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_build_target_csr(const int32_t* adj, const int64_t* type_off, int T, int V, int64_t M,
                                     int32_t* row_ptr, int32_t* gather_row, int32_t* msg_perm, int32_t* err_flag,
                                     void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    return build_csr(adj, type_off, T, V, M, 0, row_ptr, gather_row, msg_perm, err_flag, ws, ws_bytes, stream);
}

extern "C" int ggnn_build_source_csr(const int32_t* adj, const int64_t* type_off, int T, int V, int64_t M,
                                     int32_t* row_ptr, int32_t* gather_row, int32_t* msg_perm, int32_t* err_flag,
                                     void* ws, size_t ws_bytes, ggnn_stream_t stream) {
    return build_csr(adj, type_off, T, V, M, 1, row_ptr, gather_row, msg_perm, err_flag, ws, ws_bytes, stream);
}

static int gather_segment_sum_impl(const float* Hrows, const int32_t* row_ptr, const int32_t* gather_row,
                                   const float* nin, const float* bias, int use_avg, float* out, int V,
                                   int D, int T, int accumulate, ggnn_stream_t stream, const int32_t* heads = nullptr) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && T > 0, "bad sizes V=%d D=%d T=%d", V, D, T);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(Hrows && row_ptr && out, "null pointer");
    GGNN_CHECK_ARG(!(bias || use_avg) || nin, "nin is required with bias or mean aggregation");
    GGNN_CHECK_ARG(aligned16(Hrows) && aligned16(out) && (!bias || aligned16(bias)), "pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int D4 = D / 4;
    // lane-per-element when D/4 lanes per node do not tile a wave (D=100: 25 lanes -> 27.9 vs 29.8 us at config 2);
    // sub-wave-per-node otherwise (D=256, deg 10: 158 vs 166 us at config 5).  GGNN_K2_FLAT=0/1 overrides.
    static int flat_env = -2;
    if (flat_env == -2) { const char* e = getenv("GGNN_K2_FLAT"); flat_env = e ? atoi(e) : -1; }
    const bool flat = flat_env >= 0 ? flat_env != 0 : (D4 < 64 && (64 % D4) != 0);
    if (heads && flat) {   // (the sub-wave kernels keep their own slot-index broadcast: at D = 256 / in-degree 10 they measure 155 vs 179 us)
        const long long total4 = (long long)V * D4;
        static const int items = [] { const char* e = getenv("GGNN_K2_ITEMS"); return e ? atoi(e) : 2; }();
        if (items == 1)
            hipLaunchKernelGGL(gather_segment_sum_heads_kernel<1>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, Hrows, row_ptr,
                               gather_row, reinterpret_cast<const int4*>(heads), nin, bias, use_avg, out, total4, D, T, accumulate);
        else if (items == 2)
            hipLaunchKernelGGL(gather_segment_sum_heads_kernel<2>, dim3((unsigned)((total4 + 511) / 512)), dim3(256), 0, st, Hrows, row_ptr,
                               gather_row, reinterpret_cast<const int4*>(heads), nin, bias, use_avg, out, total4, D, T, accumulate);
        else if (items == 8)
            hipLaunchKernelGGL(gather_segment_sum_heads_kernel<8>, dim3((unsigned)((total4 + 2047) / 2048)), dim3(256), 0, st, Hrows, row_ptr,
                               gather_row, reinterpret_cast<const int4*>(heads), nin, bias, use_avg, out, total4, D, T, accumulate);
        else
            hipLaunchKernelGGL(gather_segment_sum_heads_kernel<4>, dim3((unsigned)((total4 + 1023) / 1024)), dim3(256), 0, st, Hrows, row_ptr,
                               gather_row, reinterpret_cast<const int4*>(heads), nin, bias, use_avg, out, total4, D, T, accumulate);
    } else if (flat) {
        const long long total4 = (long long)V * D4;
        hipLaunchKernelGGL(gather_segment_sum_flat_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, Hrows,
                           row_ptr, gather_row, nin, bias, use_avg, out, total4, D, T, accumulate);
    } else if (D4 <= 16) {
        hipLaunchKernelGGL(gather_segment_sum_kernel<16>, dim3((V + 15) / 16), dim3(256), 0, st, Hrows, row_ptr,
                           gather_row, nin, bias, use_avg, out, V, D, T, accumulate);
    } else if (D4 <= 32) {
        hipLaunchKernelGGL(gather_segment_sum_kernel<32>, dim3((V + 7) / 8), dim3(256), 0, st, Hrows, row_ptr,
                           gather_row, nin, bias, use_avg, out, V, D, T, accumulate);
    } else {
        hipLaunchKernelGGL(gather_segment_sum_kernel<64>, dim3((V + 3) / 4), dim3(256), 0, st, Hrows, row_ptr,
                           gather_row, nin, bias, use_avg, out, V, D, T, accumulate);
    }
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_gather_segment_sum_f32(const float* Hrows, const int32_t* row_ptr, const int32_t* gather_row,
                                           const float* nin, const float* bias, int use_avg, float* out, int V,
                                           int D, int T, ggnn_stream_t stream) {
    return gather_segment_sum_impl(Hrows, row_ptr, gather_row, nin, bias, use_avg, out, V, D, T, 0, stream);
}

// out[v] += sum of the gathered rows (backward pass: several gradient contributions meet in one [V,D] tensor)
extern "C" int ggnn_gather_segment_sum_acc_f32(const float* Hrows, const int32_t* row_ptr, const int32_t* gather_row, float* out,
                                               int V, int D, ggnn_stream_t stream) {
    return gather_segment_sum_impl(Hrows, row_ptr, gather_row, nullptr, nullptr, 0, out, V, D, 1, 1, stream);
}

extern "C" int ggnn_gather_segment_sum_attn_f32(const float* Hrows, const float* h, const int32_t* row_ptr,
                                                const int32_t* gather_row, const float* type_factors, const float* nin,
                                                const float* bias, int use_avg, float* out, int V, int D, int T,
                                                ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && T > 0, "bad sizes V=%d D=%d T=%d", V, D, T);
    if (D > 256) return fail(GGNN_E_UNSUPPORTED, "propagation attention supports hidden sizes up to 256 (got %d)", D);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(Hrows && h && row_ptr && type_factors && out, "null pointer");
    GGNN_CHECK_ARG(!(bias || use_avg) || nin, "nin is required with bias or mean aggregation");
    GGNN_CHECK_ARG(aligned16(Hrows) && aligned16(h) && aligned16(out) && (!bias || aligned16(bias)), "pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int D4 = D / 4;
    if (D4 <= 16)
        hipLaunchKernelGGL(gather_segment_sum_attn_kernel<16>, dim3((V + 15) / 16), dim3(256), 0, st, Hrows, h, row_ptr, gather_row,
                           type_factors, nin, bias, use_avg, out, V, D, T);
    else if (D4 <= 32)
        hipLaunchKernelGGL(gather_segment_sum_attn_kernel<32>, dim3((V + 7) / 8), dim3(256), 0, st, Hrows, h, row_ptr, gather_row,
                           type_factors, nin, bias, use_avg, out, V, D, T);
    else
        hipLaunchKernelGGL(gather_segment_sum_attn_kernel<64>, dim3((V + 3) / 4), dim3(256), 0, st, Hrows, h, row_ptr, gather_row,
                           type_factors, nin, bias, use_avg, out, V, D, T);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_unsorted_segment_sum_f32(const float* data, const int32_t* ids, float* out, int64_t M, int D,
                                             int num_segments, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(M >= 0 && D > 0 && num_segments >= 0, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (num_segments == 0) return GGNN_OK;
    GGNN_CHECK_ARG(out, "null pointer");
    GGNN_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)num_segments * D, st));
    if (M == 0) return GGNN_OK;
    GGNN_CHECK_ARG(data && ids, "null pointer");
    const long long total = (long long)M * D;
    hipLaunchKernelGGL(unsorted_segment_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, data,
                       ids, out, total, D, num_segments);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// Slot heads of a by-target (or by-source) index: heads[v] = the first four entries of gather_row[row_ptr[v] .. row_ptr[v+1]) (-1 padded).
extern "C" int ggnn_build_slot_heads(const int32_t* row_ptr, const int32_t* gather_row, int32_t* heads, int V, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0, "negative V");
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(row_ptr && heads && aligned16(heads), "null or misaligned pointer");
    hipLaunchKernelGGL(slot_heads_kernel, dim3((V + 255) / 256), dim3(256), 0, (hipStream_t)stream, row_ptr, gather_row, V,
                       reinterpret_cast<int4*>(heads));
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

// ggnn_gather_segment_sum_f32 with the slot heads of the index (ggnn_build_slot_heads): identical results, one dependent
// load level less per lane.
extern "C" int ggnn_gather_segment_sum_heads_f32(const float* Hrows, const int32_t* row_ptr, const int32_t* gather_row,
                                                 const int32_t* heads, const float* nin, const float* bias, int use_avg, float* out,
                                                 int V, int D, int T, int accumulate, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(heads && aligned16(heads), "heads null or misaligned");
    return gather_segment_sum_impl(Hrows, row_ptr, gather_row, nin, bias, use_avg, out, V, D, T, accumulate, stream, heads);
}

// ---- propagation-attention backward (see attn_bwd_target_kernel) ----------------------------------------------------------
extern "C" int ggnn_attn_bwd_target_f32(const float* Hrows, const float* h, const float* d_att, const int32_t* row_ptr,
                                        const int32_t* gather_row, const int32_t* msg_perm, const float* type_factors,
                                        float* coef_a, float* coef_s, float* dfac, float* dh, int accumulate, int V, int D, int T,
                                        ggnn_stream_t stream) {
    GGNN_CHECK_ARG(V >= 0 && D > 0 && D % 4 == 0 && T > 0, "bad sizes V=%d D=%d T=%d", V, D, T);
    if (D > 256) return fail(GGNN_E_UNSUPPORTED, "propagation attention supports hidden sizes up to 256 (got %d)", D);
    if (V == 0) return GGNN_OK;
    GGNN_CHECK_ARG(Hrows && h && d_att && row_ptr && type_factors && dh, "null pointer");
    GGNN_CHECK_ARG(aligned16(Hrows) && aligned16(h) && aligned16(d_att) && aligned16(dh), "pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int D4 = D / 4;
#define GGNN_ATTN_BWD(LPR, NODES)                                                                                     \
    hipLaunchKernelGGL(attn_bwd_target_kernel<LPR>, dim3((V + NODES - 1) / NODES), dim3(256), 0, st, Hrows, h, d_att, row_ptr,   \
                       gather_row, msg_perm, type_factors, coef_a, coef_s, dfac, dh, accumulate, V, D, T)
    if (D4 <= 16) GGNN_ATTN_BWD(16, 16);
    else if (D4 <= 32) GGNN_ATTN_BWD(32, 8);
    else GGNN_ATTN_BWD(64, 4);
#undef GGNN_ATTN_BWD
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_weighted_segment_sum_f32(const float* rows, const int32_t* row_ptr, const int32_t* gather_row,
                                             const int32_t* weight_id, const float* weights, float* out, int accumulate,
                                             int num_segments, int D, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(num_segments >= 0 && D > 0 && D % 4 == 0, "bad sizes");
    if (num_segments == 0) return GGNN_OK;
    GGNN_CHECK_ARG(rows && row_ptr && out && aligned16(rows) && aligned16(out), "null or misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const int D4 = D / 4;
    if (D4 <= 16)
        hipLaunchKernelGGL(weighted_segment_sum_kernel<16>, dim3((num_segments + 15) / 16), dim3(256), 0, st, rows, row_ptr, gather_row,
                           weight_id, weights, out, accumulate, num_segments, D);
    else if (D4 <= 32)
        hipLaunchKernelGGL(weighted_segment_sum_kernel<32>, dim3((num_segments + 7) / 8), dim3(256), 0, st, rows, row_ptr, gather_row,
                           weight_id, weights, out, accumulate, num_segments, D);
    else
        hipLaunchKernelGGL(weighted_segment_sum_kernel<64>, dim3((num_segments + 3) / 4), dim3(256), 0, st, rows, row_ptr, gather_row,
                           weight_id, weights, out, accumulate, num_segments, D);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}

extern "C" int ggnn_range_sum_f32(const float* values, const int64_t* range_off, int num_ranges, float* out, ggnn_stream_t stream) {
    GGNN_CHECK_ARG(num_ranges >= 0 && num_ranges <= kMaxTypes && range_off && out, "bad ranges");
    if (num_ranges == 0) return GGNN_OK;
    RangeOffsets ro;
    for (int b = 0; b <= num_ranges; ++b) { ro.off[b] = range_off[b]; GGNN_CHECK_ARG(b == 0 || range_off[b] >= range_off[b - 1], "ranges not monotone"); }
    GGNN_CHECK_ARG(values || ro.off[num_ranges] == ro.off[0], "null pointer");
    hipLaunchKernelGGL(range_sum_kernel, dim3(num_ranges), dim3(256), 0, (hipStream_t)stream, values, ro, out);
    GGNN_CHECK_HIP(hipGetLastError());
    return GGNN_OK;
}
