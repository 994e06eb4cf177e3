// Shared declarations of the graph-resident dense forward (ggnn_dense_graph.hip: f32 MFMA; ggnn_dense_graph_split.hip: split form).
#pragma once
#include "ggnn_common.h"

namespace ggnn {

struct DenseGraphArgs {
    const float* h0;        // [b, v, D]
    const float* A;         // [b, E, v, v]  A[g,e,dst,src]
    const float* eimg;      // E stage images of W_e            (ggnn_dense_edge_pack_f32; the split kernel: E split images)
    const float* gimg;      // 6 stage images: Wg[x,r] Wg[h,r] Wg[x,u] Wg[h,u] Wc[x] Wc[h]   (dense_gru_pack_kernel)
    const float* ebias;     // [E, D] or NULL
    const float* bg;        // [2D]
    const float* bc;        // [D]
    float* out;             // [b, v, D]
    int b, v, steps;
    unsigned long long* tdbg;   // (debug) s_memtime stamps of workgroup 0, waves 0 and 6: [step][wave sel][8]   (GGNN_DG_TPTR)
};

// split form (ggnn_dense_graph_split.hip): 1 when the kernel exists for the shape and its LDS blocks fit
int dense_split_supported(int v, int E, int D);
size_t dense_split_edge_bytes(int D, int T);      // bytes of the split section of the edge images: T bf16x3 images, then T f16x2 images
size_t dense_split_gru_bytes(int D);              // ... of the six GRU images, in the split kernel's stage order (both formats)
size_t dense_split_images_offset(int D, int T, int fmt);   // byte offset of the images of format `fmt` inside a split section of T images
int dense_split_pack_edge(const float* W, int T, int D, float* packed, hipStream_t st);
int dense_split_pack_gru(const float* Wg, const float* Wc, int D, float* packed, hipStream_t st);
int dense_split_launch(const DenseGraphArgs& a, int E, int D, int fmt, hipStream_t st);

}  // namespace ggnn
