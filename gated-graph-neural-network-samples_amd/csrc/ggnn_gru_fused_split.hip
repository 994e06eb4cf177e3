// The bf16-pipe (3-way split, ggnn_split.hpp) instantiations of the fused GRU: ggnn_gru_fused.hip compiled a second time, in a
// translation unit of its own because it is built WITHOUT packed-f32 vector instructions (build.py, NO_PACKED_F32): beside the
// partner wave's bf16 MFMAs a v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 stalls the SIMD for the length of the MFMA stream while
// plain v_fma_f32 and the transcendentals overlap with it (tools/mfma_overlap.hip -DBF16).  The f32-MFMA instantiations keep the
// packed forms: there every vector instruction costs matrix-pipe time anyway and packing halves their number.
#define GGNN_GRU_TU_SPLIT 1
#include "ggnn_gru_fused.hip"
