// The bf16-pipe (3-way split, ggnn_split.hpp) instantiations of the fused GRU backward: ggnn_gru_bwd_fused.hip compiled a second
// time, in a translation unit built WITHOUT packed-f32 vector instructions (build.py; see ggnn_gru_fused_split.hip).
#define GGNN_GRU_BWD_TU_SPLIT 1
#include "ggnn_gru_bwd_fused.hip"
