/* ggnn_hip.h -- C ABI of libggnn_hip.so: the MI355X (gfx950) GGNN propagation engine.
 *
 * This is the drop-in boundary for the hot path of microsoft/gated-graph-neural-network-samples:
 * SparseGGNNChemModel.compute_final_node_representations() (chem_tensorflow_sparse.py:117-218) and
 * its dense twin (chem_tensorflow_dense.py:93-117).  The reference has no FFI of its own (it is
 * pure Python on tensorflow==1.3.0); each entry point below replaces the TF op call sites cited
 * next to it.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer unless the
 *     parameter is documented "host".  The caller owns every buffer; the library allocates nothing.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no device
 *     synchronisation, no allocation, and is safe inside hipGraph capture.
 *   - return 0 on success, <0 on error (GGNN_E_*); ggnn_last_error() gives a thread-local message.
 *     Nothing throws or aborts across the ABI.
 *   - fp32 data, int32 indices (chem_tensorflow_sparse.py:65-71), row-major, rows 16-byte aligned,
 *     D % 4 == 0.  Supported hidden sizes D: multiples of 100, 64 or 32 (GGNN_E_UNSUPPORTED else).  Fused single-launch
 *     kernels exist for D = 32, 64, 100 (whole weight blocks as LDS stage images) and D = 128, 192, 256 (64-column
 *     panels of the weight blocks, ggnn_panel.hip); other sizes run the generic tiled GEMM kernels.
 *   - E_t = 0 and nodes with zero in-degree are valid (chem_tensorflow_sparse.py:346-347).
 */
#ifndef GGNN_HIP_H
#define GGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever the argument contract of an EXISTING entry point changes (not when symbols are added).
 *   2: ggnn_assemble_batch takes 11 output pointers (slot heads), ggnn_sparse_propagate_f32 / ggnn_gru_packed_gather_f32 take the
 *      slot-head table (round 3 changed these while the version still said 1; a caller built against that header must rebuild).
 *   3: (round 5) the operand format of the fused GRU forward is an ARGUMENT of every entry point that packs or consumes its weight
 *      images (`gru_fmt`, GGNN_GRU_FMT_*; see "Operand formats" below) instead of a process-wide environment setting:
 *      ggnn_gru_pack_weights_f32, ggnn_gru_packed_f32, ggnn_gru_packed_gather[_train]_f32, ggnn_sparse_propagate_f32,
 *      ggnn_sparse_train_prepare_f32, ggnn_sparse_train_forward_f32.  ggnn_gru_packed_bytes sizes a buffer for either format.
 *      The compacted message transform takes the format of its edge-weight images the same way: ggnn_edge_weights_pack_f32,
 *      ggnn_msg_transform_compact_f32 (`fmt`), ggnn_sparse_propagate_f32 (`edge_fmt`). */
#define GGNN_ABI_VERSION 3

#define GGNN_OK 0
#define GGNN_E_INVALID (-1)      /* bad argument (null pointer, negative size, misalignment) */
#define GGNN_E_UNSUPPORTED (-2)  /* shape outside the supported set */
#define GGNN_E_WORKSPACE (-3)    /* workspace too small */
#define GGNN_E_HIP (-4)          /* a HIP runtime call failed */
#define GGNN_E_INDEX (-5)        /* index out of range (validated entry points only) */

#define GGNN_ACT_TANH 0          /* chem_tensorflow_sparse.py:75-81 */
#define GGNN_ACT_RELU 1

typedef void* ggnn_stream_t;     /* hipStream_t */

int ggnn_abi_version(void);
const char* ggnn_last_error(void);
/* Matrix path of the fused kernels, fixed per process (environment GGNN_MATRIX, read at the first call):
 *   1 (default)  f32 products as six bf16 MFMA products of operands split exactly into three bf16 pieces (csrc/ggnn_split.hpp):
 *                f32 inputs, f32 accumulation, error bound of an f32 FMA chain, 2.5x the f32 matrix rate of gfx950;
 *   0 (GGNN_MATRIX=f32)  the f32 MFMA forms (v_mfma_f32_16x16x4_f32).
 * Packed weight images (ggnn_*_pack_*) are in the format of the mode and sized by the *_bytes functions. */
int ggnn_matrix_path_is_split(void);
/* Operand formats of the fused GRU FORWARD (ggnn_gru_*_f32 at hidden sizes 32 / 64 / 100 / 128 / 192 / 256) under the split matrix
 * path.  Chosen PER CALL by the `gru_fmt` argument of the entry points that pack or consume the GRU's weight images; an image must be
 * consumed in the format it was packed in.
 *   GGNN_GRU_FMT_BF16X3 (3; 0 means the same)  the EXACT three-piece bf16 split, six products per f32 product (the format of every
 *       other split-form kernel): f32 semantics for every finite f32 input, Inf / NaN stay non-finite.  Valid on ALL inputs.
 *   GGNN_GRU_FMT_F16X2 (2)  every f32 operand as TWO f16 pieces (round to nearest: 22 of its 24 significand bits), THREE f16 MFMA
 *       products per f32 product, f32 accumulation; weights packed x 2^8, activations unscaled.  Half the MFMAs, 48 KiB stage
 *       images.  Measured against f64 its error is below the six-product form's and the f32 MFMA's (tests/test_gpu_split_precision.py)
 *       -- INSIDE ITS OPERAND RANGE, which is a PRECONDITION the caller must establish:
 *           every GRU weight   |w| <= GGNN_F16X2_MAX_WEIGHT      (255.875 = 65504 / 2^8: beyond, the packed f16 piece overflows)
 *           every activation   |a| <= GGNN_F16X2_MAX_ACTIVATION  (65504: x segments, h, hence r*h; beyond, its f16 piece overflows)
 *           and all of them finite.
 *       Outside it the result is NOT the f32 result: an operand beyond the range becomes Inf / NaN pieces and the output non-finite
 *       (nothing is clamped or saturated silently).  ggnn_absmax_f32 below
 *       computes the maxima a caller needs; the Python host layer (formats.py) selects this format only when the bounds are PROVEN
 *       from max|h0|, the weights' maxima, the cell's activation and the aggregation -- and BF16X3 otherwise.
 * Under GGNN_MATRIX=f32 the argument is ignored (f32 MFMA kernels, f32 images).
 * ggnn_gru_forward_format(): the process default of the HOST POLICY (environment GGNN_GRU_FMT, read at the first call): 2 = "auto"
 * (F16X2 where the bounds are proven; the default), 3 = always BF16X3, 0 = the matrix path is not split.  The library's kernels do
 * not read it. */
#define GGNN_GRU_FMT_F16X2 2
#define GGNN_GRU_FMT_BF16X3 3
#define GGNN_F16X2_MAX_WEIGHT 255.875f
#define GGNN_F16X2_MAX_ACTIVATION 65504.0f
int ggnn_gru_forward_format(void);
/* Ring form of the gather-fused GRU launches (ggnn_gru_packed_gather[_train]_f32 and the drivers built on them): every form computes
 * the same products in the same order per accumulator -- results are bit-identical -- they differ in how a pass streams the stage
 * images and how many 16-row tiles a wave owns.  -1: the library's default per (fan-in, operand format); 0 / 1 / 2: the ring forms of
 * csrc/ggnn_gru_fused.hip (8 waves on whole images | two 4-wave workgroups per CU on half images | 8 waves, three half-image
 * slots); 6 / 62: the wide form of csrc/ggnn_gru_wide.hip (one wave per SIMD, two tiles per wave, gate-sequential stages; hidden
 * size 100, inference launch of the tanh cell in the two-piece f16 format -- other launches keep the ring forms), 61 / 64: the same
 * pass body as 8 waves x one tile / as two 4-wave workgroups per CU on half-image rings.  Process default: environment GGNN_GRU_FORM.  Returns the previous setting.  (Tests and experiments: compare forms
 * inside one process.) */
int ggnn_gru_form_set(int form);
/* out[i] = max |x| over the numel[i] floats at ptrs[i], i < n, in one launch per 32 tensors -- the operand-range check of
 * GGNN_GRU_FMT_F16X2.  A NaN anywhere in tensor i gives out[i] = NaN, an Inf gives Inf (the maximum is taken over the bit patterns
 * of |x|), so a host test `out[i] <= bound` fails on every non-finite input.  ptrs / numel: HOST arrays of n DEVICE pointers /
 * element counts (numel[i] == 0 gives 0); out: DEVICE [n] (read it back after synchronising the stream). */
int ggnn_absmax_f32(const float* const* ptrs, const int64_t* numel, int n, float* out, ggnn_stream_t stream);

/* ---- (a-1) message index prep: chem_tensorflow_sparse.py:120-129 -------------------------------
 * The reference concatenates the per-type target columns into message_targets[M] (type ascending,
 * list order).  We additionally bucket the M messages by target with a STABLE sort, so that the
 * segment sum can be done atomics-free with the reference's accumulation order inside each node.
 *
 *   adj        [M,2] int32  the T adjacency lists concatenated in type order, rows (src,dst)
 *   type_off   HOST [T+1]   type t owns rows type_off[t] .. type_off[t+1]-1 ; type_off[T] == M
 *   row_ptr    [V+1] int32  out: messages into node v are slots row_ptr[v] .. row_ptr[v+1]-1
 *   gather_row [M]   int32  out: slot -> src*T + type (row of the [V*T, D] transformed-state matrix)
 *   msg_perm   [M]   int32  out (may be NULL): slot -> original message index
 *   ws         scratch of at least ggnn_csr_workspace_bytes(M, V) bytes
 * Indices are validated on the device: out-of-range src/dst set *err_flag (device int32, may be
 * NULL) to 1 and are clamped, instead of faulting (TF-CPU raises InvalidArgument there).
 */
size_t ggnn_csr_workspace_bytes(int64_t M, int V);
int ggnn_build_target_csr(const int32_t* adj, const int64_t* type_off, int T, int V, int64_t M,
                          int32_t* row_ptr, int32_t* gather_row, int32_t* msg_perm, int32_t* err_flag,
                          void* ws, size_t ws_bytes, ggnn_stream_t stream);

/* The transpose index for the backward pass (a-B): messages bucketed by (src*T + type) -- V*T segments,
 * row_ptr [V*T+1] -- with gather_row[slot] = dst, so that
 *   dH[src*T+type, :] = sum over the messages leaving (src,type) of d_incoming[dst, :]
 * is the SAME gather/segment-sum kernel as the forward pass (ggnn_gather_segment_sum_f32 with V*T
 * segments).  Same workspace size and argument meaning as ggnn_build_target_csr. */
int ggnn_build_source_csr(const int32_t* adj, const int64_t* type_off, int T, int V, int64_t M,
                          int32_t* row_ptr, int32_t* gather_row, int32_t* msg_perm, int32_t* err_flag,
                          void* ws, size_t ws_bytes, ggnn_stream_t stream);

/* ---- (a-3) per-edge-type message transform: chem_tensorflow_sparse.py:160-164 -------------------
 * H[v, t*D:(t+1)*D] = h[v,:] @ W[t]  for all nodes and types in ONE [V,D]x[D,T*D] FP32-MFMA GEMM
 * (transform-first: (h[src]) W == (h W)[src]).
 *   h [V,D] (row stride ldh floats), W [T,D,D] (the reference's reshaped edge_weights, :90),
 *   H [V,T*D] out.
 */
int ggnn_msg_transform_f32(const float* h, int ldh, const float* W, float* H, int V, int D, int T,
                           ggnn_stream_t stream);

/* ---- (a-3, compacted) message transform over the ACTIVE (source node, edge type) pairs only ----------
 * The reference transforms every message row h[src] W_t (M rows, :161-164); a node with several outgoing
 * edges of one type sends the same row several times, and most (node,type) pairs emit nothing.  The
 * active pairs are enumerated once per batch, type-major / node-ascending ("compact rows"):
 *
 *   ggnn_build_compact_sources   src_row_ptr [V*T+1] (row_ptr of ggnn_build_source_csr) ->
 *                                pair_node [R] (compact row -> node; caller sizes it min(M, V*T)),
 *                                pair_id [V*T] ((v*T+t) -> compact row, -1 if inactive),
 *                                type_row_off DEVICE [T+1] (rows of type t are type_row_off[t] .. [t+1]-1)
 *   ggnn_remap_gather_rows       gather_row [M] (src*T+type) -> compact rows, for ggnn_gather_segment_sum_f32
 *   ggnn_msg_transform_compact_f32  Hc[r,:] = h[pair_node[r],:] @ W[type(r)]   (Hc [R,D]; type_row_off on the HOST)
 * Supported hidden sizes: ggnn_msg_transform_compact_supported(D) (32, 64, 100).
 */
int ggnn_msg_transform_compact_supported(int D);
size_t ggnn_compact_workspace_bytes(int V, int T);
int ggnn_build_compact_sources(const int32_t* src_row_ptr, int V, int T, int32_t* pair_node, int32_t* pair_id,
                               int32_t* type_row_off, void* ws, size_t ws_bytes, ggnn_stream_t stream);
int ggnn_remap_gather_rows(const int32_t* gather_row, const int32_t* pair_id, int32_t* gather_row_compact, int64_t M,
                           ggnn_stream_t stream);
size_t ggnn_msg_transform_compact_workspace_bytes(int D, int T);
int ggnn_msg_transform_compact_f32(const float* h, const float* W, const int32_t* pair_node, const int64_t* type_row_off,
                                   float* Hc, void* ws, size_t ws_bytes, int V, int D, int T, int fmt, ggnn_stream_t stream);

/* ---- (a-2,a-4..a-7) gather + segment sum + bias + mean: chem_tensorflow_sparse.py:160-162,168,
 *      198-209 ------------------------------------------------------------------------------------
 * out[v,:] = ( sum_{slot in row_ptr[v]..row_ptr[v+1]} Hrows[gather_row[slot], :]
 *              + sum_t nin[v,t]*bias[t,:] )  /  ( sum_t nin[v,t] + 1e-7 )
 * with the bias term only if bias != NULL (:202-204) and the division only if use_avg (:206-209).
 * Nodes without incoming messages get the zero row (tf.unsorted_segment_sum semantics).
 *   Hrows  rows of D floats (H of ggnn_msg_transform_f32 viewed as [V*T, D])
 *   nin    [V,T] fp32 incoming-edge counts (required if bias or use_avg), out [V,D].
 */
int ggnn_gather_segment_sum_f32(const float* Hrows, const int32_t* row_ptr, const int32_t* gather_row,
                                const float* nin, const float* bias, int use_avg, float* out,
                                int V, int D, int T, ggnn_stream_t stream);

/* use_propagation_attention (chem_tensorflow_sparse.py:147-149, 170-196) fused into the segment sum:
 *   score_m = <h[src_m], h[tgt_m]> * type_factors[type_m];  a = per-target softmax of the scores (max-shifted,
 *   denominator + 1e-7);  incoming[v] = sum_m a_m * Hrows[gather_row[m]]  (then bias / mean as above).
 * gather_row must be the dense form src*T + type (ggnn_build_target_csr); h [V,D] are the un-transformed states;
 * type_factors [T] = edge_type_attention_weights (:94-96).  D <= 256. */
int ggnn_gather_segment_sum_attn_f32(const float* Hrows, const float* h, const int32_t* row_ptr, const int32_t* gather_row,
                                     const float* type_factors, const float* nin, const float* bias, int use_avg,
                                     float* out, int V, int D, int T, ggnn_stream_t stream);

/* Slot heads: heads[v] (int32 [V,4], 16-byte aligned) = the gather rows of the first four message slots of node v (-1 = no such
 * slot), built once per batch from (row_ptr, gather_row).  ggnn_gather_segment_sum_heads_f32 is ggnn_gather_segment_sum_f32 with
 * them (bit-identical results): the slot indices of a node arrive with one load that does not depend on row_ptr, so a lane has
 * its source rows in flight after ONE round trip instead of two; accumulate != 0 adds to `out` instead of overwriting it. */
int ggnn_build_slot_heads(const int32_t* row_ptr, const int32_t* gather_row, int32_t* heads, int V, ggnn_stream_t stream);
int ggnn_gather_segment_sum_heads_f32(const float* Hrows, const int32_t* row_ptr, const int32_t* gather_row, const int32_t* heads,
                                      const float* nin, const float* bias, int use_avg, float* out, int V, int D, int T,
                                      int accumulate, ggnn_stream_t stream);

/* Backward of the propagation attention (TF autodiff of chem_tensorflow_sparse.py:170-196), three kernels:
 * ggnn_attn_bwd_target_f32: per target node (by-target index: row_ptr, gather_row = src*T+type, msg_perm = slot -> message id),
 *   from d_att = dL/d(attention-weighted sum) [V,D]: the target-side state gradient dh[v] (+)= sum_e ds_e f_t h[src_e], and per
 *   MESSAGE (indexed by message id): coef_a = softmax weight a_e, coef_s = ds_e f_t, dfac = ds_e <h_src, h_tgt>;
 * ggnn_weighted_segment_sum_f32: out[seg] (+)= sum_slots weights[weight_id[slot]] * rows[gather_row[slot]] -- with the by-source
 *   index it is dH[src,type] = sum a_e d_att[dst] (weights = coef_a) and dh[src] += sum ds_e f_t h[dst] (weights = coef_s);
 * ggnn_range_sum_f32: out[b] = sum values[range_off[b] .. range_off[b+1]) (HOST offsets; d attention factor of type t: messages are
 *   type-major).  All deterministic. */
int ggnn_attn_bwd_target_f32(const float* Hrows, const float* h, const float* d_att, const int32_t* row_ptr,
                             const int32_t* gather_row, const int32_t* msg_perm, const float* type_factors, float* coef_a,
                             float* coef_s, float* dfac, float* dh, int accumulate, int V, int D, int T, ggnn_stream_t stream);
int ggnn_weighted_segment_sum_f32(const float* rows, const int32_t* row_ptr, const int32_t* gather_row, const int32_t* weight_id,
                                  const float* weights, float* out, int accumulate, int num_segments, int D, ggnn_stream_t stream);
int ggnn_range_sum_f32(const float* values, const int64_t* range_off, int num_ranges, float* out, ggnn_stream_t stream);

/* tf.unsorted_segment_sum in its general form (chem_tensorflow_sparse.py:198-200, 226-228):
 * out[ids[m],:] += data[m,:] with out zero-filled first; fp32 atomics, any id order.  Used for the
 * readout's per-graph sum and available for un-bucketed message lists. */
int ggnn_unsorted_segment_sum_f32(const float* data, const int32_t* ids, float* out, int64_t M, int D,
                                  int num_segments, ggnn_stream_t stream);

/* ---- (a-R) fused graph-level readout: chem_tensorflow_sparse.py:220-231 + utils.py:39-70 -------------
 * out[g] = sum_{v in graph g} sigmoid([hT[v] | h0[v]] . gate_W + gate_b) * (hT[v] . transform_W + transform_b)
 *   hT, h0 [V,D]; graph_nodes_list [V] int32 (:71, :304); gate_W [2D] (the [2D,1] MLP weight), transform_W [D];
 *   gate_b, transform_b DEVICE [1]; out [num_graphs] (zero-filled by the call).  fp32 atomics (one add per node): the
 *   form for an UNSORTED graph_nodes_list; the models use ggnn_readout_loss_fwd_f32 (deterministic) for batcher output. */
int ggnn_gated_readout_f32(const float* hT, const float* h0, const int32_t* graph_nodes_list, const float* gate_W,
                           const float* gate_b, const float* transform_W, const float* transform_b, float* out, int V,
                           int D, int num_graphs, ggnn_stream_t stream);

/* ---- (f-2) fused readout + masked loss, forward and backward ------------------------------------------------------------
 * chem_tensorflow_sparse.py:220-231 (gated_regression; dense: chem_tensorflow_dense.py:119-129 with node_mask) +
 * chem_tensorflow.py:158-170 (masked loss / MAE of one task) + utils.py:39-70 (MLP with hid_sizes = []).
 * DETERMINISTIC and atomics-free: graph_nodes_list must be NON-DECREASING (the reference batchers append graph after
 * graph, :297-304), so out[g] is a segmented sum in node order; all cross-block reductions run in a fixed order.
 *
 * ggnn_readout_loss_fwd_f32
 *   hT, h0 [V,D]; graph_nodes_list [V] int32 sorted; graph_ptr [G+1] int32 or NULL (first node of every graph; NULL: found by
 *   binary search); node_mask [V] or NULL (dense model: 0 for padding vertices); gate_W [2D], transform_W [D]; gate_b,
 *   transform_b DEVICE [1]; target, mask [G] or NULL (this task's row of target_values / target_mask)
 *   out [G]:  out[g] = sum_{v in g} sigmoid([hT|h0][v] . gate_W + gate_b) (hT[v] . transform_W + transform_b) (node_mask[v])
 *   node_gate, node_val [V]: the per-node gate and value, kept for the backward pass
 *   stats DEVICE [3] or NULL: sum_g 0.5 diff_g^2, sum_g |diff_g|, sum_g mask_g with diff = (out - target) mask   (:161-166; the
 *   caller divides by (sum mask + 1e-7) -- under data parallelism by the ALL-REDUCED mask count)
 *   ws: ggnn_readout_workspace_bytes(V, D, G) bytes.
 * ggnn_readout_loss_bwd_f32
 *   d_out [G] or NULL (gradient w.r.t. out), d_stats DEVICE [2] or NULL (gradients w.r.t. stats[0] and stats[1]);
 *   d_hT [V,D] written (accumulate = 0) or added to (accumulate != 0: second and later tasks);
 *   d_gate_W [2D], d_gate_b [1], d_transform_W [D], d_transform_b [1] written.  D <= 256. */
size_t ggnn_readout_workspace_bytes(int V, int D, int num_graphs);
int ggnn_readout_loss_fwd_f32(const float* hT, const float* h0, const int32_t* graph_nodes_list, const int32_t* graph_ptr,
                              const float* node_mask, const float* gate_W, const float* gate_b, const float* transform_W,
                              const float* transform_b, const float* target, const float* mask, float* out,
                              float* node_gate, float* node_val, float* stats, void* ws, size_t ws_bytes, int V, int D,
                              int num_graphs, ggnn_stream_t stream);
int ggnn_readout_loss_bwd_f32(const float* hT, const float* h0, const int32_t* graph_nodes_list, const float* node_mask,
                              const float* gate_W, const float* transform_W, const float* node_gate, const float* node_val,
                              const float* out, const float* target, const float* mask, const float* d_out,
                              const float* d_stats, float* d_hT, int accumulate, float* d_gate_W, float* d_gate_b,
                              float* d_transform_W, float* d_transform_b, void* ws, size_t ws_bytes, int V, int D,
                              int num_graphs, ggnn_stream_t stream);

/* ---- (a-8, a-G) residual concat + GRU node update: chem_tensorflow_sparse.py:211-216 -----------
 * TF-1.3 GRUCell: [r|u] = sigmoid([x|h] Wg + bg); c = act([x | r*h] Wc + bc); h' = u*h + (1-u)*c
 * with x = [x_segs[0] | ... | x_segs[nx-1]] read through nx pointers (no concat is materialised;
 * residual states first, aggregated messages last, :211-212).
 *   x_segs HOST array of nx (1..3) device pointers to [V,D]; h [V,D]; Wg [(nx+1)D, 2D]; bg [2D];
 *   Wc [(nx+1)D, D]; bc [D]; h_out [V,D] (must not alias h);
 *   ws: ggnn_gru_workspace_bytes(V,D) bytes of scratch (r*h and u);
 *   save_r/save_u/save_c: optional [V,D] outputs for a backward pass (NULL to skip).
 */
size_t ggnn_gru_workspace_bytes(int V, int D);
int ggnn_gru_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                 const float* Wc, const float* bc, float* h_out, void* ws, size_t ws_bytes,
                 float* save_r, float* save_u, float* save_c, int V, int D, int act,
                 ggnn_stream_t stream);

/* The other two cell types of chem_tensorflow_sparse.py:102-112 (forward):
 *   ggnn_rnn_f32        tf.nn.rnn_cell.BasicRNNCell (:109-110): h' = act([x|h] W + b), W [(nx+1)D, D], b [D]
 *   ggnn_cudnn_gru_f32  tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell (:105-108): gates as GRUCell, then
 *                       c = tanh(x Wcx + bcx + r * (h Wch + bch)); h' = u*h + (1-u)*c
 *                       Wg [(nx+1)D,2D], Wcx [nx*D, D], Wch [D,D]; ws = ggnn_cudnn_gru_workspace_bytes(V,D). */
int ggnn_rnn_f32(const float* const* x_segs, int nx, const float* h, const float* W, const float* b, float* h_out,
                 int V, int D, int act, ggnn_stream_t stream);
size_t ggnn_cudnn_gru_workspace_bytes(int V, int D);
int ggnn_cudnn_gru_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                       const float* Wcx, const float* bcx, const float* Wch, const float* bch, float* h_out,
                       void* ws, size_t ws_bytes, int V, int D, ggnn_stream_t stream);

/* Non-zero if ggnn_gru_f32 runs as ONE fused launch for this hidden size (gates -> r*h -> candidate -> blend chained in
 * registers): 1 for D in {32, 64, 100} (whole-block stage images; ggnn_gru_packed_gather_f32 exists for these),
 * 2 for D in {128, 192, 256} (column-panel kernel, ggnn_panel.hip; no gather-fused variant);
 * 0 if it runs as the two launches below (ws is then used). */
int ggnn_gru_is_fused(int D);

/* Pre-packed weights (inference: weights are constant across batches).  The fused GRU and the compacted
 * transform consume k-interleaved LDS stage images of their weights; ggnn_gru_f32 /
 * ggnn_msg_transform_compact_f32 build them in a small pre-pass on every call, the functions below let the
 * caller build them once per weight version:
 *   ggnn_gru_pack_weights_f32   Wg [(nx+1)D,2D], Wc [(nx+1)D,D] -> packed (ggnn_gru_packed_bytes(D,nx) bytes: enough for either
 *                               operand format) in the format gru_fmt (GGNN_GRU_FMT_*, see "Operand formats" at the top)
 *   ggnn_gru_packed_f32         == ggnn_gru_f32 with the packed images instead of Wg / Wc (fused sizes only); gru_fmt = the format
 *                               the images were packed in.  (ggnn_gru_f32 itself, on raw weights, always multiplies in BF16X3.)
 *   ggnn_edge_weights_pack_f32  W [T,D,D] -> packed (ggnn_msg_transform_compact_workspace_bytes(D,T) bytes: enough for either
 *                               operand format), in the format `fmt` (GGNN_GRU_FMT_*; "Operand formats" at the top: BF16X3 is
 *                               exact on every input; F16X2 needs |h| <= 65504 and |W| <= 255.875 PROVEN by the caller --
 *                               forward transforms only; hidden sizes 32 / 64 / 100 and the ring kernel of 128 / 192 / 256);
 *                               then call ggnn_msg_transform_compact_f32 with W = NULL and ws = packed.
 * tile_counter (ggnn_gru_packed_f32, ggnn_gru_packed_gather_f32): NULL, or a DEVICE int32 that is 0 when the launch
 * starts (the kernel leaves it non-zero).  With a counter the 16-row tiles are handed to the workgroups dynamically:
 * same results, but the launch no longer stretches when other streams hold part of the GPU while it starts. */
size_t ggnn_gru_packed_bytes(int D, int nx);
int ggnn_gru_pack_weights_f32(const float* Wg, const float* Wc, int nx, int D, int gru_fmt, float* packed, ggnn_stream_t stream);
int ggnn_gru_packed_f32(const float* const* x_segs, int nx, const float* h, const float* packed, const float* bg,
                        const float* bc, float* h_out, float* save_r, float* save_u, float* save_c, int V, int D, int act,
                        int gru_fmt, int32_t* tile_counter, ggnn_stream_t stream);
int ggnn_edge_weights_pack_f32(const float* W, int T, int D, int fmt, float* packed, ggnn_stream_t stream);

/* GRU with the segment sum fused in (chem_tensorflow_sparse.py:198-216 in one launch, no edge bias): the
 * aggregated-messages input -- the LAST of the nx concatenated inputs -- is gathered inside the kernel,
 *   incoming[v] = (sum over the slots row_ptr[v]..row_ptr[v+1] of Hrows[gather_row[slot]]) / (sum_t nin[v,t] + 1e-7),
 * in the same slot order and arithmetic as ggnn_gather_segment_sum_f32 (bit-identical results), so the separate
 * segment-sum launch and the HBM round trip of `incoming` disappear.  x_segs: the nx-1 residual segments.
 * Hrows must hold at least one row (row 0 is fetched for the unused slots of low-degree nodes); V*T*D < 2^30. */
int ggnn_gru_packed_gather_f32(const float* const* x_segs, int nx, const float* h, const float* packed, const float* bg,
                               const float* bc, float* h_out, const float* Hrows, const int32_t* row_ptr,
                               const int32_t* gather_row, const float* nin, int T, int use_avg, int V, int D, int act,
                               int gru_fmt, int32_t* tile_counter, ggnn_stream_t stream);
/* The training form of the same launch: r, u, c [V,D] and the gathered segment `incoming` [V,D] (an operand of the weight gradients)
 * are written for the backward pass -- all four pointers or none. */
int ggnn_gru_packed_gather_train_f32(const float* const* x_segs, int nx, const float* h, const float* packed, const float* bg,
                                     const float* bc, float* h_out, const float* Hrows, const int32_t* row_ptr,
                                     const int32_t* gather_row, const float* nin, int T, int use_avg, float* save_r, float* save_u,
                                     float* save_c, float* save_incoming, int V, int D, int act, int gru_fmt,
                                     int32_t* tile_counter, ggnn_stream_t stream);

/* The two launches of the un-fused ggnn_gru_f32, separately addressable (profiling, large D):
 *   gates:     [r|u] = sigmoid([x|h] Wg + bg) -> rh = r*h [V,D], u [V,D] (save_r optional)
 *   candidate: c = act([x|rh] Wc + bc); h_out = u*h + (1-u)*c            (save_c optional) */
int ggnn_gru_gates_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                       float* rh, float* u, float* save_r, int V, int D, ggnn_stream_t stream);
int ggnn_gru_candidate_f32(const float* const* x_segs, int nx, const float* rh, const float* h, const float* u,
                           const float* Wc, const float* bc, float* h_out, float* save_c, int V, int D, int act,
                           ggnn_stream_t stream);

/* ---- (a-9) layer / timestep driver: chem_tensorflow_sparse.py:131-218 -------------------------------
 * The whole forward propagation (every layer and timestep: transform -> gather/segment-sum -> GRU) enqueued
 * by ONE call.  All `const T* const*` parameters are HOST arrays of `num_layers` DEVICE pointers.
 *   row_ptr / gather_row        message index by target (ggnn_build_target_csr); with the compacted transform
 *                               gather_row must be the remapped rows (ggnn_remap_gather_rows)
 *   pair_node, type_row_off     compacted transform (type_row_off on the HOST); both NULL -> dense transform
 *   layer_timesteps             HOST [num_layers]                                  (:53, :131)
 *   res_ptr / res_idx           HOST CSR of residual_connections: layer l reads the states
 *                               res_idx[res_ptr[l] .. res_ptr[l+1]-1] (0 = h0, k = output of layer k-1), :140-145
 *   edge_w [T,D,D] raw and/or edge_packed (ggnn_edge_weights_pack_f32); edge_bias entries may be NULL
 *   Wg/Wc raw and/or gru_packed (ggnn_gru_pack_weights_f32); bg [2D], bc [D]
 *   gru_fmt                     HOST [num_layers] of GGNN_GRU_FMT_*: the format gru_packed[l] was packed in (NULL: BF16X3 for all;
 *                               layers that run on raw weights always multiply in BF16X3)
 *   edge_fmt                    the same for edge_packed[l] (ggnn_edge_weights_pack_f32's fmt; NULL: BF16X3 for all)
 *   layer_out                   HOST [num_layers] of DEVICE [V,D]: node_states_per_layer[l+1]; the last one is
 *                               the function's return value (:218)
 *   fuse_gather                 k > 0: layers with at most k concatenated GRU inputs (residuals + messages; 1 = no
 *                               residual inputs), packed GRU weights, a fused hidden size and no edge bias run 2 launches
 *                               per timestep (transform, ggnn_gru_packed_gather_f32) instead of 3;  0: never
 *   ws                          ggnn_sparse_propagate_workspace_bytes(V, D, T, compact_rows or -1) bytes
 */
size_t ggnn_sparse_propagate_workspace_bytes(int V, int D, int T, int64_t compact_rows);
int ggnn_sparse_propagate_f32(const float* h0, int V, int D, int T,
                              const int32_t* row_ptr, const int32_t* gather_row, const int32_t* pair_node,
                              const int64_t* type_row_off, const float* nin, int use_avg,
                              int num_layers, const int32_t* layer_timesteps, const int32_t* res_ptr, const int32_t* res_idx,
                              const float* const* edge_w, const float* const* edge_packed, const float* const* edge_bias,
                              const float* const* Wg, const float* const* bg, const float* const* Wc, const float* const* bc,
                              const float* const* gru_packed, const int32_t* gru_fmt, const int32_t* edge_fmt, int act,
                              int fuse_gather, float* const* layer_out, void* ws, size_t ws_bytes, ggnn_stream_t stream);

/* ---- (a-B) element-wise stages of the GRU backward (TF autodiff of GRUCell, chem_tensorflow.py:184) -------
 * stage 1: dpc = g*(1-u)*act'(c) -> dpc [V,D];  g*(h-c)*u*(1-u) -> dpg[:, D:2D];  g*u -> dh [V,D];
 *          r*h -> a_c[:, col0:col0+D] (row stride lda: the [x | r*h] operand of the dWc product)
 * stage 2: dh += drh*r;  drh*h*r*(1-r) -> dpg[:, 0:D]   (drh [V,D] with row stride ld_drh)
 * The dense contractions between the stages (dX = dY W^T, dW = X^T dY) are left to the vendor BLAS. */
int ggnn_gru_bwd_stage1_f32(const float* g, const float* h, const float* r, const float* u, const float* c, int act,
                            float* dpc, float* dpg, float* dh, float* a_c, int lda, int col0, int V, int D,
                            ggnn_stream_t stream);
int ggnn_gru_bwd_stage2_f32(const float* drh, int ld_drh, const float* h, const float* r, float* dh, float* dpg,
                            int V, int D, ggnn_stream_t stream);

/* ---- (a-D) dense-adjacency aggregation: chem_tensorflow_dense.py:103-112 ---------------------------
 * acts[g,i,:] = sum_e sum_j A[g,e,i,j] * ( Hm[g*v+j, e*D:(e+1)*D] + bias[e,:] )
 *   A [b,e,v,v] fp32 (A[g,e,dst,src], chem_tensorflow_dense.py:30-36), Hm [b*v, e*D] = h W_e for all e
 *   (ggnn_msg_transform_f32 output), bias [e,D] or NULL (:107-108), acts [b*v, D] out.
 * The dense step is ggnn_msg_transform_f32 -> ggnn_dense_aggregate_f32 -> ggnn_gru_f32 (nx = 1). */
/* The WHOLE dense forward (chem_tensorflow_dense.py:93-117, `steps` timesteps of  h <- GRU(sum_e A_e (h W_e + b_e), h)) in one launch:
 * workgroup g keeps graph g's v <= 32 vertex states on its CU for all timesteps (a graph reads only its own vertices).
 *   ggnn_dense_propagate_supported(v, E, D): v <= 32, E in {2,4,6,8}, D in {32,64,100}.
 *   edge_packed: ggnn_dense_edge_pack_f32 of W [E,D,D] (ggnn_dense_edge_packed_bytes(D,E) bytes: the f32 stage images followed by the
 *   split ones);  gru_packed: ggnn_dense_gru_pack_f32 of Wg [2D,2D], Wc [2D,D] (ggnn_dense_gru_packed_bytes(D) bytes, likewise);
 *   edge_bias [E,D] or NULL;  h0, out [b,v,D];  A [b,E,v,v] (A[g,e,dst,src]).
 *   ggnn_dense_propagate_is_split(v, E, D): 1 when the launch runs the split-form kernel (bf16 matrix pipe, f32 arithmetic: the
 *   process's default matrix path, and the kernel's LDS blocks fit E edge types), 0 for the f32-MFMA kernel.
 *   fmt (split-form launches; ABI 3): operand format of every D x D product, GGNN_GRU_FMT_BF16X3 (exact; 0 means the same) or
 *   GGNN_GRU_FMT_F16X2 under the precondition of "Operand formats" above -- here: every state, every aggregated message
 *   (|acts| <= v E (D max|W_e| max|h| + max|b_e|)) and r*h within 65504, every weight within 255.875.  The packed buffers hold the
 *   images of both formats (f32 | bf16x3 | f16x2 sections). */
int ggnn_dense_propagate_supported(int v, int E, int D);
int ggnn_dense_propagate_is_split(int v, int E, int D);
size_t ggnn_dense_edge_packed_bytes(int D, int T);
int ggnn_dense_edge_pack_f32(const float* W, int T, int D, float* packed, ggnn_stream_t stream);
size_t ggnn_dense_gru_packed_bytes(int D);
int ggnn_dense_gru_pack_f32(const float* Wg, const float* Wc, int D, float* packed, ggnn_stream_t stream);
int ggnn_dense_propagate_f32(const float* h0, const float* A, const float* edge_packed, const float* gru_packed, const float* edge_bias,
                             const float* bg, const float* bc, float* out, int b, int v, int E, int D, int steps, int fmt,
                             ggnn_stream_t stream);
int ggnn_dense_aggregate_f32(const float* A, const float* Hm, const float* bias, float* acts, int b, int v,
                             int e, int D, ggnn_stream_t stream);

/* Generic FP32-MFMA GEMM used by the above and by the host layer for the backward pass:
 * C[M,N] = [A0 | A1 | ...] (nseg <= 4 segments of width D each, row stride lda floats, K = nseg*D) x B[K,N].
 * a_segs is a HOST array of device pointers. */
int ggnn_gemm_f32(const float* const* a_segs, int nseg, int D, int lda, const float* B, int ldb, float* C, int ldc,
                  int M, int N, ggnn_stream_t stream);

/* Weight-gradient product of the backward pass (a-B; TF autodiff of tf.matmul, chem_tensorflow.py:184):
 * C[K,N] = A[M,K]^T B[M,N], M ~ 1e5 rows reduced into a small matrix.  The rows are split over the whole GPU and
 * the per-split partials are added in a fixed order (deterministic).  N a multiple of 4, <= 512; A row stride lda,
 * B row stride ldb (multiple of 4, B 16-byte aligned); ws: ggnn_gemm_tn_workspace_bytes(M, K, N) bytes. */
size_t ggnn_gemm_tn_workspace_bytes(int M, int K, int N);
int ggnn_gemm_tn_f32(const float* A, int lda, const float* B, int ldb, float* C, int M, int K, int N, void* ws,
                     size_t ws_bytes, ggnn_stream_t stream);

/* ---- (a-B) backward GEMMs of one propagation timestep, hand-written (ggnn_bwd_gemm.hip) ---------------------------------------
 * What TF autodiff derives from chem_tensorflow_sparse.py:160-164, 211-216 through compute_gradients (chem_tensorflow.py:184).
 *
 * ggnn_xty_f32: C[b] = X[rows of batch b]^T Y[rows of batch b]   -- every weight gradient (dW = X^T dY).
 *   X is given as nseg <= 4 column segments of Dseg columns each (HOST arrays x_segs / ldx: pointers and row strides), so the
 *   [residuals | incoming | h] operand of the GRU kernels is never concatenated; x_rows (device int32, or NULL) gathers the X
 *   rows (edge-weight gradients: X row of compact row r = h[pair_node[r]]); row_off HOST [nbatch+1] splits the rows into batches
 *   with one [K,N] output each (one per edge type).  K = nseg * Dseg, N <= 208, N % 4 == 0.  Deterministic (fixed-order split
 *   reduction).  ones_row != 0: X gets a virtual column of ones, i.e. C is [K+1, N] per batch and its last row is the column
 *   sum of Y -- the bias gradient comes out of the same pass.  ws: ggnn_xty_workspace_bytes(largest batch, K, N, nbatch).
 * ggnn_colsum_f32: out[n] = sum_v Y[v, n]  (bias gradients), deterministic.
 * ggnn_gru_bwd_dx_cand_f32:  P = dpc Wc^T (WcT = Wc^T, [D, (nx+1)D] row-major):  dx [V, nx*D] = P[:, x columns];
 *   dh += P[:, h columns] * r;  dpg[:, 0:D] = P[:, h columns] * h * r * (1 - r)       (the stage-2 algebra, fused)
 * ggnn_gru_bwd_dx_gates_f32: Q = dpg Wg^T (WgT [2D, (nx+1)D]):  dx[:, residual columns] += Q;
 *   dinc [V,D] = (dx[:, last segment] + Q) (/ (sum_t nin + 1e-7) with use_avg, chem_tensorflow_sparse.py:206-209);  dh += Q[:, h columns]
 * ggnn_gather_segment_sum_acc_f32: out[v] += sum of the gathered rows (several gradient contributions meet in one tensor). */
size_t ggnn_xty_workspace_bytes(int M_max, int K, int N, int nbatch);
int ggnn_xty_f32(const float* const* x_segs, int nseg, int Dseg, const int32_t* ldx, const int32_t* x_rows, const float* Y,
                 int ldy, float* C, int K, int N, int ones_row, const int32_t* row_off, int nbatch, void* ws, size_t ws_bytes,
                 ggnn_stream_t stream);
/* The same product written the way a training step consumes it (what TF's gradient accumulation over the timesteps of a layer
 * does, chem_tensorflow.py:184):  Cb != NULL (needs ones_row): the K weight rows go to C [nbatch][K][N] and the column sums to
 * Cb [nbatch][N];  accumulate != 0: the results are ADDED to the destinations' contents. */
int ggnn_xty_acc_f32(const float* const* x_segs, int nseg, int Dseg, const int32_t* ldx, const int32_t* x_rows, const float* Y,
                     int ldy, float* C, float* Cb, int accumulate, int K, int N, int ones_row, const int32_t* row_off, int nbatch,
                     void* ws, size_t ws_bytes, ggnn_stream_t stream);
size_t ggnn_colsum_workspace_bytes(int N);
int ggnn_colsum_f32(const float* Y, int ldy, int M, int N, float* out, void* ws, size_t ws_bytes, ggnn_stream_t stream);
int ggnn_gru_bwd_dx_cand_f32(const float* dpc, const float* WcT, const float* h, const float* r, float* dx, float* dh,
                             float* dpg, int nx, int V, int D, ggnn_stream_t stream);
int ggnn_gru_bwd_dx_gates_f32(const float* dpg, const float* WgT, float* dx, float* dinc, const float* nin, int T,
                              int use_avg, float* dh, int nx, int V, int D, ggnn_stream_t stream);
int ggnn_gather_segment_sum_acc_f32(const float* Hrows, const int32_t* row_ptr, const int32_t* gather_row, float* out,
                                    int V, int D, ggnn_stream_t stream);

/* Backward building blocks of the other two cell types (chem_tensorflow_sparse.py:105-110) and of any product dX = dY W^T:
 * ggnn_bwd_dx_f32: Q = dY WT with dY [V, nseg_y*D] (column segments of one matrix, row stride ldy) and WT [nseg_y*D, K];
 *   columns [0, xcols) of Q are the x segments (dx is [V, xcols]: written or added to; with split_inc the LAST D of them are
 *   the aggregated messages and go to dinc [V,D] = (acc_dx ? dx + Q : Q), divided by (sum_t nin + 1e-7) with use_avg);
 *   columns [xcols, K) (K == xcols + D; or K == xcols: no h block) -> dh (written, or added to with acc_dh).
 * ggnn_act_bwd_f32: dP = g * act'(out) for out = act(P)   (BasicRNNCell).
 * ggnn_cudnn_gru_train_f32: ggnn_cudnn_gru_f32 that also keeps the candidate c; afterwards ws holds [r*h | u | r | hc] ([V,D] each).
 * ggnn_cudnn_gru_bwd_stage_f32: dpc = g(1-u)(1-c^2); dpg = [dpc hc r(1-r) | g(h-c)u(1-u)]; dh = g u; dhc = dpc r. */
int ggnn_bwd_dx_f32(const float* dY, int ldy, int nseg_y, const float* WT, int K, float* dx, int xcols, int split_inc, float* dinc,
                    const float* nin, int T, int use_avg, float* dh, int acc_dx, int acc_dh, int V, int D, ggnn_stream_t stream);
int ggnn_act_bwd_f32(const float* g, const float* out, int act, float* dP, int V, int D, ggnn_stream_t stream);
int ggnn_cudnn_gru_train_f32(const float* const* x_segs, int nx, const float* h, const float* Wg, const float* bg,
                             const float* Wcx, const float* bcx, const float* Wch, const float* bch, float* h_out,
                             float* save_c, void* ws, size_t ws_bytes, int V, int D, ggnn_stream_t stream);
int ggnn_cudnn_gru_bwd_stage_f32(const float* g, const float* h, const float* r, const float* u, const float* c, const float* hc,
                                 float* dpc, float* dpg, float* dh, float* dhc, int V, int D, ggnn_stream_t stream);

/* The whole GRU backward of a timestep in ONE launch (D in {32, 64, 100}; ggnn_gru_bwd_fused.hip) -- the mirror image of the
 * fused forward kernel: from g = dL/dh' and the saved h, r, u, c it computes, chained through registers,
 *   dpc = g (1-u) act'(c);  dpu = g (h-c) u (1-u);  drh = dpc Wc^T[h rows];  dpr = drh h r (1-r);
 *   dh  = g u + drh r + [dpr|dpu] Wg^T[h rows];   dx[s] = dpc Wc^T[x_s rows] + [dpr|dpu] Wg^T[x_s rows]   (s < nx)
 *   dx[nx-1] is divided by (sum_t nin + 1e-7) when use_avg (d_incoming of the mean aggregation, chem_tensorflow_sparse.py:206-209)
 * and writes dpc [V,D], dpg = [dpr|dpu] [V,2D] and rh = r*h [V,D] for the weight-gradient products (ggnn_xty_f32).
 *   Wg [(nx+1)D, 2D], Wc [(nx+1)D, D] or NULL: with weights given, their transposed-block stage images are (re)built into
 *   `packed` (ggnn_gru_bwd_packed_bytes(D, nx) bytes) first; NULL: `packed` holds them already.  g == NULL: pack only.
 *   dx: HOST array of nx device pointers [V,D]. */
int ggnn_gru_bwd_is_fused(int D);
size_t ggnn_gru_bwd_packed_bytes(int D, int nx);
int ggnn_gru_bwd_fused_f32(const float* g, const float* h, const float* r, const float* u, const float* c, const float* Wg,
                           const float* Wc, float* packed, float* dpc, float* dpg, float* rh, float* dh, float* const* dx,
                           const float* nin, int T, int use_avg, int nx, int V, int D, int act, ggnn_stream_t stream);
/* ... with the incoming gradient gathered on load:  g_eff[v] = g[v] + sum of the rows of gz named by gz_heads[v] (one int4 slot-head
 * record per node as ggnn_slot_heads_i32 builds them, -1 = no such slot) -- the per-node sum that closes the transform backward of
 * the timestep processed BEFORE this one (dh[v] += sum_t Z[row(v,t)], chem_tensorflow_sparse.py:160-168 under autodiff), taken by
 * its consumer instead of by a launch of its own.  Bit for bit ggnn_gather_segment_sum_heads_f32(gz, .., accumulate = 1) into g
 * followed by ggnn_gru_bwd_fused_f32, for nodes with at most four rows (more than four slots: run the stand-alone sum).
 * `packed` holds the stage images (no Wg / Wc form). */
int ggnn_gru_bwd_fused_gather_f32(const float* g, const float* gz, const int32_t* gz_heads, const float* h, const float* r,
                                  const float* u, const float* c, float* packed, float* dpc, float* dpg, float* rh, float* dh,
                                  float* const* dx, const float* nin, int T, int use_avg, int nx, int V, int D, int act,
                                  ggnn_stream_t stream);

/* ---- optimiser: per-variable clip_by_norm + TF-1.3 Adam for ALL variables in two launches (chem_tensorflow.py:183-191) --------
 * grads / m / v are FLAT buffers of nblocks * ggnn_optim_block_floats() floats in which every variable starts on a block
 * boundary (padding zero); param_ptrs: DEVICE array [nvars] of the variables' base pointers (each 16-byte aligned; the
 * parameters stay in the model's tensors), var_numel [nvars] their element counts; block_var [nblocks]: variable of each block; var_first [nvars+1]: first block of each variable;
 * var_active [nvars]: 0 = the variable has no gradient this step (left untouched, like a None gradient); partial: nblocks floats.
 *   g <- g * clip / max(||g||_2, clip) per variable (clip_norm <= 0: no clipping);
 *   m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  p <- p - lr_t m / (sqrt(v) + eps)      (lr_t = lr sqrt(1-b2^t) / (1-b1^t), host) */
int ggnn_optim_block_floats(void);
int ggnn_clip_adam_f32(float* const* param_ptrs, const int32_t* var_numel, const float* grads, float* m, float* v, float* partial,
                       const int32_t* block_var, const int32_t* var_first, const int32_t* var_active, int nblocks, float clip_norm,
                       float lr_t, float beta1, float beta2, float epsilon, ggnn_stream_t stream);

/* ---- batch assembly from dataset-level tables (the step BEFORE the path: chem_tensorflow_sparse.py:278-350 packs a minibatch as
 * one disconnected super-graph, :120-129 derives the message index) --------------------------------------------------------------
 * Every per-batch index structure is a concatenation of per-molecule pieces plus offsets, so the general builders (sorts, scans)
 * run ONCE over the whole dataset taken as one batch and a batch is gathered from their outputs.
 *   ds_tables [11] device pointers for the whole dataset (Gd graphs, Nd nodes, Md messages, Rd compact rows):
 *     0 node_ptr i32[Gd+1]   1 annotations f32[Nd,A]   2 nin f32[Nd,T]   3 row_ptr i32[Nd+1]   4 adj i32[Md,2] (type-major)
 *     5 slot -> src*T+type i32[Md]   6 slot -> message id i32[Md]   7 slot -> compact row i32[Md] (NULL: no compaction)
 *     8 compact row -> node i32[Rd]   9 first message of graph g in the type-t list i32[Gd,T]   10 first compact row, i32[Gd,T]
 *   ds_type_off / ds_type_row_off: host [T+1], type ranges of tables 4 and 8.
 *   batch_tab (device i32): gid[G] | node_off[G+1] | slot_off[G+1] | msg_off[T][G+1] | pair_off[T][G+1]   (exclusive prefix sums over
 *     the batch's graphs, in batch order);  type_off / type_row_off: host [T+1] of the batch.
 *   out [10]: 0 h0 f32[V,D] (annotation, zero-padded: :300-302)  1 graph_nodes_list i32[V]  2 graph_ptr i32[G+1]  3 nin f32[V,T]
 *     4 adj i32[M,2]  5 row_ptr i32[V+1]  6 gather_row i32[M]  7 msg_perm i32[M]  8 pair_node i32[R]  9 compact gather rows i32[M]
 *     10 slot heads i32[V,4] (may be NULL): what ggnn_build_slot_heads gives for (5, 9) -- for (5, 6) without compaction
 *   -- exactly what ggnn_build_target_csr + ggnn_build_compact_sources + ggnn_remap_gather_rows produce for the batch.
 *   ONE launch (the gathers are independent; block ranges of one grid work on nodes / state rows / messages / slots / compact rows):
 *   next to a running forward pass every dependent launch of the packer waits for a launch boundary of the forward stream. */
/* ... and the transpose structures of the backward pass for the same batch (graph_nodes_list = out[1] of ggnn_assemble_batch; batch_tab
 * carries one more array at its end: ptot[G+1], compact rows of the graphs before k over all types).
 *   bwd_tables [8]: 0 by-(src*T+type) row_ptr i32[Nd*T+1]  1 slot -> dst  2 slot -> message id  3 compact row -> first message slot
 *     i32[Rd+1]  4 slot -> dst  5 slot -> message id (type-major slots)  6 node -> first of its compact rows i32[Nd+1]  7 those rows i32[Rd]
 *   out [8]: the same eight arrays for the batch (V*T+1, M, M, R+1, M, M, V+1, R elements). */
int ggnn_assemble_batch_backward(const void* const* ds_tables, const void* const* bwd_tables, int A, int T, const int64_t* ds_type_off,
                                 const int64_t* ds_type_row_off, const int32_t* batch_tab, const int32_t* graph_nodes_list, int G, int V,
                                 int M, int R, int D, const int64_t* type_off, const int64_t* type_row_off, void* const* out,
                                 ggnn_stream_t stream);
int ggnn_assemble_batch(const void* const* ds_tables, int A, int T, const int64_t* ds_type_off, const int64_t* ds_type_row_off,
                        const int32_t* batch_tab, int G, int V, int M, int R, int D, const int64_t* type_off,
                        const int64_t* type_row_off, void* const* out, ggnn_stream_t stream);
/* batch_tab of the two calls above, formed on the device from the batch's graph ids (no host->device copy per batch: it would make
 * the host wait for everything queued on the stream), together with the batch's labels -- one launch:
 *   counts_t i32[rows][Gd]: per dataset graph the node count, message count, messages per type, compact rows per type, compact rows
 *     (rows = 2 + 2T + 1);  gids i64[G] dataset graph ids in batch order;
 *   batch_tab out i32[G + rows*(G+1)] = gid | exclusive prefix sums of every counts row over the batch's graphs;
 *   targets f32[Gd, num_targets], label_mask f32[Gd, num_targets] or NULL (all ones), task_ids i64[K] (device):
 *   target_values[i][j] = targets[gid[j]][task_ids[i]] * mask, target_mask[i][j] = mask   (chem_tensorflow_sparse.py:319-321, 335). */
int ggnn_pack_batch_tables(const int32_t* counts_t, int Gd, int rows, const int64_t* gids, int G, const float* targets,
                           const float* label_mask, int num_targets, const int64_t* task_ids, int K, int32_t* batch_tab,
                           float* target_values, float* target_mask, ggnn_stream_t stream);

/* ---- the optimisation step of the default sparse model as native launch sequences (chem_tensorflow.py:183-191 over
 * chem_tensorflow_sparse.py:117-218; ggnn_train.hip) -----------------------------------------------------------------------------
 * Default cell (GRU), no edge bias, no attention, a hidden size with the gather-fused GRU and the compacted transform (32, 64, 100),
 * at most 2 residual inputs and at least one timestep per layer.  Layer description as in ggnn_sparse_propagate_f32.
 *   forward:  per timestep the compacted transform + ggnn_gru_packed_gather_train_f32; every timestep's state and r / u / c /
 *     incoming stay in `ws` (ggnn_sparse_train_workspace_bytes, 256-byte aligned) for the backward call on the SAME ws.
 *     edge_packed[l]: stage images of the (weight-dropout-masked) edge weights (ggnn_edge_weights_pack_f32), gru_packed[l]:
 *     ggnn_gru_pack_weights_f32.  *final_state_offset: byte offset inside ws of the final node states [V,D].
 *   backward: d_final [V,D] = dL/d(final states) (from ggnn_readout_loss_bwd_f32).  Per timestep, last to first: the fused GRU
 *     backward, the transpose segment sum over (rows_rp, rows_gather[, rows_heads]) (segments = compact rows, gather = target node),
 *     the compacted transform with the images of W^T (edge_packed_t[l]) on identity_rows = 0..R-1, the per-node sum over
 *     (node_rp, node_order[, node_heads]); on `side_stream` the weight-gradient products, ADDED into g_edge[l] [T,D,D], g_Wg[l]
 *     [(nx+1)D, 2D], g_bg[l] [2D], g_Wc[l] [(nx+1)D, D], g_bc[l] [D] (the caller zeroes them; under weight dropout it masks g_edge
 *     afterwards).  d_state_ws[l] [V,D] (l < num_layers): scratch for the gradients of the layer inputs; their contents are UNDEFINED
 *     on return (d_state_ws[0], the gradient of h0, is not even completed: h0 is data).  Returns with `stream`
 *     ordered behind the side stream's last product.  Events come from a per-device pool the library creates on first use.
 *   prepare: ALL of a step's stage images in one launch (the weights change every step): per layer l the T edge-weight images
 *     (edge_packed[l], ggnn_msg_transform_compact_workspace_bytes) and those of the transposed weights (edge_packed_t[l]) -- of the
 *     variable edge_w[l] [T*D, D] masked on the fly like ggnn_dropout_f32(keep_prob, seeds[l]) masks it (keep_prob 1: unmasked) --,
 *     the fused GRU's images (gru_packed[l], ggnn_gru_packed_bytes(D, nx[l])) and its backward's (gru_bwd_packed[l]).  num_layers <= 16;
 *     nx, seeds, and the pointer arrays are HOST arrays.
 *   gru_fmt (prepare and forward; HOST [num_layers] of GGNN_GRU_FMT_*, or NULL = BF16X3 for every layer): the operand format of
 *     layer l's GRU forward images / launches -- the same array must be given to both calls of a step. */
int ggnn_sparse_train_prepare_f32(int num_layers, int T, int D, const int32_t* nx, const float* const* edge_w, float keep_prob,
                                  const uint64_t* seeds, const float* const* Wg, const float* const* Wc, const int32_t* gru_fmt,
                                  float* const* edge_packed, float* const* edge_packed_t, float* const* gru_packed,
                                  float* const* gru_bwd_packed, ggnn_stream_t stream);
size_t ggnn_sparse_train_workspace_bytes(int V, int D, int T, int64_t compact_rows, int total_steps);
int ggnn_sparse_train_forward_f32(const float* h0, int V, int D, int T, const int32_t* row_ptr, const int32_t* gather_row_c,
                                  const int32_t* pair_node, const int64_t* type_row_off, const float* nin, int use_avg, int num_layers,
                                  const int32_t* layer_timesteps, const int32_t* res_ptr, const int32_t* res_idx,
                                  const float* const* edge_packed, const float* const* bg, const float* const* bc,
                                  const float* const* gru_packed, const int32_t* gru_fmt, int act, void* ws, size_t ws_bytes,
                                  int64_t* final_state_offset, ggnn_stream_t stream);
int ggnn_sparse_train_backward_f32(const float* h0, int V, int D, int T, const int32_t* pair_node, const int64_t* type_row_off,
                                   const float* nin, int use_avg, int num_layers, const int32_t* layer_timesteps,
                                   const int32_t* res_ptr, const int32_t* res_idx, const int32_t* rows_rp, const int32_t* rows_gather,
                                   const int32_t* rows_heads, const int32_t* node_rp, const int32_t* node_order,
                                   const int32_t* node_heads, const int32_t* identity_rows, const float* const* edge_packed_t,
                                   const float* const* gru_bwd_packed, int act, float* const* g_edge, float* const* g_Wg,
                                   float* const* g_bg, float* const* g_Wc, float* const* g_bc, float* d_final, float* const* d_state_ws,
                                   void* ws, size_t ws_bytes, ggnn_stream_t stream, ggnn_stream_t side_stream);

/* ---- tf.nn.dropout with a counter-based mask (chem_tensorflow_sparse.py:91 edge-weight dropout, :113-114 DropoutWrapper on the
 * new node state; chem_tensorflow_dense.py:104; utils.py:68 readout weights) ------------------------------------------------------
 *   out[r,c] = x[r,c] / keep_prob * floor(keep_prob + U),   U = (Philox4x32-10(counter, key)[c % 4] >> 8) * 2^-24
 *   key = (seed lo, seed hi), counter = (rowkey lo, rowkey hi, c / 4, 0), rowkey = row_key[r] (row_key != NULL) or row_key_base + r.
 * The mask is a pure function of (seed, row key, column): ranks of a data-parallel job derive identical weight masks from the
 * same seed, a node's state mask does not depend on the batch it sits in, and the backward pass is the same call on dL/dout.
 * x, out [rows, cols] contiguous fp32 (in place allowed); any cols (16-byte accesses when cols % 4 == 0). */
int ggnn_dropout_f32(const float* x, float* out, const int64_t* row_key, int64_t row_key_base, uint64_t seed, float keep_prob,
                     int64_t rows, int cols, ggnn_stream_t stream);

/* ---- measurement aid (bench.py's roofline leg; not on the product path) --------------------------------------------------------
 * The dense bf16 MFMA rate the chip SUSTAINS: `launches` back-to-back launches (~16 ms each) of v_mfma_f32_16x16x32_bf16 on
 * register operands, 8 waves per CU; mode 0 all-zero operands, 1 random operands, 2 the operand pattern and planes of the 3-way
 * split product (csrc/ggnn_split.hpp).  Reports the last launch: whole-chip dense TFLOP/s and the mean shader clock inside it
 * (the data-sheet peak, 2.5 PF, is 256 CUs x 4 x 8192 flops / 16 clocks at 2.4 GHz; under random operands the chip runs at its
 * socket power limit well below that clock).  ws: ggnn_probe_mfma_workspace_bytes() of device memory.  Synchronises `stream`. */
size_t ggnn_probe_mfma_workspace_bytes(void);
int ggnn_probe_mfma_rate(int mode, int launches, void* ws, size_t ws_bytes, double* tflops, double* shader_mhz, ggnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GGNN_HIP_H */
