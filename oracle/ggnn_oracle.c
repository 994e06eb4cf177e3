/* CPU ORACLE #3 (test infrastructure, NOT product code): scalar C restatement, fp32, of one sparse
 * GGNN propagation timestep in the reference's op order.  Parity status: checked against ggnn_oracle.py, which is
 * held to the reference-run vectors; TensorFlow's kernels themselves unpinned (see ggnn_oracle.py).
 *
 * Follows /root/reference/chem_tensorflow_sparse.py:153-216 (attention off) with TF-1.3 op
 * semantics: embedding_lookup = row gather, matmul = k-ordered dot products,
 * unsorted_segment_sum = zero-filled output + serial accumulation in message order (type
 * ascending, then list order), GRUCell = r,u gates then candidate (r columns first, gate bias
 * supplied by the caller).
 *
 * Built by oracle/Makefile into oracle/_build/libggnn_oracle.so; loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg (single thread: "cores": 1).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* y[n] = sum_k x[k] * W[k*ldw + n], k ascending (row-vector times row-major matrix) */
static void vecmat(const float* x, const float* W, int K, int N, int ldw, float* y) {
    for (int n = 0; n < N; ++n) y[n] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float xv = x[k];
        const float* w = W + (size_t)k * ldw;
        for (int n = 0; n < N; ++n) y[n] += xv * w[n];
    }
}

/* One timestep.  Returns 0, or -1 on an out-of-range index (TF-CPU raises InvalidArgument).
 *   h        [V,D]       current states
 *   adj      concatenation over types of (src,dst) int32 pairs; type t owns pairs
 *            type_off[t] .. type_off[t+1]-1                      (sparse.py:67,159-160)
 *   nin      [V,T]       incoming-edge counts                    (sparse.py:69)
 *   W        [T,D,D]     edge weights                            (sparse.py:88-92)
 *   bias     [T,D] or NULL                                       (sparse.py:98-100,202-204)
 *   res      R pointers to [V,D] residual states, placed BEFORE the aggregate (sparse.py:211-212)
 *   Wg [(R+2)D, 2D], bg [2D], Wc [(R+2)D, D], bc [D]             (TF-1.3 GRUCell)
 *   act      0 = tanh, 1 = relu                                  (sparse.py:75-81)
 *   h_out    [V,D]
 */
int ggnn_oracle_sparse_step_f32(const float* h, const int* adj, const int* type_off, const float* nin,
                                const float* W, const float* bias, const float* const* res, int R,
                                const float* Wg, const float* bg, const float* Wc, const float* bc,
                                int use_avg, int act, int V, int D, int T, float* h_out) {
    const int M = type_off[T];
    float* incoming = (float*)calloc((size_t)V * D, sizeof(float));   /* segment_sum zero fill */
    float* msg = (float*)malloc(sizeof(float) * D);
    const int in_dim = (R + 1) * D;
    float* x = (float*)malloc(sizeof(float) * (in_dim + D));
    float* g = (float*)malloc(sizeof(float) * 2 * D);
    float* c = (float*)malloc(sizeof(float) * D);
    int rc = 0;
    (void)M;
    for (int t = 0; t < T && rc == 0; ++t) {                          /* sparse.py:159 */
        for (int m = type_off[t]; m < type_off[t + 1]; ++m) {
            const int src = adj[2 * m], dst = adj[2 * m + 1];
            if (src < 0 || src >= V || dst < 0 || dst >= V) { rc = -1; break; }
            vecmat(h + (size_t)src * D, W + (size_t)t * D * D, D, D, D, msg);   /* :161-164 */
            float* o = incoming + (size_t)dst * D;                               /* :198-200 */
            for (int d = 0; d < D; ++d) o[d] += msg[d];
        }
    }
    for (int v = 0; v < V && rc == 0; ++v) {
        float* inc = incoming + (size_t)v * D;
        if (bias) {                                                   /* :202-204 */
            for (int d = 0; d < D; ++d) {
                float s = 0.0f;
                for (int t = 0; t < T; ++t) s += nin[(size_t)v * T + t] * bias[(size_t)t * D + d];
                inc[d] += s;
            }
        }
        if (use_avg) {                                                /* :206-209 */
            float deg = 0.0f;
            for (int t = 0; t < T; ++t) deg += nin[(size_t)v * T + t];
            const float den = deg + 1e-7f;
            for (int d = 0; d < D; ++d) inc[d] = inc[d] / den;
        }
        for (int r = 0; r < R; ++r) memcpy(x + (size_t)r * D, res[r] + (size_t)v * D, sizeof(float) * D);
        memcpy(x + (size_t)R * D, inc, sizeof(float) * D);            /* :211-212 */
        const float* hv = h + (size_t)v * D;
        memcpy(x + in_dim, hv, sizeof(float) * D);                    /* [x, h] */
        vecmat(x, Wg, in_dim + D, 2 * D, 2 * D, g);
        for (int n = 0; n < 2 * D; ++n) g[n] = sigmoidf_(g[n] + bg[n]);
        for (int d = 0; d < D; ++d) x[in_dim + d] = g[d] * hv[d];     /* [x, r*h] */
        vecmat(x, Wc, in_dim + D, D, D, c);
        for (int d = 0; d < D; ++d) {
            float cv = c[d] + bc[d];
            cv = act == 0 ? tanhf(cv) : (cv > 0.0f ? cv : 0.0f);
            const float u = g[D + d];
            h_out[(size_t)v * D + d] = u * hv[d] + (1.0f - u) * cv;
        }
    }
    free(incoming); free(msg); free(x); free(g); free(c);
    return rc;
}

int ggnn_oracle_abi_version(void) { return 1; }
