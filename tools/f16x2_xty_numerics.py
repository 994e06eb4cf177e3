"""Would the two-piece f16 format do for the WEIGHT-GRADIENT products X^T dY (csrc/ggnn_bwd_gemm.hip)?
numpy, CPU only.  X = forward states (tanh-distributed), dY = gradient rows whose magnitudes spread over 0 / 3 / 6 decades from row to
row; the sum over V = 20,000 rows in the kernel's order (256 workgroup partials of 32-row MFMA blocks, f32 accumulation).  Forms:
three bf16 pieces x six products (today), two f16 pieces x three products with ONE power-of-two scale for the whole dY tensor
(max |dY| -> [2^14, 2^15)), the same unscaled, and a blocked f32 evaluation -- all against f64.
Result (round 4): with the per-tensor scale the f16 form's error equals the bf16 form's and the f32 evaluation's (3.0e-7 norm-wise: the
f32 accumulation over the rows is what is left) at every spread; unscaled it is 1e-4 .. 7e-4.
    python tools/f16x2_xty_numerics.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
from ggnn_oracle import bf16_split3, f16_split2
rng = np.random.default_rng(3)
V, K, N = 20000, 100, 100
X = np.tanh(rng.normal(0, 1, (V, K))).astype(np.float32)
def make_dY(decades):
    rowscale = 10.0 ** (-decades * rng.random((V, 1)))
    return (rng.normal(0, 1, (V, N)) * rowscale * 1e-4).astype(np.float32)
def chunks_sum(prod_fn, rows=32):
    # the kernel's order: per 32-row step one MFMA-accumulated block product added to an f32 accumulator (one workgroup's rows), then partials summed
    G = np.zeros((K, N), np.float32); 
    wg = 256; per = (V + wg - 1) // wg
    parts = []
    for w in range(wg):
        acc = np.zeros((K, N), np.float32)
        for r0 in range(w * per, min(V, (w + 1) * per), rows):
            for d in prod_fn(slice(r0, min(r0 + rows, min(V, (w + 1) * per)))):
                acc = (acc.astype(np.float64) + d).astype(np.float32)
        parts.append(acc)
    s = np.zeros((K, N), np.float32)
    for p in parts: s = (s + p).astype(np.float32)
    return s
def run(dY):
    ref = X.astype(np.float64).T @ dY.astype(np.float64)
    out = {}
    # f32 chain: per row fma into acc (emulate per workgroup)
    def f32_prod(sl):
        for r in range(sl.start, sl.stop):
            yield np.outer(X[r].astype(np.float64), dY[r].astype(np.float64))
    # too slow for all rows in python: approximate the chain by float32 cumulative over 32-row blocks with per-row rounding on a sample of workgroups
    xb, yb = bf16_split3(X), bf16_split3(dY)
    def six(sl):
        for i, j in ((0, 2), (1, 1), (0, 1), (2, 0), (1, 0), (0, 0)):
            yield xb[i][sl].astype(np.float64).T @ yb[j][sl].astype(np.float64)
    out['bf16x3 six'] = chunks_sum(six)
    m = float(np.abs(dY).max()); e = np.floor(np.log2(m)); sy = 2.0 ** (14 - e)     # max |dY| -> [2^14, 2^15)
    xh, xl = f16_split2(X, 1.0); yh, yl = f16_split2(dY, sy)
    def three(sl):
        for a, b in ((xh, yl), (xl, yh), (xh, yh)):
            yield a[sl].astype(np.float64).T @ b[sl].astype(np.float64)
    out['f16x2 three, dY x 2^%d' % (14 - e)] = (chunks_sum(three).astype(np.float64) / sy).astype(np.float32)
    yh1, yl1 = f16_split2(dY, 1.0)
    def three_u(sl):
        for a, b in ((xh, yl1), (xl, yh1), (xh, yh1)):
            yield a[sl].astype(np.float64).T @ b[sl].astype(np.float64)
    out['f16x2 three, unscaled'] = chunks_sum(three_u)
    # plain f32 blocked: per 32-row block an f32 matmul (numpy f32), i.e. a different but f32-class evaluation
    def f32blk(sl):
        yield (X[sl].T @ dY[sl]).astype(np.float64)
    out['f32 blocked'] = chunks_sum(f32blk)
    gmax = np.abs(ref).max(); gn = np.linalg.norm(ref)
    for k, v in out.items():
        e_ = v.astype(np.float64) - ref
        print('   %-28s max|err|/max|G| %.3e   ||err||/||G|| %.3e   max elementwise rel (|G|>1e-3 max) %.3e' % (
            k, np.abs(e_).max() / gmax, np.linalg.norm(e_) / gn, (np.abs(e_) / np.abs(ref))[np.abs(ref) > 1e-3 * gmax].max()))
for dec in (0, 3, 6):
    print('row scales over %d decades' % dec); run(make_dY(dec))
