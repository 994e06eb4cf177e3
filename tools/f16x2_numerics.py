"""Numerics of the two-piece f16 split (csrc/ggnn_split.hpp, kSplitF16x2) in numpy, CPU only -- the study behind DESIGN.md K0 "round 4":
GRU-shaped products A [512, K] x W [K, 100] evaluated as (1) an f32 FMA chain, (2) three bf16 pieces x six products, (3) two f16
pieces x three products -- unscaled, with f16 subnormals flushed (what the MFMA does NOT do: tools/f16_mfma_denorm_probe.hip), and
with power-of-two scales on the operands -- against f64; then the same over pairs of scales (which operand needs the scale).
    python tools/f16x2_numerics.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
from ggnn_oracle import bf16_split3, split6_matmul
rng = np.random.default_rng(0)

def f16_split2(a, scale=1.0, ftz=False):
    a = (np.asarray(a, np.float32) * np.float32(scale)).astype(np.float32)
    def cvt(x):
        h = x.astype(np.float16)
        if ftz:
            h = np.where(np.abs(h.astype(np.float32)) < 2.0**-14, np.float16(0), h)
        return h
    hi = cvt(a)
    r = (a - hi.astype(np.float32)).astype(np.float32)
    lo = cvt(r)
    return hi.astype(np.float32), lo.astype(np.float32)

def split3_f16_matmul(A, W, chunk=32, sa=1.0, sw=1.0, ftz=False):
    a = f16_split2(A, sa, ftz); w = f16_split2(W, sw, ftz)
    order = [(0, 1), (1, 0), (0, 0)]     # a_hi w_lo | a_lo w_hi | a_hi w_hi
    acc = np.zeros((A.shape[0], W.shape[1]), np.float32)
    for c in range(0, A.shape[1], chunk):
        for i, j in order:
            d = a[i][:, c:c+chunk].astype(np.float64) @ w[j][c:c+chunk].astype(np.float64)
            acc = (acc.astype(np.float64) + d).astype(np.float32)
    return (acc.astype(np.float64) / (sa * sw)).astype(np.float32)

def f32_chain(A, W):
    acc = np.zeros((A.shape[0], W.shape[1]), np.float32)
    for k in range(A.shape[1]):
        # fma: exact product + one rounding
        acc = (acc.astype(np.float64) + A[:, k:k+1].astype(np.float64) * W[k:k+1].astype(np.float64)).astype(np.float32)
    return acc

def report(name, A, W):
    ref = A.astype(np.float64) @ W.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64)
    out = {}
    out['f32 chain'] = f32_chain(A, W)
    out['bf16x3 six'] = split6_matmul(A, W)
    out['f16x2 three'] = split3_f16_matmul(A, W)
    out['f16x2 three ftz'] = split3_f16_matmul(A, W, ftz=True)
    out['f16x2 three s=2^8,2^8'] = split3_f16_matmul(A, W, sa=256., sw=256.)
    out['f16x2 three s=2^8,2^8 ftz'] = split3_f16_matmul(A, W, sa=256., sw=256., ftz=True)
    print(f"== {name}: M={A.shape[0]} K={A.shape[1]} N={W.shape[1]}")
    for k, v in out.items():
        e = np.abs(v.astype(np.float64) - ref)
        print(f"  {k:28s} max|err| {e.max():.3e}  rms {np.sqrt((e**2).mean()):.3e}  max err/sum|a||w| {(e/den).max():.3e}  rms {np.sqrt(((e/den)**2).mean()):.3e}")

for K in (100, 200, 300):
    lim = np.sqrt(6.0 / (K + 100))
    W = rng.uniform(-lim, lim, (K, 100)).astype(np.float32)
    A = np.tanh(rng.normal(0, 1, (512, K))).astype(np.float32)
    report(f"tanh states x glorot K={K}", A, W)
W = rng.uniform(-0.17, 0.17, (200, 100)).astype(np.float32)
A = (rng.normal(0, 1, (512, 200)) * 10 ** rng.uniform(-4, 0, (512, 200))).astype(np.float32)
report("wide-range activations", A, W)
A = rng.uniform(-1, 1, (512, 200)).astype(np.float32); W = rng.uniform(-1, 1, (200, 100)).astype(np.float32)
report("U(-1,1) x U(-1,1)", A, W)

print("==== scale pairs")
def rms_err(fn, A, W):
    ref = A.astype(np.float64) @ W.astype(np.float64)
    return np.sqrt(((fn(A, W).astype(np.float64) - ref) ** 2).mean())
for K in (100, 300):
    lim = np.sqrt(6.0 / (K + 100))
    W = rng.uniform(-lim, lim, (K, 100)).astype(np.float32)
    A = np.tanh(rng.normal(0, 1, (512, K))).astype(np.float32)
    print(K, 'bf16x3', rms_err(split6_matmul, A, W), 'f32chain', rms_err(f32_chain, A, W))
    for sa, sw in ((1,1),(1,256),(16,256),(256,256),(16,4096),(4,256),(64,256)):
        print('   sa=%d sw=%d' % (sa, sw), rms_err(lambda a, w: split3_f16_matmul(a, w, sa=float(sa), sw=float(sw)), A, W))
