"""Stand-alone timing of the fused GRU launch (ungathered operands) at one V:  python tools/gru_launch_bench.py [V] [nx]
(GRU_FMT=2|3: operand format of the launch, default 2 = two f16 pieces; the operands here are inside its range)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 99990
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 1
D, dev = 100, "cuda:0"
torch.manual_seed(0)
xs = [torch.rand(V, D, device=dev) * 2 - 1 for _ in range(nx)]
h = torch.rand(V, D, device=dev) * 2 - 1
Wg = (torch.rand((nx + 1) * D, 2 * D, device=dev) - 0.5) * 0.3; bg = torch.ones(2 * D, device=dev)
Wc = (torch.rand((nx + 1) * D, D, device=dev) - 0.5) * 0.3; bc = torch.zeros(D, device=dev)
FMT = int(os.environ.get("GRU_FMT", "2"))
packed = pkg.ops.PackedWeights().gru(Wg, Wc, nx, D, FMT)
out = torch.empty_like(h)
run = lambda: pkg.ops.gru_packed(xs, h, packed, bg, bc, out=out, fmt=FMT)
import numpy as np
for _ in range(10): run()
torch.cuda.synchronize()
with pkg.ops.kernel_timing() as kt:
    for _ in range(50): run()
res = kt.results()
print("V=%d nx=%d: %s us per launch, checksum %.6f" % (V, nx, {k: round(float(np.median(v)) * 1e3, 1) for k, v in res.items()}, float(out.double().sum())))
