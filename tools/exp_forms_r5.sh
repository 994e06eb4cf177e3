#!/bin/bash
# Round 5: the ring forms of the gather-fused GRU launch under the two-piece f16 operand format on ONE box (GGNN_GRU_FORM: 0 whole
# images / 2 slots, 1 two 4-wave workgroups on half images, 2 half images / 3 slots / partial waits, 5 whole images / 3 slots / the DMA
# two stages ahead / barriers that leave the stage's own DMA and fetches in flight), bit-identity of their states, the tail-pass cost.
OUT=gpurun_out/${1:-forms}; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" timeout 200 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
for f in 0 5 1 2 0 5; do run GGNN_GRU_FORM=$f; done
run GGNN_GRU_FORM=0 GGNN_FWD_BATCH_NODES=98320; run GGNN_GRU_FORM=5 GGNN_FWD_BATCH_NODES=98320
grep -E "^==|^V =|one stream" $OUT/fwd.txt
# bit-identity of the forms' states + parity subset under form 5
cat > /tmp/form_hash.py <<'PY'
import importlib, os, sys, hashlib, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("gated-graph-neural-network-samples_amd")
ms = pkg.synthetic_qm9(5700, mean_nodes=18, seed=7)
model = pkg.SparseGGNNChemModel({"--quiet": True, "--device": "cuda:0", "train_data": None, "valid_data": ms})
f = list(model.make_minibatch_iterator(model.valid_data, False))[0]
g = torch.Generator(device="cpu").manual_seed(1)
f["initial_node_representation"] = (torch.rand(f["initial_node_representation"].shape, generator=g) * 2 - 1).cuda()
with torch.no_grad():
    model.feed(f); out = model.compute_final_node_representations()
print("form", os.environ.get("GGNN_GRU_FORM"), "formats", model.last_gru_formats, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest())
PY
for f in 0 5 1 2; do GGNN_GRU_FORM=$f python /tmp/form_hash.py 2>&1 | tail -1; done
GGNN_GRU_FORM=5 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_golden.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
