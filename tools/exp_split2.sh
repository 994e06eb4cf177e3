#!/bin/bash
# HISTORICAL (commit "Experiment: f32 products as TWO f16 pieces x three MFMAs"): the first measurement of the two-piece f16 split, then a
# per-translation-unit build switch (-DGGNN_SPLIT2=1), on the fused GRU and the compacted transform; its logs are
# profiles/r04_experiments/split2_first_run/.  The format has since become a template parameter (SplitFmt<FMT>, csrc/ggnn_split.hpp) and
# the fused GRU forward's default: tools/exp_fmt.sh / tools/exp_panel.sh are the runs of the final form.  The build lines below no longer apply.
# Round-4 experiment: the two-piece f16 split (GGNN_SPLIT2, csrc/ggnn_split.hpp) on the fused GRU and the compacted transform --
# INFERENCE ONLY (the training step packs its images in ggnn_train.hip, which the variant libraries leave in the bf16 x 3 format).
#   tools/variant_lib.sh s2  ggnn_gru_fused_split.hip,ggnn_msg_compact.hip -DGGNN_SPLIT2=1
#   tools/variant_lib.sh s2u ggnn_gru_fused_split.hip,ggnn_msg_compact.hip -DGGNN_SPLIT2=1 -DGGNN_SPLIT2_SCALED=0
#   hipcc --offload-arch=gfx950 -O2 tools/f16_mfma_denorm_probe.hip -o tools/_bin/f16_mfma_denorm_probe
#   gpurun -- 'bash tools/exp_split2.sh'      -> gpurun_out/split2/
OUT=gpurun_out/${1:-split2}; mkdir -p $OUT; export TMPDIR=/tmp
date +%s > $OUT/t0
timeout 60 tools/_bin/f16_mfma_denorm_probe > $OUT/denorm.txt 2>&1
for v in "" s2; do
    echo "== variant=${v:-default}" >> $OUT/fwd.txt;   GGNN_LIB_VARIANT=$v timeout 300 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1
    echo "== variant=${v:-default}" >> $OUT/probe.txt; GGNN_LIB_VARIANT=$v timeout 300 python tools/split_probe.py 2>&1 | tail -3 >> $OUT/probe.txt
done
GGNN_LIB_VARIANT=s2 timeout 300 python bench.py --no-secondary > $OUT/bench_s2.json 2> $OUT/bench_s2.err
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# the inference parity tests on the variant, tolerances unchanged
GGNN_LIB_VARIANT=s2 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x --deselect tests/test_gpu_parity.py::test_library_loaded \
    -k "test_gru or test_msg_transform or test_sparse_model_matches_oracle or test_random_model_shapes_match_oracle or test_golden_fixture_on_gpu or test_full_size_batch_properties or test_compact_transform" \
    > $OUT/pytest_s2.txt 2>&1
for v in s2u; do
    echo "== variant=$v" >> $OUT/fwd.txt;   GGNN_LIB_VARIANT=$v timeout 300 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1
    echo "== variant=$v" >> $OUT/probe.txt; GGNN_LIB_VARIANT=$v timeout 300 python tools/split_probe.py 2>&1 | tail -3 >> $OUT/probe.txt
done
echo "== GGNN_MATRIX=f32" >> $OUT/probe.txt; GGNN_MATRIX=f32 timeout 300 python tools/split_probe.py 2>&1 | tail -3 >> $OUT/probe.txt
date +%s > $OUT/t1
# the default library's GPU suite, with what is left of the call
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.txt 2>&1
date +%s > $OUT/t2
tail -3 $OUT/pytest_default.txt; cat $OUT/denorm.txt; grep -E "^==|^V =|one stream" $OUT/fwd.txt; cat $OUT/probe.txt; tail -5 $OUT/pytest_s2.txt
