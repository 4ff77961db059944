#!/bin/bash
# round 3: live threshold refinement (k_refine_live) against the staged form -- config 3's shard, a parity hunt with the
# live form on, and the small-batch shapes at N = 10 M whose grids leave CUs idle.  Outputs: gpurun_out/r3live/.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3live; mkdir -p $OUT
for L in 0 1 0 1; do
  VL_MFMA_LIVE=$L python3 tools/bench_mfma.py --config c3 --reps 5 --check 32 > $OUT/c3_live$L.json 2> $OUT/c3_live$L.err || exit 1
  python3 - <<PY
import json
d = json.load(open("$OUT/c3_live$L.json"))
print("c3 live=$L", "filter ms", d["filter_kernels_ms_per_batch"], "frac", d["roofline"]["frac"], "whole", d["ms_per_batch"], "dev", d["device_queries"]["ms_per_batch"], d["parity"][:40])
PY
done
VL_MFMA_LIVE=1 timeout -k 10 300 python3 tools/fuzz_splitk.py ${1:-120} 991 2>&1 | tee $OUT/fuzz_splitk_live.txt | tail -2 || exit 2
VL_MFMA_LIVE=1 timeout -k 10 400 python3 tools/fuzz_campaign.py ${2:-120} 4100 2>&1 | tee $OUT/fuzz_campaign_live.txt | tail -2 || exit 3
VL_MFMA_LIVE=1 python3 -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_sharded.py tests/test_gpu_device_queries.py -m gpu -x -q 2>&1 | tail -3
for L in 0 1; do
  VL_MFMA_LIVE=$L python3 tools/bench_small_batches.py 10000000 > $OUT/small_batches_live$L.txt 2> $OUT/small_batches_live$L.err || exit 4
done
paste -d'|' $OUT/small_batches_live0.txt $OUT/small_batches_live1.txt | cut -c1-230
