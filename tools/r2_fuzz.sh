#!/bin/bash
# regression hunt after the round-2 kernel changes (row-stationary MFMA filter, per-walk visited sets): the three
# randomised campaigns, a progress line every 30 s.  Outputs: gpurun_out/r2fuzz/.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2fuzz; mkdir -p $OUT
timeout -k 10 400 python3 tools/fuzz_campaign.py ${1:-200} 2026 2>&1 | tee $OUT/fuzz_campaign.txt | tail -4 || exit 1
timeout -k 10 300 python3 tools/fuzz_splitk.py ${2:-120} 77 2>&1 | tee $OUT/fuzz_splitk.txt | tail -3 || exit 2
timeout -k 10 300 python3 tools/fuzz_hnsw.py ${3:-120} 5 2>&1 | tee $OUT/fuzz_hnsw.txt | tail -3 || exit 3
