#!/bin/bash
# round 3, final build: longer campaigns from fresh seeds.  Outputs: gpurun_out/r3fuzz_long/.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3fuzz_long; mkdir -p $OUT
timeout -k 10 520 python3 tools/fuzz_campaign.py 400 777000 2>&1 | tee $OUT/fuzz_campaign.txt | tail -2 || exit 1
timeout -k 10 320 python3 tools/fuzz_splitk.py 200 5151 2>&1 | tee $OUT/fuzz_splitk.txt | tail -2 || exit 2
timeout -k 10 320 python3 tools/fuzz_hnsw.py 200 909 2>&1 | tee $OUT/fuzz_hnsw.txt | tail -2 || exit 3
