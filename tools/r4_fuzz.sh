#!/bin/bash
# round 4: the randomised campaigns + the soak on the build of the day (multi-part handles in the fuzz rotation).
# Outputs: gpurun_out/r4fuzz/.  usage: tools/r3_fuzz.sh [fuzz_s] [splitk_s] [hnsw_s] [soak_s]
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4fuzz; mkdir -p $OUT
timeout -k 10 $((${1:-200} + 120)) python3 tools/fuzz_campaign.py ${1:-200} 4026 2>&1 | tee $OUT/fuzz_campaign.txt | tail -3 || exit 1
timeout -k 10 $((${2:-120} + 120)) python3 tools/fuzz_splitk.py ${2:-120} 477 2>&1 | tee $OUT/fuzz_splitk.txt | tail -2 || exit 2
timeout -k 10 $((${3:-120} + 120)) python3 tools/fuzz_hnsw.py ${3:-120} 45 2>&1 | tee $OUT/fuzz_hnsw.txt | tail -2 || exit 3
timeout -k 10 $((${4:-90} + 120)) python3 tools/soak.py ${4:-90} 2>&1 | tee $OUT/soak.txt | tail -3 || exit 4
