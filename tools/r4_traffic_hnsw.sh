#!/bin/bash
# FETCH_SIZE of the HNSW walk kernel for config 4's three beams, one --pmc pass per beam (no trace domains), the program itself
# behind `--`.  Outputs: gpurun_out/r4t/c4_ef<ef>_pmc_fetch_size.csv + c4_ef<ef>_target.json -> tools/make_traffic_json.py hnsw
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4t
mkdir -p $OUT
for ef in ${@:-10 32 128}; do
  rocprofv3 --pmc FETCH_SIZE -d $OUT/c4_${ef}_pmc -o run -- python3 tools/pmc_hnsw_target.py --ef $ef --batches 4 > $OUT/c4_ef${ef}_target.json 2> $OUT/c4_ef${ef}_pmc.err || exit 4
  python3 tools/rocpd_summary.py pmc $OUT/c4_${ef}_pmc/run_results.db > $OUT/c4_ef${ef}_pmc_fetch_size.csv
  cat $OUT/c4_ef${ef}_target.json; grep -E "k_hnsw_search" $OUT/c4_ef${ef}_pmc_fetch_size.csv | cut -c1-160
  rm -rf $OUT/c4_${ef}_pmc
  echo "[r4t] c4 ef $ef done"
done
