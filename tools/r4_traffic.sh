#!/bin/bash
# FETCH_SIZE of the batch filter's pass-1 kernel for configs 3 and 5, in its own --pmc pass (no trace domains), the
# program itself behind `--`.  Outputs: gpurun_out/r4t/<cfg>_pmc_fetch_size.csv + <cfg>_target.json; tools/make_traffic_json.py
# turns them into profiles/traffic.json entries.       usage: tools/r4_traffic.sh [c3] [c5]
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4t
mkdir -p $OUT
for w in ${@:-c3 c5}; do
  rocprofv3 --pmc FETCH_SIZE -d $OUT/${w}_pmc -o run -- python3 tools/pmc_filter_target.py --config $w --batches 4 > $OUT/${w}_target.json 2> $OUT/${w}_pmc.err || exit 4
  python3 tools/rocpd_summary.py pmc $OUT/${w}_pmc/run_results.db > $OUT/${w}_pmc_fetch_size.csv
  cat $OUT/${w}_target.json; grep -E "k_mfma_rows|k_ingest|k_batch_rescore|k_merge_finalize" $OUT/${w}_pmc_fetch_size.csv | cut -c1-160
  echo "[r4t] $w done"
done
rm -rf $OUT/*_pmc
