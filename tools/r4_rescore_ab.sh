#!/bin/bash
# k_batch_rescore with 4 (shipped) or 8 (libvl_nch8.so) column tiles in flight: rocprofv3 kernel stats of config 3's batch
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4ab
mkdir -p $OUT
for T in base nch8 base nch8; do
  if [ $T = base ]; then unset VL_LIB_PATH; else export VL_LIB_PATH=$GRAFT_REPO_ROOT/vectorlite_amd/libvl_$T.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/$T -o run -- python3 tools/pmc_filter_target.py --config c3 --batches 6 > $OUT/$T.json 2> $OUT/$T.err || exit 3
  python3 tools/rocpd_summary.py stats $OUT/$T/run_results.db | grep -E "k_batch|k_select" | sed "s/^/$T /" | cut -c1-120
  rm -rf $OUT/$T
done
