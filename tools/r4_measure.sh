#!/bin/bash
# Round-4 evidence pass on the GPU box.  The program itself follows `--` (no env/bash hop under the profiler);
# counters are collected in their own passes.  Outputs: gpurun_out/r4m/ (summaries as CSV next to the raw output).
#   headline: kernel-trace stats + FETCH_SIZE of the driver's command (--inline form)
#   c3 / c5 : kernel-trace stats + MFMA busy + FETCH_SIZE of tools/bench_mfma.py
#   hnsw    : kernel-trace stats of tools/hnsw_eval.py (build + walks, 1 M x 384, latent-16)
# usage: tools/r3_measure.sh [headline] [c3] [c5] [hnsw]      (default: all)
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4m
mkdir -p $OUT
WHAT=${@:-headline c3 c5 hnsw}
for w in $WHAT; do
  case $w in
    headline)
      rocprofv3 --kernel-trace --stats -d $OUT/headline_stats -o run -- python3 bench.py --inline --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-checks > $OUT/headline_bench_under_rocprofv3.json 2> $OUT/headline_stats.err || exit 3
      python3 tools/rocpd_summary.py stats $OUT/headline_stats/run_results.db > $OUT/headline_rocprofv3_kernel_stats.csv
      rocprofv3 --pmc FETCH_SIZE -d $OUT/headline_pmc -o run -- python3 bench.py --inline --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-checks > $OUT/headline_bench_under_pmc.json 2> $OUT/headline_pmc.err || exit 4
      python3 tools/rocpd_summary.py pmc $OUT/headline_pmc/run_results.db > $OUT/headline_rocprofv3_pmc_fetch_size.csv
      head -6 $OUT/headline_rocprofv3_kernel_stats.csv | cut -c1-150; head -6 $OUT/headline_rocprofv3_pmc_fetch_size.csv | cut -c1-150
      ;;
    c3|c5)
      python3 tools/bench_mfma.py --config $w --reps 5 > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err || exit 5
      rocprofv3 --kernel-trace --stats -d $OUT/${w}_stats -o run -- python3 tools/bench_mfma.py --config $w --reps 3 --check 4 > $OUT/${w}_prof.json 2> $OUT/${w}_prof.err || exit 6
      python3 tools/rocpd_summary.py stats $OUT/${w}_stats/run_results.db > $OUT/${w}_rocprofv3_kernel_stats.csv
      rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/${w}_pmc1 -o run -- python3 tools/bench_mfma.py --config $w --reps 2 --check 4 > $OUT/${w}_pmc1.json 2> $OUT/${w}_pmc1.err || exit 7
      python3 tools/rocpd_summary.py pmc $OUT/${w}_pmc1/run_results.db > $OUT/${w}_rocprofv3_pmc_mfma_busy.csv
      rocprofv3 --pmc FETCH_SIZE -d $OUT/${w}_pmc2 -o run -- python3 tools/bench_mfma.py --config $w --reps 2 --check 4 > $OUT/${w}_pmc2.json 2> $OUT/${w}_pmc2.err || exit 8
      python3 tools/rocpd_summary.py pmc $OUT/${w}_pmc2/run_results.db > $OUT/${w}_rocprofv3_pmc_fetch_size.csv
      head -8 $OUT/${w}_rocprofv3_kernel_stats.csv | cut -c1-150
      ;;
    hnsw)
      rocprofv3 --kernel-trace --stats -d $OUT/hnsw_stats -o run -- python3 tools/hnsw_eval.py --rows 1000000 --dim 384 --latent 16 --efs 10,32,128 > $OUT/hnsw_eval_under_rocprofv3.txt 2> $OUT/hnsw_stats.err || exit 9
      python3 tools/rocpd_summary.py stats $OUT/hnsw_stats/run_results.db > $OUT/hnsw_rocprofv3_kernel_stats.csv
      head -12 $OUT/hnsw_rocprofv3_kernel_stats.csv | cut -c1-150
      ;;
  esac
  echo "[r4m] $w done"
done
# the raw databases stay on the box (large); only the summaries travel back
rm -rf $OUT/*_stats $OUT/*_pmc $OUT/*_pmc1 $OUT/*_pmc2
