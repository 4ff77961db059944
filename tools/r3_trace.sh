#!/bin/bash
# round 3: kernel-trace summaries of (a) the headline single-query run, (b) config 3's shard batch, (c) config 5
# usage: tools/r3_trace.sh <tag>     writes gpurun_out/<tag>_{headline,c3,c5}_kernel_stats.csv
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r3}
for what in headline c3 c5; do
  OUT=gpurun_out/${TAG}_trace_$what; rm -rf $OUT; mkdir -p $OUT
  case $what in
    headline) rocprofv3 --kernel-trace -d $OUT -o run -- python3 bench.py --inline --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-checks > $OUT/bench.json 2> $OUT/bench.err ;;
    c3) rocprofv3 --kernel-trace -d $OUT -o run -- python3 tools/bench_mfma.py --config c3 --reps 3 --check 4 > $OUT/bench.json 2> $OUT/bench.err ;;
    c5) rocprofv3 --kernel-trace -d $OUT -o run -- python3 tools/bench_mfma.py --config c5 --reps 3 --check 4 > $OUT/bench.json 2> $OUT/bench.err ;;
  esac
  echo "== $what rc=$?"
  python3 tools/rocpd_summary.py stats $OUT/run_results.db > gpurun_out/${TAG}_${what}_kernel_stats.csv
  cp $OUT/bench.json gpurun_out/${TAG}_${what}_bench_under_rocprofv3.json
  head -14 gpurun_out/${TAG}_${what}_kernel_stats.csv | cut -c1-140
done
