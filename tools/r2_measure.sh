#!/bin/bash
# Round-2 measurement pass on the GPU box: the driver's command, the same under rocprofv3 (--inline: the program
# itself after `--`, no child process under the profiler), a separate PMC pass, the nccl branch at world 1,
# and config 3's shard through vl_shard_search_batch.  Outputs under gpurun_out/r2m/.
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2m
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err || exit 1
echo "[r2m] driver command done"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --rows 2000000 --no-cpu-baseline --no-checks > $OUT/bench_nccl_world1.json 2> $OUT/bench_nccl_world1.err || exit 2
echo "[r2m] nccl world-1 done"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- python3 bench.py --inline --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-checks > $OUT/bench_prof.json 2> $OUT/bench_prof.err || exit 3
echo "[r2m] kernel-trace done"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o run -- python3 bench.py --inline --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-checks > $OUT/bench_pmc.json 2> $OUT/bench_pmc.err || exit 4
echo "[r2m] pmc done"
python3 tools/bench_sharded.py --rows-per-rank 1250000 --dim 768 --batch 1024 --steps 5 > $OUT/c3_shard_bench.json 2> $OUT/c3_shard_bench.err || exit 5
echo "[r2m] c3 shard done"
ls -R $OUT | head -50
