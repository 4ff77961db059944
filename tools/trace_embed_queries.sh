export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/embtrace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o run -- python3 tools/bench_embed_queries.py 300000 > $OUT/out.txt 2> $OUT/err.txt
tail -1 $OUT/out.txt
python3 - <<PY
import sqlite3, re, collections
c = sqlite3.connect("$OUT/run_results.db")
agg = collections.defaultdict(list)
for name, dur in c.execute("select name, duration from kernels"):
    m = re.search(r"k_\w+", name)
    agg[m.group(0) if m else name[:40]].append(dur)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k:32s} calls {len(v):5d} avg {sum(v)/len(v)/1e3:9.1f} us  min {min(v)/1e3:8.1f}")
PY
