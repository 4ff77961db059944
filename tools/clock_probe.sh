#!/bin/bash
# effective clock of the MFMA kernels: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (rocprofv3 --pmc pass; durations from the same pass)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/clock; mkdir -p $OUT
for T in "$@"; do
  if [ $T = base ]; then unset VL_LIB_PATH; else export VL_LIB_PATH=$GRAFT_REPO_ROOT/vectorlite_amd/libvl_$T.so; fi
  rm -rf $OUT/$T
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/$T -o run -- python3 tools/bench_mfma.py --config c5 --rows 4000000 --nq 1024 --reps 2 --check 1 > $OUT/$T.json 2> $OUT/$T.err
  python3 - <<PY
import sqlite3, collections
c = sqlite3.connect("$OUT/$T/run_results.db")
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection where kernel_name like '%k_mfma_%' order by dispatch_id").fetchall()
d = collections.defaultdict(dict)
for disp, name, cn, val, dur in rows:
    d[disp][cn] = d[disp].get(cn, 0) + val; d[disp]['dur'] = dur; d[disp]['mode'] = 'M1' if 'ELi1ELi' in name.split('k_mfma_')[1][:22] else 'M0'
last = [v for k, v in sorted(d.items())][-4:]
for v in last:
    ghz = v.get('GRBM_GUI_ACTIVE', 0) / 8 / v['dur']
    busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (v.get('GRBM_GUI_ACTIVE', 1) / 8)
    print("$T", v['mode'], "dur_us", round(v['dur'] / 1e3), "clock_GHz", round(ghz, 3), "mfma_busy_frac_of_gui_cycles", round(busy, 3))
PY
done
