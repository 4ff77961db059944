#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this turns one into the two small CSV
summaries kept under profiles/: per-kernel time statistics (what `--stats` reports) and, for a `--pmc` pass,
per-kernel counter averages.  Kernel names are shortened to the function name + template arguments.

  python tools/rocpd_summary.py stats  gpurun_out/x/stats/run_results.db  > profiles/rNN_kernel_stats.csv
  python tools/rocpd_summary.py pmc    gpurun_out/x/pmc/run_results.db    > profiles/rNN_pmc.csv
  python tools/rocpd_summary.py stats DB --where "grid_x >= 768"           (filter dispatches, SQL on `kernels`)
"""
import re
import sqlite3
import subprocess
import sys


_dm = {}


def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    if name not in _dm:
        try:
            _dm[name] = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            _dm[name] = name
    return _dm[name]


def itanium_fallback(name: str) -> str:
    """c++filt does not know the bf16 type code (DF16b): render `11k_mfma_rowsILi24ELi1ELi0EE...` as k_mfma_rows<24, 1, 0>."""
    m = re.search(r"\d+(k_\w+?)I((?:L[ib]\d+E)+)E", name)
    if not m:
        return name
    args = [v if t == "i" else ("true" if v == "1" else "false") for t, v in re.findall(r"L([ib])(\d+)E", m.group(2))]
    return "vl::" + m.group(1) + "<" + ", ".join(args) + ">("


def short(name: str) -> str:
    if name.startswith("_Z") and demangle(name).startswith("_Z"):
        name = itanium_fallback(name)
    name = re.sub(r"\(anonymous namespace\)::", "", demangle(name))
    m = re.match(r"(?:void\s+)?([\w:]+(?:<[^()]*?>)?)\(", name)
    s = m.group(1) if m else name
    return s if len(s) <= 120 else s[:117] + "..."


def stats(db, where=None):
    c = sqlite3.connect(db)
    q = "select name, duration from kernels" + (f" where {where}" if where else "")
    agg = {}
    for name, dur in c.execute(q):
        a = agg.setdefault(short(name), [0, 0, None, None])
        a[0] += 1
        a[1] += dur
        a[2] = dur if a[2] is None else min(a[2], dur)
        a[3] = dur if a[3] is None else max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    print("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{k}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{a[2]},{a[3]},{100.0 * a[1] / total:.3f}')


def pmc(db):
    c = sqlite3.connect(db)
    agg = {}
    for name, counter, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
        a = agg.setdefault((short(name), counter), [0, 0.0])
        a[0] += 1
        a[1] += value
    print("kernel,counter,dispatches,avg_value,avg_value_x2_KB_to_bytes_gfx950")
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for (k, cn), a in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        if only and only not in k:
            continue
        avg = a[1] / a[0]
        # MI355X_MICROARCH.md, HBM/rocprofv3 section: FETCH_SIZE / WRITE_SIZE are in KB and read 2x low on gfx950
        corr = f"{avg * 1024 * 2:.0f}" if cn in ("FETCH_SIZE", "WRITE_SIZE") else ""
        print(f'"{k}",{cn},{a[0]},{avg:.3f},{corr}')


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    where = sys.argv[sys.argv.index("--where") + 1] if "--where" in sys.argv else None
    stats(db, where) if mode == "stats" else pmc(db)
