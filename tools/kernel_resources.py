#!/usr/bin/env python3
"""Per-kernel register / LDS / spill figures of one csrc/*.hip file, from the metadata hipcc emits for gfx950
(cross-compiles without a GPU).  Usage: python tools/kernel_resources.py hnsw.hip [name-filter] [-D...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def resources(src: str, extra=()):
    from vectorlite_amd import build as vbuild
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [vbuild.hipcc(), f"--offload-arch={vbuild.ARCH}"] + vbuild.COMMON + list(extra) + [
            "--cuda-device-only", "-S", os.path.join(vbuild.CSRC, src), "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        asm = open(out).read()
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                         r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
                         r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", asm, re.S):
        agpr, lds, name, scratch, sgpr, sspill, vgpr, vspill = m.groups()
        rows.append(dict(name=name, vgpr=int(vgpr), agpr=int(agpr), sgpr=int(sgpr), lds=int(lds), scratch=int(scratch),
                         vspill=int(vspill), sspill=int(sspill)))
    return rows, asm


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except Exception:
        return names


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    src = args[0]
    filt = args[1] if len(args) > 1 else ""
    rows, _ = resources(src, extra)
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        n = re.sub(r"vl::\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n)
        if filt and filt not in n:
            continue
        print(f"{n:60s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} scratch {r['scratch']:5d} "
              f"spill v{r['vspill']} s{r['sspill']}")
