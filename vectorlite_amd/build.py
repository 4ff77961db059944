"""In-tree build of libvectorlite_amd.so (hipcc, gfx950 only).

`python -m vectorlite_amd.build` or `vectorlite_amd.build.build()`.  hipcc cross-compiles
without a GPU; the resulting .so is git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
SO = os.path.join(HERE, "libvectorlite_amd.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
# -ffp-contract=off: the exact kernels must round every multiply and every add separately, in the
# reference's order (src/lib.rs:425-572); the f32 scan spells its FMAs out with fmaf().
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}"]
SOURCES = ["kernels.hip", "hnsw.hip", "mfma_scan.hip", "shard.hip", "flat_index.cpp", "multi_index.cpp", "hnsw_index.cpp", "shard_comm.cpp",
           "vlc_loader.cpp", "c_api.cpp"]
# RCCL (the row-shard all-gather, shard_comm.cpp): the ROCm copy; in a process that has imported torch the loader
# resolves the same SONAME (librccl.so.1) to the copy torch already mapped, so there is one RCCL per process
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
LINK = [f"-L{ROCM_LIB}", "-lrccl", f"-Wl,-rpath,{ROCM_LIB}"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(INCLUDE, "vectorlite_amd.h"))
    hs.append(os.path.abspath(__file__))
    return hs


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    headers = _headers()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc, f"--offload-arch={ARCH}"] + COMMON + os.environ.get("VL_EXTRA_CFLAGS", "").split() + ["-c", s, "-o", o]
            if src.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(SO, objs):
        run([cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", SO] + objs + LINK)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
