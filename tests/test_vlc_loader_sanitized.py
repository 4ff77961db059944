"""The .vlc streaming reader (csrc/vlc_loader.cpp: parses files it did not write) under AddressSanitizer + UBSan on
the CPU build -- GPU sanitizers are not available on this pool, and the reader's structural pass, number conversion
and error paths run on the host anyway.  The reader is compiled host-only with -fsanitize=address,undefined, linked
with the ordinary objects into a diagnostic library, and tests/test_persistence_format.py's CPU tests (among them
1500 seeded mutations of valid documents) run against it in a child process with the sanitizer runtime preloaded."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vlc_reader_cpu_tests_under_asan_ubsan(tmp_path):
    sys.path.insert(0, ROOT)
    from vectorlite_amd import build as vbuild
    if not os.path.exists(vbuild.SO):
        vbuild.build()
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("clang's ASan runtime is not in this image")
    cc = vbuild.hipcc()
    obj = str(tmp_path / "vlc_loader_asan.o")
    lib = str(tmp_path / "libvl_asanvlc.so")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    subprocess.check_call([cc, f"--offload-arch={vbuild.ARCH}", "-x", "hip", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC",
                           "-ffp-contract=off", f"-I{vbuild.INCLUDE}"] + san +
                          ["-c", os.path.join(vbuild.CSRC, "vlc_loader.cpp"), "-o", obj])
    others = [o for o in sorted(glob.glob(os.path.join(vbuild.OBJ, "*.o"))) if os.path.basename(o) != "vlc_loader.o"]
    assert others, "the ordinary objects are built by vectorlite_amd.build"
    subprocess.check_call([cc, f"--offload-arch={vbuild.ARCH}", "-shared", "-fPIC", "-shared-libsan"] + san + ["-o", lib] + others + [obj] + vbuild.LINK)
    env = dict(os.environ, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", VL_LIB_PATH=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_persistence_format.py"), "-x", "-q",
                        "-m", "not gpu", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
