"""Hardware unit test of the wave-level top-64 list (insert / offer) against a host model:
tests/native/toplist_test.hip is compiled with hipcc and run on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_toplist_insert_offer_on_hardware(tmp_path):
    from vectorlite_amd import build as vbuild
    exe = tmp_path / "toplist_test"
    src = os.path.join(ROOT, "tests", "native", "toplist_test.hip")
    subprocess.run([vbuild.hipcc(), f"--offload-arch={vbuild.ARCH}", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-Wno-unused-result", src, "-o", str(exe)], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "total bad 0" in r.stdout, r.stdout[-2000:]


@pytest.mark.gpu
def test_plain_c_caller_runs_the_reference_kats(tmp_path):
    """integration/c/example.c: a C99 program over the C ABI reproduces src/index/flat.rs:187-201 and
    src/index/hnsw.rs:628-633 and the error statuses."""
    import subprocess
    from vectorlite_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(_lib.SO_PATH)
    exe = tmp_path / "example"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "integration", "c", "example.c"),
                           "-L", libdir, "-lvectorlite_amd", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-lm", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok")
