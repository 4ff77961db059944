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


@pytest.mark.gpu
def test_search_cap_bounds_the_write_while_another_thread_grows_the_index(tmp_path):
    """tests/native/search_cap_race.c: buffers sized from an earlier vl_index_len(), a second thread adding rows the
    whole time; vl_index_search_cap never writes past the caller's capacity (single-GPU and row-sharded handle)."""
    from vectorlite_amd import _lib
    libdir = os.path.dirname(_lib.SO_PATH)
    exe = tmp_path / "search_cap_race"
    subprocess.check_call(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "search_cap_race.c"), "-L", libdir, "-lvectorlite_amd",
                           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-lpthread", "-lm", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("ok"), r.stdout


def test_a_failed_device_allocation_leaves_the_handle_and_the_thread_usable():
    """reserve() of far more rows than the card holds must fail with the out-of-memory status and NOTHING else: the
    failed hipMalloc leaves HIP's per-thread sticky error behind, and the next kernel launch of this thread used to
    report it as its own ("launch_scan(...): out of memory" from a search on a perfectly healthy index)."""
    import numpy as np
    import vectorlite_amd as V
    rows = np.random.default_rng(1).standard_normal((1000, 384))
    idx = V.FlatIndex(384)
    idx.add_rows(np.arange(1000, dtype=np.uint64), rows)
    with pytest.raises(V.VectorLiteError) as ei:
        idx.reserve(1_000_000_000)          # 3 TB of f64 rows
    assert "memory" in str(ei.value).lower(), str(ei.value)
    r = idx.search(rows[5], 3, 0)           # same thread, next launch
    assert r[0].id == 5 and len(idx) == 1000
    idx.add(V.Vector(5000, rows[3] * 2.0))
    assert len(idx) == 1001
    hn = V.HNSWIndex(384, 0)                # another handle on the same thread
    hn.add_rows(np.arange(500, dtype=np.uint64), rows[:500])
    assert hn.search(rows[9], 1, 0)[0].id == 9
    m = V.MultiFlatIndex(384, [0, 0], "row_shards")
    with pytest.raises(V.VectorLiteError):
        m.reserve(2_000_000_000)
    m.add_rows(np.arange(1000, dtype=np.uint64), rows)
    assert len(m) == 1000 and m.search(rows[7], 1, 0)[0].id == 7


def test_a_rejected_device_ordinal_does_not_poison_the_next_call():
    """Creating a handle on a device that does not exist fails -- and used to leave HIP's sticky per-thread error behind
    (the destructor of the half-made handle selects the bad ordinal), so the NEXT healthy call of the thread reported
    'invalid device ordinal' from its first kernel launch.  Every ABI entry point now starts from a clean error slate."""
    import numpy as np
    import vectorlite_amd as V
    rows = np.random.default_rng(2).standard_normal((200, 16))
    for make in (lambda: V.FlatIndex(16, device=7), lambda: V.HNSWIndex(16, 0, device=7),
                 lambda: V.MultiFlatIndex(16, [0, 9], "replicas"), lambda: V.MultiFlatIndex(16, [], "row_shards")):
        with pytest.raises(V.VectorLiteError):
            make()
        idx = V.FlatIndex(16)
        idx.add_rows(np.arange(200, dtype=np.uint64), rows)      # first launches after the failure
        assert idx.search(rows[3], 2, 0)[0].id == 3
    # absurd k: min(k, len) results, no allocation sized by the caller's k
    assert len(idx.search_arrays(rows[3], 2 ** 63, 0)[0]) == 200
    assert idx.search_batch(rows[:3], 2 ** 62, 0)[2].tolist() == [200, 200, 200]
