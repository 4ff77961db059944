"""csrc/rwlock.hpp on the CPU (AddressSanitizer + UBSan build): writers are not starved by overlapping readers, readers
run side by side, writers exclude everybody."""
import os  # the native CPU tests run under AddressSanitizer + UBSan (sanitizers on the CPU build only: no GPU ASan on this pool)
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_writers_are_not_starved_by_overlapping_readers(tmp_path):
    exe = tmp_path / "rwlock_test"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-DVL_RWLOCK_DEBUG",
                           "-o", str(exe), os.path.join(ROOT, "tests", "native", "rwlock_test.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "worst wait" in r.stdout
