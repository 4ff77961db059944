"""csrc/coalescer.hpp on the CPU: a pass that throws releases every waiting caller (followers used to
block forever, and every later coalesced search with them)."""
import os  # the native CPU tests run under AddressSanitizer + UBSan (sanitizers on the CPU build only: no GPU ASan on this pool)
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_throwing_pass_releases_every_caller(tmp_path):
    exe = tmp_path / "coalescer_test"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe),
                           os.path.join(ROOT, "tests", "native", "coalescer_test.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)  # a deadlock shows up as the timeout
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "answered" in r.stdout
