"""csrc/filter_plan.hpp on the CPU: the MFMA batch filter's launch plan (sample size, sampling grid, pass-1 stages) keeps
its invariants over a sweep of index sizes, chunk counts and knobs, and the defaults are the measured ones."""
import os  # the native CPU tests run under AddressSanitizer + UBSan (sanitizers on the CPU build only: no GPU ASan on this pool)
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_filter_plan_invariants(tmp_path):
    exe = tmp_path / "filter_plan_test"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe), os.path.join(ROOT, "tests", "native", "filter_plan_test.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "plans checked" in r.stdout
