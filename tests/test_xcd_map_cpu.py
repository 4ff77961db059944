"""csrc/xcd_map.hpp on the CPU: the XCD-aware workgroup -> (row-block lane, query chunk) map of k_mfma_rows is a bijection
for every grid shape, and the chunks of a lane share an XCD (its L2) except for at most 7 lanes per grid."""
import os  # the native CPU tests run under AddressSanitizer + UBSan (sanitizers on the CPU build only: no GPU ASan on this pool)
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_xcd_pair_is_a_bijection_and_keeps_a_lane_on_one_xcd(tmp_path):
    exe = tmp_path / "xcd_map_test"
    subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe), os.path.join(ROOT, "tests", "native", "xcd_map_test.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "every map a bijection" in r.stdout
