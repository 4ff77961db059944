"""Readers, writers and cloners at once on one handle (tools/mutate_while_searching.py): four searcher threads, a clone +
export + lookup thread, one adder and one deleter.  The reference orders these with a fair RwLock in the caller
(src/client.rs:245,333,383,398); the library's own handle lock must do the same for hosts that call the C ABI from
their own threads: nothing hangs, every answer is well-formed, writers are not starved (csrc/rwlock.hpp), and the
final state equals an oracle that applied the same mutations."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["flat", "hnsw", "replicas", "row_shards"])
def test_writers_readers_and_cloners_at_once(kind):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mutate_while_searching.py"), kind],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "errors []" in r.stdout and "hung 0" in r.stdout and r.stdout.rstrip().endswith("ok"), r.stdout[-1000:]
    if kind != "hnsw":
        assert "final state == oracle" in r.stdout
