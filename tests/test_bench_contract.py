"""bench.py's own logic, on the CPU: the query plan never depends on (--steps, --warmup), the
supervisor builds the launch line the driver's contract describes, and without a GPU the command
fails loudly (non-zero, nothing on stdout) instead of printing a made-up line.

The GPU half (bench.py as a fresh child process, every contract field present) is
tests/test_gpu_bench_child.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("steps,warmup", [(1, 0), (5, 1), (20, 5), (200, 20), (3, 1), (1000, 0)])
@pytest.mark.parametrize("cpu_queries", [1, 4, 32, 100])
def test_plan_is_independent_of_steps_and_warmup(steps, warmup, cpu_queries):
    p = bench.plan_queries(steps, warmup, cpu_queries)
    assert p["timed"]["n"] == steps + warmup
    # every later block indexes the check set only, and stays inside it
    for blk in ("cpu", "bf16", "exact"):
        assert 1 <= p[blk]["n"] <= p["check"]["n"]
    assert p["cpu"]["n"] == cpu_queries or p["cpu"]["n"] == p["check"]["n"]
    assert p["check"]["n"] >= bench.CHECK_QUERIES_MIN
    # and the plan of those blocks does not move with the timed region's length
    q = bench.plan_queries(steps + 7, warmup + 3, cpu_queries)
    for blk in ("check", "cpu", "bf16", "exact", "prewarm"):
        assert p[blk] == q[blk]
    # three disjoint generators
    assert len({p["timed"]["seed"], p["prewarm"]["seed"], p["check"]["seed"]}) == 3


def test_every_planned_index_exists_for_the_drivers_arguments():
    """Round 1's crash: the CPU baseline indexed the timed queries (25 rows) with range(32)."""
    for steps, warmup in [(20, 5), (1, 0), (5, 1), (200, 20)]:
        p = bench.plan_queries(steps, warmup, 32)
        Q = bench.unit_queries(p["timed"]["seed"], p["timed"]["n"], 8)
        Qc = bench.unit_queries(p["check"]["seed"], p["check"]["n"], 8)
        assert Q.shape == (steps + warmup, 8)
        for i in range(warmup, warmup + steps):
            Q[i]
        for blk in ("cpu", "bf16", "exact"):
            for i in range(p[blk]["n"]):
                Qc[i]
        assert np.allclose(np.linalg.norm(Qc, axis=1), 1.0)


def test_unit_queries_zero_rows():
    assert bench.unit_queries(1, 0, 4).shape == (0, 4)


def test_step_rates():
    assert bench.step_rates([], 0.0) == {}
    r = bench.step_rates([1.0, 2.0, 3.0], 0.0)
    assert r == {"value_first_5_steps": 1.0}
    r = bench.step_rates([0.01 * (i + 1) for i in range(20)], 0.0)
    assert r["value_first_5_steps"] == pytest.approx(100.0)
    assert r["value_after_first_5_steps"] == pytest.approx(100.0)


def test_launch_command_forms():
    argv = ["--gpus", "4", "--steps", "20", "--warmup", "5", "--deadline-s", "9", "--result-file=/x"]
    a = bench.parse_args(argv)
    cmd = bench.launch_command(a, argv, "/tmp/r.json", 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "20", "--warmup", "5", "--worker", "--result-file", "/tmp/r.json"]
    a1 = bench.parse_args(["--steps", "3"])
    c1 = bench.launch_command(a1, ["--steps", "3"], "/tmp/r.json", 1)
    assert c1 == [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--worker", "--result-file", "/tmp/r.json"]


def _no_gpu_here() -> bool:
    import torch
    return not torch.cuda.is_available()


@pytest.mark.parametrize("gpus", [1, 2])
def test_without_a_gpu_the_command_fails_loudly(gpus):
    """No CPU fallback: the supervisor starts the rank(s), they refuse, and the command exits non-zero
    with an empty stdout (N = 2 goes through torch.distributed.run exactly as on the box)."""
    if not _no_gpu_here():
        pytest.skip("a GPU is visible")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2",
                        "--warmup", "1", "--rows", "1000"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "needs a GPU" in r.stderr


def test_supervisor_reports_a_dead_worker_without_losing_the_line(tmp_path, monkeypatch):
    """A worker that published the contract line and then died: the line is printed, with the death
    recorded under "errors", and the supervisor's exit code is 0; no line -> non-zero."""
    fake = tmp_path / "fake_worker.py"
    fake.write_text(
        "import json, os, sys\n"
        "rf = sys.argv[sys.argv.index('--result-file') + 1]\n"
        "if '--publish' in sys.argv:\n"
        "    json.dump({'metric': 'm', 'value': 1.5, 'roofline': {'frac': 0.5}}, open(rf, 'w'))\n"
        "os._exit(7)\n")

    def fake_cmd(args, argv, result_file, port):
        extra = ["--publish"] if getattr(args, "_publish", False) else []
        return [sys.executable, str(fake), "--result-file", result_file] + extra
    monkeypatch.setattr(bench, "launch_command", fake_cmd)
    a = bench.parse_args(["--steps", "2"])
    a._publish = True
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.supervise(a, ["--steps", "2"])
    assert rc == 0
    line = json.loads(buf.getvalue())
    assert line["value"] == 1.5
    assert line["errors"][0]["block"] == "worker_exit" and "7" in line["errors"][0]["error"]
    a._publish = False
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.supervise(a, ["--steps", "2"])
    assert rc == 7 and buf.getvalue() == ""
