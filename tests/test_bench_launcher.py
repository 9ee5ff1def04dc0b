"""`python bench.py --gpus N` without WORLD_SIZE (the form the driver's scaling run may use) must start its own ranks: bench.launch_ranks
with a stub rank (tests/helpers/stub_rank.py; gloo, no GPU) -- the JSON line is relayed, a failed rank and a hung rank end the
launcher non-zero."""
import io
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "stub_rank.py")


def _launch(monkeypatch, mode, n=2, watchdog=120.0):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("STUB_MODE", mode)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(n, ["--steps", "3"], script=STUB, watchdog=watchdog, out=out, err=err)
    return rc, out.getvalue(), err.getvalue()


def test_launcher_relays_rank0_json(monkeypatch):
    rc, out, err = _launch(monkeypatch, "ok")
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1                                  # ONE JSON line on stdout, the chatter went to stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == [0, 1] and rec["argv"] == ["--steps", "3"]
    assert "chatter" in err


def test_launcher_reports_a_failed_rank(monkeypatch):
    rc, out, err = _launch(monkeypatch, "fail")
    assert rc != 0                                          # (rank 0 may have printed its line already: it is relayed, the rc decides)
    assert "giving up on purpose" in err and "ended with rc" in err


def test_launcher_kills_hung_ranks(monkeypatch):
    rc, out, err = _launch(monkeypatch, "hang", watchdog=20.0)
    assert rc == 124
    assert "did not finish" in err


def test_bench_main_takes_the_launcher_branch_before_any_gpu_call(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in a process without a GPU: the launcher branch runs (its ranks then fail on the
    missing GPU and the launcher says so) -- instead of the SystemExit of earlier rounds."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_WATCHDOG"] = "100"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert "starting 2 ranks" in p.stderr
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "needs an MI355X" in p.stderr
