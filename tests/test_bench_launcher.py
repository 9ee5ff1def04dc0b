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
    assert rc != 0
    assert "giving up on purpose" in err and "ended with rc" in err
    # rank 0 may have printed its line before the other rank failed: whatever reaches stdout is ONE line that says the run failed
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["failed"] is True and rec["rc"] == rc and rec["failing_rank"] == 1 and "error" in rec


@pytest.mark.parametrize("phase", ["comm_init", "calibrate", "comm_stream_trial", "warmup", "timed", "parity"])
def test_launcher_says_which_rank_died_in_which_phase(monkeypatch, phase):
    """A rank that dies in any phase of bench.py's N > 1 run: the launcher still prints one JSON line -- "failed", the rank, the phase
    it was in, the phases of the others, and the per-rank timings the ranks had already recorded -- and ends non-zero."""
    rc, out, err = _launch(monkeypatch, "die:" + phase)
    assert rc != 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["failed"] is True and rec["rc"] == rc
    assert rec["failing_rank"] == 1 and rec["phase"] == phase, rec
    assert set(rec["phases"]) == {"0", "1"} and rec["phases"]["1"] == phase
    assert "dying in phase " + phase in "\n".join(rec["stderr_tail"]) or "dying in phase" in err
    if phase == "parity":
        assert rec["per_rank"] and rec["per_rank"][0]["ms_per_step"] == 0.5


def test_launcher_with_eight_ranks(monkeypatch):
    """The command the driver's scaling run ends with, `--gpus 8`: eight children, one JSON line relayed; and when one of the eight dies in
    the timed region, one JSON line that says so, with all eight phases."""
    rc, out, err = _launch(monkeypatch, "ok", n=8, watchdog=240.0)
    assert rc == 0, err
    rec = json.loads([l for l in out.splitlines() if l.strip()][0])
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == list(range(8))
    rc, out, err = _launch(monkeypatch, "die:timed", n=8, watchdog=240.0)
    assert rc != 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["failed"] is True and rec["failing_rank"] == 7 and rec["phase"] == "timed" and set(rec["phases"]) == {str(r) for r in range(8)}


def test_launcher_names_the_rank_that_hangs(monkeypatch):
    rc, out, err = _launch(monkeypatch, "hang:warmup", watchdog=25.0)
    assert rc == 124
    rec = json.loads([l for l in out.splitlines() if l.strip()][0])
    assert rec["failed"] is True and rec["failing_rank"] == 1 and rec["phase"] == "warmup" and "watchdog" in rec["error"]


def test_launcher_kills_hung_ranks(monkeypatch):
    rc, out, err = _launch(monkeypatch, "hang", watchdog=20.0)
    assert rc == 124
    assert "did not finish" in err
    rec = json.loads([l for l in out.splitlines() if l.strip()][0])
    assert rec["failed"] is True and rec["rc"] == 124


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


def test_expected_compute_only_reads_the_committed_strip_periods():
    """bench.py's N > 1 line quotes the compute-only frame period of the split from profiles/r06_strip_period_c<config>.json
    (tools/strip_period.py): the figure the first multi-GPU run is to be compared with."""
    sys.path.insert(0, ROOT)
    import bench
    for config in (3, 4, 5):
        one = bench.expected_compute_only(config, 1)
        for world in (2, 4, 8):
            e = bench.expected_compute_only(config, world)
            assert e is not None and len(e["per_rank_ms"]) == world == len(e["rows"])
            assert sum(e["rows"]) == (2160 if config == 4 else 1080)
            assert e["ms"] == max(e["per_rank_ms"]) and 0 < e["ms"] < one["ms"]
            assert abs(e["speedup_compute_only"] - one["ms"] / e["ms"]) < 1e-9 and 1.0 < e["speedup_compute_only"] <= world
    assert bench.expected_compute_only(3, 3) is None and bench.expected_compute_only(7, 2) is None


def test_diagnose_names_the_first_failure_of_the_elastic_summary():
    """The failing rank comes from torch.distributed.run's own summary when it printed one, else from the ranks' status files."""
    sys.path.insert(0, ROOT)
    import bench
    tail = ["noise\n", "Root Cause (first observed failure):\n", "[0]:\n", "  time      : 2026\n", "  rank      : 5 (local_rank: 5)\n", "  exitcode  : 3 (pid: 1)\n"]
    status = {r: {"rank": r, "phase": "timed" if r != 5 else "warmup"} for r in range(8)}
    d = bench.diagnose(8, 1, status, tail)
    assert d["failed"] and d["failing_rank"] == 5 and d["phase"] == "warmup" and d["rc"] == 1
    # no summary (a hang killed by the watchdog): the rank that got least far; ranks without a status file died before "start"
    d = bench.diagnose(4, 124, {0: {"phase": "timed"}, 1: {"phase": "timed"}, 3: {"phase": "comm_init"}}, [])
    assert d["failing_rank"] == 2 and d["phase"] == "before start" and "watchdog" in d["error"]
    d = bench.diagnose(2, 124, {0: {"phase": "done"}, 1: {"phase": "first_frames", "mine": {"rank": 1}}}, [])
    assert d["failing_rank"] == 1 and d["phase"] == "first_frames" and d["per_rank"] == [{"rank": 1}]
