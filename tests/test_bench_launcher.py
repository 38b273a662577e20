"""bench.py as its own launcher: `python bench.py --gpus N` (no torchrun) must start N rank processes, and rank 0's single
JSON line must report n_gpus == N.  Runs on the CPU (gloo, --plan-only: the ranks form the process group and exit)."""
import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", REPO / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plan_ranks_one_process_per_gpu():
    b = _bench()
    assert b.plan_ranks(1, {}) == []
    assert b.plan_ranks(8, {"WORLD_SIZE": "8"}) == []          # torchrun already made this process a rank
    plans = b.plan_ranks(4, {"PATH": "x"}, port=29999)
    assert [p["RANK"] for p in plans] == ["0", "1", "2", "3"]
    assert [p["LOCAL_RANK"] for p in plans] == ["0", "1", "2", "3"]
    assert all(p["WORLD_SIZE"] == "4" and p["MASTER_ADDR"] == "127.0.0.1" and p["MASTER_PORT"] == "29999" for p in plans)
    assert all(p["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and p["PATH"] == "x" for p in plans)
    assert [p["LOCAL_RANK"] for p in b.plan_ranks(2, {}, share_gpu=True, port=1)] == ["0", "0"]


def test_default_workload_is_the_cross_attention_config():
    b = _bench()
    assert "cfg3" in b.WORKLOADS and b.WORKLOADS["cfg3"][0]["use_cross_attention"] is True
    src = (REPO / "bench.py").read_text()
    assert '"--workload", default="cfg3"' in src


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--plan-only"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["workload"] == "cfg3"


def test_gpus_2_multimodal_workload_spawns_two_ranks():
    """BASELINE configs[4] is a multi-GPU configuration: --workload mm5 must accept N > 1 (it used to exit)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--workload", "mm5", "--backend", "gloo",
                        "--share-gpu", "--plan-only"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["workload"] == "mm5"
    assert "single-GPU measurement" not in (REPO / "bench.py").read_text()


def test_launcher_ends_the_run_when_one_rank_dies(tmp_path):
    """one rank exits non-zero at once, its sibling would sleep for minutes: the launcher must return that status promptly
    and leave no child behind"""
    import time
    b = _bench()
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "open(os.environ['PIDFILE'] + os.environ['RANK'], 'w').write(str(os.getpid()))\n"
                      "sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(600)\n")
    plans = [dict(os.environ, RANK=str(r), PIDFILE=str(tmp_path / "pid")) for r in range(2)]
    real = b.Path
    try:
        b.Path = lambda *_a, **_k: real(script)          # the launcher starts `python <this file>`: point it at the stub rank
        t0 = time.monotonic()
        rc = b.launch_ranks(plans, argv=[], timeout_s=120)
    finally:
        b.Path = real
    assert rc == 7 and time.monotonic() - t0 < 60
    pid0 = int((tmp_path / "pid0").read_text())
    time.sleep(0.2)
    alive = True
    try:
        os.kill(pid0, 0)
        alive = Path(f"/proc/{pid0}/stat").read_text().split()[2] != "Z"
    except (ProcessLookupError, FileNotFoundError):
        alive = False
    assert not alive


def test_launcher_timeout(tmp_path):
    import time
    b = _bench()
    script = tmp_path / "rank.py"
    script.write_text("import time\ntime.sleep(600)\n")
    real = b.Path
    try:
        b.Path = lambda *_a, **_k: real(script)
        t0 = time.monotonic()
        rc = b.launch_ranks([dict(os.environ), dict(os.environ)], argv=[], timeout_s=2)
    finally:
        b.Path = real
    assert rc == 124 and time.monotonic() - t0 < 40
