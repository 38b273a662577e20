"""GPU, two ranks on one card (gloo): the data-parallel step end to end — parameter broadcast, bucketed all-reduce fed by
Engine.backward's segment callbacks, grad_scale folded into clip/AdamW — equals the single-process step on the global batch.
(The driver's multi-GPU benchmark uses the same code with backend nccl = RCCL.)"""
import json
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_two_rank_step_equals_global_batch_step(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(REPO / "tests" / "ddp_gpu_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(REPO), timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    r0 = json.loads((tmp_path / "rank0.json").read_text())
    r1 = json.loads((tmp_path / "rank1.json").read_text())
    assert r0["same_params_as_rank0"] and r1["same_params_as_rank0"]
    assert r0["moved"] > 1e-4                                    # the optimiser really stepped
    assert r0["segments"] == ["heads", "cross", "encoder.norm", "layer1", "layer0", "tokens", "conv1", "frontend"]
    assert r0["grad_rel_err"] < 2e-5, r0                         # mean of shard gradients == global-batch gradient (f32)
    assert r0["param_rel_err_after_2_steps"] < 2e-3, r0          # two clip + AdamW steps later the replicas match the single run
