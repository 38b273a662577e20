"""GPU, two ranks on one card (gloo): the data-parallel multimodal logit-fusion step (BASELINE configs[4]) end to end --
three-set parameter broadcast, ddp.MultimodalReducers fed by the engines' backwards, 1/world folded into the shared clip and the
per-group AdamW, the collective overflow flag of the fp16 loss scaler.  (The driver's 8-GPU run uses the same code over RCCL.)"""
import json
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def test_two_rank_multimodal_step(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(REPO / "tests" / "ddp_mm5_gpu_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(REPO), timeout=900)
    # on failure show each rank's own traceback (torchrun's summary alone hides the root cause)
    assert res.returncode == 0, "\n".join([ln for ln in res.stderr.splitlines() if "[rank0]" in ln][-40:]
                                          + [ln for ln in res.stderr.splitlines() if "[rank1]" in ln][-25:]) + res.stderr[-1500:]
    r0 = json.loads((tmp_path / "rank0.json").read_text())
    r1 = json.loads((tmp_path / "rank1.json").read_text())
    for r in (r0, r1):
        assert r["A_active"] and r["A_identical"], r
        # rank 1 overflowed in step 1; BOTH ranks skipped exactly that step, halved the scale, and stayed bit-identical
        assert r["B_identical"] and r["B_finite"], r
        assert r["B_skipped"] == [0, 1, 1] and r["B_opt_steps"] == 2, r
        assert r["B_scales"][1] == 0.5 * r["B_scales"][0] and r["B_scales"][2] == r["B_scales"][1], r
    assert r0["A_moved"] > 1e-4
    assert r0["A_param_rel_err_after_2_steps"] < 5e-3, r0       # two clip + AdamW steps: replicas == the single global-batch run
