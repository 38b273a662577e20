"""Wide-tile GEMM (csrc/widegemm.hip: 160 x 256 tile, 3-stage LDS-DMA ring) against the 128 x 128 tiled kernel: the same
k-ordered MFMA chain, so eg_gemm_nt must give BIT-IDENTICAL results whichever kernel serves the call (a child process with
EYEGAZE_WIDE=0 EYEGAZE_RS=0 runs the tiled kernel).  Shapes: the encoder's N = 256 products at every K they occur with."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_rsgemm import run_case  # noqa: E402

REPO = Path(__file__).resolve().parent.parent
CASES = [
    # M, N, residual, gate, act, drop, out_pre, ln, K
    (33280, 256, 1, 0, 0, 0.1, 0, 0, 1024),    # FFN-2: dropout + residual
    (33280, 256, 1, 0, 0, 0.0, 0, 0, 768),     # q|k|v backward-data + residual
    (33280, 256, 0, 1, 0, 0.0, 0, 0, 1792),    # conv-1 backward-data phase: gate
    (33280, 256, 1, 0, 1, 0.1, 1, 0, 6400),    # conv-1 forward: ReLU + dropout + second output + residual, 100 K steps
    (4173, 256, 1, 0, 0, 0.2, 0, 0, 128),      # ragged M, two K steps
    (1024, 256, 0, 0, 2, 0.0, 0, 0, 64),       # GELU, one K step (below the kernel's K floor: served by the tiled kernel)
    (2048, 256, 0, 0, 0, 0.0, 0, 0, 192),      # three K steps: the ring wraps exactly once
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "M%d_N%d_r%d_g%d_a%d_p%g_o%d_ln%d_K%d" % c)
def test_wide_gemm_is_bit_identical_to_the_tiled_kernel(case, tmp_path):
    *head, K = case
    _, got = run_case(*head, K=K)
    dump = tmp_path / "ref.pt"
    code = ("import sys, torch; sys.path.insert(0, %r); from tests.test_gpu_rsgemm import run_case; "
            "_, r = run_case(*%r, K=%d); torch.save(r, %r)" % (str(REPO), tuple(head), K, str(dump)))
    env = dict(os.environ, EYEGAZE_RS="0", EYEGAZE_WIDE="0")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=str(REPO))
    ref = torch.load(dump, weights_only=True)
    assert torch.equal(got["out"], ref["out"]), float((got["out"] - ref["out"]).abs().max())
    if head[6]:
        assert torch.equal(got["pre"], ref["pre"])
