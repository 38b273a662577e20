"""Diagnostic (manual, GPU box): launch-size sweep of eg_gemm_nt (K = 256) to separate fixed cost from per-row cost."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["RS_SHORT"] = "skip"
import importlib
import tests.diag_rs_bench as b  # noqa  (RS_SHORT=skip makes the module only define bench())
for N in (256, 768, 1024):
    for pairs in (1, 2, 4, 8, 16):
        groups = 256 // (N // 256)
        M = groups * pairs * 32
        b.bench(M, N, 256, residual=(1 if N == 256 else 0))
