"""Diagnostic (manual, GPU box): time the row-stream GEMM with parts switched off (EYEGAZE_RS_DBG bits, wrong results)."""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
code = '''
import sys; sys.path.insert(0, %r)
import tests.diag_rs_bench as b
''' % str(REPO)
for dbg in (0, 1, 2, 4, 8, 16, 3, 7, 15, 31):
    env = dict(os.environ, EYEGAZE_RS_DBG=str(dbg), RS_SHORT="1")
    print("== dbg", dbg, "(1 no stores, 2 no mfma, 4 no vmcnt wait, 8 no ring refill, 16 no barrier)", flush=True)
    subprocess.run([sys.executable, str(REPO / "tests" / "diag_rs_bench.py")], env=env)
