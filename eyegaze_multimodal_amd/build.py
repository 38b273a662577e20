"""Builds libeyegaze_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is a plain
C-ABI shared object (include/eyegaze_hip.h); Python binds it with ctypes (eyegaze_multimodal_amd/_lib.py)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libeyegaze_hip.so"
SOURCES = ["prep.hip", "gemm.hip", "rsgemm.hip", "widegemm.hip", "ffn.hip", "lnproj.hip", "attnblock.hip", "norm.hip", "attention.hip", "heads.hip", "optim.hip", "signal.hip", "spec.hip", "aux.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _stale(obj: Path, src: Path) -> bool:
    if not obj.exists():
        return True
    deps = [src, CSRC / "common.h", PKG.parent / "include" / "eyegaze_hip.h"]
    return any(d.stat().st_mtime > obj.stat().st_mtime for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    objs = [CSRC / (s.stem + ".o") for s in srcs]

    def cc(pair):
        src, obj = pair
        if force or _stale(obj, src):
            cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return True
        return False

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        rebuilt = list(ex.map(cc, zip(srcs, objs)))
    if any(rebuilt) or not LIB.exists() or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
