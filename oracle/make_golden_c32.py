"""
TEST INFRASTRUCTURE — fixture at the reference's DEFAULT channel count (4_Experiments/configs/dual_eeg_transformer.yaml:38-53:
in_channels 32, A5 flags: spectrogram tokens + 42 synchrony tokens of 32x32 = 1024-wide rows + cross-attention; S = 139).
Runs ONLY in the build container (imports the reference by path, exactly as oracle/make_golden.py).  B = 2 window pairs.

Usage:  python oracle/make_golden_c32.py
"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import make_golden as G  # noqa: E402

NAME = "a5_c32"
KW = dict(in_channels=32, max_len=256, num_classes=3, use_spectrogram=True, use_ibs=True, use_robust_ibs=True)

if __name__ == "__main__":
    torch.set_num_threads(8)
    torch.manual_seed(0)
    G.B = 2
    model_mod, gen_mod, _ = G.load_reference()
    G.run_config(NAME, KW, model_mod, gen_mod, REPO / "tests" / "golden")
