"""Shared test helpers: golden-fixture loading (data only; see oracle/make_golden.py)."""
import ast
from pathlib import Path

import numpy as np
import torch

from oracle.dual_eeg_oracle import ModelCfg, synthetic_state_dict

GOLDEN = Path(__file__).resolve().parent / "golden"
WEIGHT_SEED = 20260128
FULL_CONFIGS = ["cfg1_a1_2class", "cfg2_concat", "cfg3_xattn", "cfg5_a2_spec", "a3_ibs_scalar", "a5_full",
                "b1_no_inorm", "b2_phase", "b3_amplitude"]
TINY_CONFIGS = ["tiny_full", "tiny_a1"]
ALL_CONFIGS = FULL_CONFIGS + TINY_CONFIGS


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    kw = ast.literal_eval(str(z["cfg_json"]))
    cfg = ModelCfg(**kw)
    sd = synthetic_state_dict(cfg, WEIGHT_SEED)
    assert list(sd.keys()) == [str(k) for k in z["state_keys"]]
    return z, kw, cfg, sd


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))
