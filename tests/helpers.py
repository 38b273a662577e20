"""Shared test helpers: golden-fixture loading (data only; see oracle/make_golden.py)."""
import ast
from pathlib import Path

import numpy as np
import torch

from oracle.dual_eeg_oracle import ModelCfg, synthetic_state_dict

GOLDEN = Path(__file__).resolve().parent / "golden"
WEIGHT_SEED = 20260128
FULL_CONFIGS = ["cfg1_a1_2class", "cfg2_concat", "cfg3_xattn", "cfg5_a2_spec", "a3_ibs_scalar", "a5_full",
                "b1_no_inorm", "b2_phase", "b3_amplitude", "a5_c32"]   # a5_c32: the reference's default in_channels = 32 (S = 139)
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


# ------------------------------------------------------------------------------------------------------
# CPU replica of the HIP path's counter-based dropout (eyegaze_multimodal_amd/csrc/common.h: eg_hash / eg_dropout):
# lets the oracle run TRAIN mode with exactly the masks the kernels draw, so train-mode parity is exact, not statistical.
# ------------------------------------------------------------------------------------------------------
def hip_keep_mask(seed: int, site: int, idx: np.ndarray, p: float) -> np.ndarray:
    """keep[i] for element indices idx (uint32 array): one 32-bit hash per PAIR of consecutive elements, 16 bits each."""
    from eyegaze_multimodal_amd.engine import scramble_seed
    M = np.uint64(0xFFFFFFFF)
    seed = scramble_seed(seed)                      # Engine.set_state does the same before filling eg_step_state
    seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    idx = idx.astype(np.uint64)
    k = np.uint64((site * 0x9E3779B9) & 0xFFFFFFFF)
    a = ((seed_lo ^ k) * np.uint64(0x85EBCA6B)) & M
    a ^= a >> np.uint64(15)
    b = (((seed_hi + k) & M) * np.uint64(0xC2B2AE35)) & M
    b ^= b >> np.uint64(13)
    x = (idx >> np.uint64(1)) ^ a
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x + b) & M
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    half = np.where((idx & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
    return half >= np.uint64(int(p * 65536.0 + 0.5))


def hip_dropout_override(seed: int, B: int, num_layers: int, ctx: dict):
    """Returns a DROPOUT_OVERRIDE for oracle.dual_eeg_oracle that reproduces the engine's masks (engine.py: SITE_* ids,
    _layer_sites; index = row * N + column with rows ordered (stream, window, token); attention: ((w*H+h)*S+q)*Sp2 + key)."""
    from eyegaze_multimodal_amd import engine as E

    def fn(tens, p, site):
        st = ctx["stream"]
        kind = site[0]
        shp = tens.shape
        if kind == "conv":                                   # [B, d, T_i]  <-  rows (window, t), columns = channel
            Bn, d, Tn = shp
            w = (st * B + torch.arange(Bn)).view(Bn, 1, 1)
            idx = (w * Tn + torch.arange(Tn).view(1, 1, Tn)) * d + torch.arange(d).view(1, d, 1)
            sid = E.SITE_CONV0 if site[1] == 0 else E.SITE_CONV1
        elif kind == "attn":                                 # [B, H, S, S]
            Bn, H, S, _ = shp
            Sp2 = (S + 1) & ~1
            w = (st * B + torch.arange(Bn)).view(Bn, 1, 1, 1)
            idx = ((w * H + torch.arange(H).view(1, H, 1, 1)) * S + torch.arange(S).view(1, 1, S, 1)) * Sp2 + torch.arange(S).view(1, 1, 1, S)
            layer = num_layers if site[1].startswith("cross_attn") else int(site[1].split(".")[2])
            sid = E._layer_sites(layer)["attn"]
        elif kind in ("drop1", "ffn_a", "ffn_b", "drop2", "xdrop1"):   # [B, S, N]
            Bn, S, N = shp
            m = ((st * B + torch.arange(Bn)).view(Bn, 1, 1) * S + torch.arange(S).view(1, S, 1))
            idx = m * N + torch.arange(N).view(1, 1, N)
            sid = E._layer_sites(num_layers)["drop1"] if kind == "xdrop1" else E._layer_sites(site[1])[kind]
        elif kind == "cls":                                  # [B, d]
            Bn, N = shp
            idx = torch.arange(Bn).view(Bn, 1) * N + torch.arange(N).view(1, N)
            sid = E.SITE_CLS
        else:
            raise KeyError(site)
        keep = torch.from_numpy(hip_keep_mask(seed, sid, idx.numpy().astype(np.uint32), p))
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        return torch.where(keep, tens * float(scale), torch.zeros((), dtype=tens.dtype))
    return fn
