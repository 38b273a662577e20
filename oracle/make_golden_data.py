"""
TEST INFRASTRUCTURE — golden vectors for the data path.  Runs ONLY in the build container.
Writes a few small synthetic recordings as CSV files into a temporary directory, runs the REFERENCE's
`DualEEGDataset` + `collate_fn` (1_Data/processed/dual_eeg_dataset.py) over them in both preprocessing modes, and stores
the recordings and the batches it produced.  Data only.

Usage:  python oracle/make_golden_data.py      -> tests/golden/dataset_windows.npz
"""
from __future__ import annotations

import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle.make_golden import REF, _load  # noqa: E402

W, STRIDE = 128, 48
LABEL2ID = {"Single": 0, "Competition": 1, "Cooperation": 2}
ITEMS = [
    {"player1": "p1_a", "player2": "p2_a", "class": "Cooperation"},
    {"player1": "p1_b", "player2": "p2_b", "class": "Single"},          # player 2 is shorter and has more channels
    {"player1": "p1_c", "player2": "missing", "class": "Competition"},  # missing file -> skipped
    {"player1": "p1_d", "player2": "p2_d", "class": "Competition"},     # stored transposed (T rows, C columns)
    {"player1": "p1_e", "player2": "p2_e", "class": "Single"},          # shorter than one window -> skipped
]


def recordings():
    rng = np.random.default_rng(7)
    t = np.arange(4096) / 256.0

    def rec(C, n, drift):
        f = rng.uniform(2, 40, (C, 1))
        x = 20e-6 * np.sin(2 * np.pi * f * t[None, :n] + rng.uniform(0, 6.28, (C, 1))) + 5e-6 * rng.standard_normal((C, n))
        return (x + drift * np.linspace(0, 1, n)[None] + 3e-5 * rng.standard_normal((C, 1))).astype(np.float32)
    return {"p1_a": rec(6, 400, 1e-5), "p2_a": rec(6, 400, 0.0), "p1_b": rec(6, 333, 0.0), "p2_b": rec(8, 301, 2e-5),
            "p1_c": rec(6, 300, 0.0), "p1_d": rec(6, 260, 0.0), "p2_d": rec(6, 270, 0.0), "p1_e": rec(6, 100, 0.0),
            "p2_e": rec(6, 100, 0.0)}


def main():
    ds_mod = _load("dual_eeg_dataset", REF / "1_Data" / "processed" / "dual_eeg_dataset.py")
    recs = recordings()
    blob = {f"rec/{k}": v for k, v in recs.items()}
    with tempfile.TemporaryDirectory() as tmp:
        for k, v in recs.items():
            arr = v.T if k in ("p1_d", "p2_d") else v
            np.savetxt(Path(tmp) / f"{k}.csv", arr, delimiter=",", fmt="%.9e")
        for mode in (False, True):
            ds = ds_mod.DualEEGDataset(ITEMS, tmp, LABEL2ID, window_size=W, stride=STRIDE, sampling_rate=256,
                                       enable_preprocessing=mode)
            batch = ds_mod.collate_fn([ds[i] for i in range(len(ds))])
            tag = "car" if mode else "zscore"
            blob[f"{tag}/eeg1"] = batch["eeg1"].numpy()
            blob[f"{tag}/eeg2"] = batch["eeg2"].numpy()
            blob[f"{tag}/labels"] = batch["labels"].numpy()
            blob[f"{tag}/dataset_idx"] = np.asarray(batch["dataset_idx"])
            blob[f"{tag}/starts"] = np.asarray([w["start"] for w in ds.valid_windows])
    out = REPO / "tests" / "golden" / "dataset_windows.npz"
    np.savez_compressed(out, **blob)
    print("wrote", out, {k: v.shape for k, v in blob.items() if "/" in k and not k.startswith("rec")}, out.stat().st_size)


if __name__ == "__main__":
    main()
