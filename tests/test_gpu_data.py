"""GPU parity tests of the data path: `eg_window_normalize` and the shard loader against the batches the reference's
DualEEGDataset + collate_fn produce on the same recordings (tests/golden/dataset_windows.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from eyegaze_multimodal_amd._lib import call, ptr  # noqa: E402
from eyegaze_multimodal_amd.data import WindowShards, build_window_shards  # noqa: E402
from oracle import dual_eeg_oracle as O  # noqa: E402
from tests.test_data_host import FILES, ITEMS, LABEL2ID, STRIDE, W, Z  # noqa: E402

DEV = "cuda"
# fp32: the reference reduces mean/std in float32 (pairwise), the kernel in double; both round to float32
TOL = dict(atol=5e-6, rtol=0)


@pytest.mark.parametrize("mode,tag", [(0, "zscore"), (1, "car")])
def test_shard_loader_matches_reference_batches(tmp_path, mode, tag):
    build_window_shards(ITEMS, None, LABEL2ID, tmp_path, W, STRIDE, shard_windows=4, recordings=FILES)
    for B in (10, 4, 3):      # one batch; ragged tail; batches straddling shard boundaries
        ld = WindowShards(tmp_path, B, DEV, preprocessing=bool(mode))
        got1, got2, lab, idx = [], [], [], []
        for b in ld:
            got1.append(b["eeg1"].cpu()), got2.append(b["eeg2"].cpu()), lab.append(b["labels"].cpu()), idx.extend(b["dataset_idx"])
        assert len(got1) == len(ld)
        np.testing.assert_allclose(torch.cat(got1).numpy(), Z[f"{tag}/eeg1"], **TOL)
        np.testing.assert_allclose(torch.cat(got2).numpy(), Z[f"{tag}/eeg2"], **TOL)
        np.testing.assert_array_equal(torch.cat(lab).numpy(), Z[f"{tag}/labels"])
        assert idx == Z[f"{tag}/dataset_idx"].tolist()


def test_loader_rank_partition_and_shuffle(tmp_path):
    build_window_shards(ITEMS, None, LABEL2ID, tmp_path, W, STRIDE, shard_windows=4, recordings=FILES)
    seen = []
    for r in range(2):
        ld = WindowShards(tmp_path, 3, DEV, rank=r, world=2, shuffle=True, seed=5)
        ld.set_epoch(1)
        rows = torch.cat([b["eeg1"].cpu() for b in ld])
        assert rows.shape[0] == 5
        seen.append(rows)
    got = torch.cat(seen).numpy()
    ref = Z["zscore/eeg1"]
    # every window appears exactly once across the two ranks
    used = set()
    for row in got:
        j = int(np.argmin(np.abs(ref - row[None]).reshape(len(ref), -1).max(1)))
        assert np.abs(ref[j] - row).max() < 1e-5 and j not in used
        used.add(j)
    assert len(used) == 10


@pytest.mark.parametrize("mode", [0, 1])
def test_window_normalize_full_size(mode):
    """BASELINE shape (C=8, T=1024) and the reference default C=32 against the oracle; constant and huge-offset windows."""
    for C, T, N in ((8, 1024, 64), (32, 1024, 8), (5, 1000, 3)):
        g = torch.Generator().manual_seed(C)
        raw = torch.randn(N, 2, C, T, generator=g) * 3e-5 + 1e-3       # EEG-like scale with an offset
        raw[0, 0] = 0.25                                             # constant window: std = 0 -> 0 / 1e-8 = 0
        d = raw.to(DEV)
        e1, e2 = torch.empty(N, C, T, device=DEV), torch.empty(N, C, T, device=DEV)
        call("eg_window_normalize", ptr(d), ptr(e1), ptr(e2), N, C, T, mode, 0)
        torch.cuda.synchronize()
        fn = O.preprocess_window if mode else O.zscore_window
        for n in range(N):
            for who, e in ((0, e1), (1, e2)):
                ref = fn(raw[n, who].numpy())
                np.testing.assert_allclose(e[n].cpu().numpy(), ref, atol=3e-5, rtol=1e-5)
        assert torch.isfinite(e1).all() and float(e1[0].abs().max()) == 0.0


def test_window_normalize_rejects_bad_arguments():
    from eyegaze_multimodal_amd._lib import EgError
    x = torch.zeros(1, 2, 4, 64, device=DEV)
    with pytest.raises(EgError):
        call("eg_window_normalize", ptr(x), ptr(x), ptr(x), 1, 4, 64, 2, 0)
    with pytest.raises(EgError):
        call("eg_window_normalize", 0, ptr(x), ptr(x), 1, 4, 64, 0, 0)
