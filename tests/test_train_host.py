"""Host-side pieces of the train_art.py counterpart that need no GPU: the PRODUCT macro metrics against the reference's own
known answer and against sklearn, the checkpoint key sets (train_art.py:469-475, :482-489), optimizer / scheduler state
interchange with torch.optim, the six-key loss dict, and the window-shard cache fingerprint."""
import json

import numpy as np
import pytest
import torch

from eyegaze_multimodal_amd import DualEEGTransformer, HipAdamW
from eyegaze_multimodal_amd import train_art as TA


def test_product_macro_metrics_reference_known_answer():
    """5_Metrics/classification_metrics.py:436-472 — np.random.seed(42) data -> Accuracy 0.8600, F1 (macro) 0.8612,
    CM [[30,2,1],[5,30,1],[3,2,26]] (measured from the reference)."""
    np.random.seed(42)
    n = 100
    y_true = np.random.randint(0, 3, n)
    y_pred = y_true.copy()
    err = np.random.choice(n, 20, replace=False)
    y_pred[err] = np.random.randint(0, 3, 20)
    cm = np.zeros((3, 3), int)
    for t_, p_ in zip(y_true, y_pred):
        cm[t_, p_] += 1
    assert cm.tolist() == [[30, 2, 1], [5, 30, 1], [3, 2, 26]]
    m = TA.macro_metrics(y_true, y_pred)
    assert abs(m["eval/accuracy"] - 0.86) < 1e-12
    assert abs(m["eval/f1"] - 0.8612) < 5e-5


@pytest.mark.parametrize("seed", range(6))
def test_product_macro_metrics_equal_sklearn(seed):
    """the sklearn calls of train_art.py:299-304 (macro average, zero_division=0), including classes that are never predicted
    or never present"""
    sk = pytest.importorskip("sklearn.metrics")
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 200))
    yt = rng.integers(0, 3, n)
    yp = rng.integers(0, 3 if seed % 2 else 2, n)
    if seed == 4:
        yt[:] = 1
    m = TA.macro_metrics(yt, yp)
    p, r, f, _ = sk.precision_recall_fscore_support(yt, yp, average="macro", zero_division=0)
    assert abs(m["eval/accuracy"] - sk.accuracy_score(yt, yp)) < 1e-12
    assert abs(m["eval/precision"] - p) < 1e-12 and abs(m["eval/recall"] - r) < 1e-12 and abs(m["eval/f1"] - f) < 1e-12


class _Tr:
    def __init__(self):
        torch.manual_seed(0)
        self.model = DualEEGTransformer(in_channels=8, num_classes=3, d_model=64, num_layers=1, num_heads=2, d_ff=128, max_len=256,
                                        use_spectrogram=False, use_ibs=False)
        self.model._flat.ensure(torch.device("cpu"))
        self.opt = HipAdamW(self.model, lr=1e-4, weight_decay=0.01)
        self.opt.set_epoch(3, 10)


def test_checkpoint_key_sets_match_the_reference():
    tr = _Tr()
    best = TA.checkpoint_dict(tr, 4, {"a": 1}, best_f1=0.5)
    assert list(best) == ["epoch", "model_state_dict", "optimizer_state_dict", "best_f1", "config"]            # train_art.py:469-475
    per = TA.checkpoint_dict(tr, 4, {"a": 1}, metrics={"x": 1.0}, periodic=True)
    assert list(per) == ["epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "metrics", "config"]  # :482-489


def test_optimizer_state_interchanges_with_torch_adamw():
    tr = _Tr()
    tr.opt._ensure()
    tr.opt.t = 7
    tr.opt.m.normal_()
    tr.opt.v.uniform_()
    sd = tr.opt.state_dict()
    ref = torch.optim.AdamW(tr.model.parameters(), lr=1e-4, weight_decay=0.01)
    ref.load_state_dict(sd)                                   # the reference's optimizer accepts our checkpoint
    ps = list(tr.model.parameters())
    assert torch.equal(ref.state[ps[3]]["exp_avg"], sd["state"][3]["exp_avg"]) and float(ref.state[ps[0]]["step"]) == 7.0
    back = HipAdamW(tr.model, lr=5e-5)
    back.load_state_dict(ref.state_dict())                    # and ours accepts torch's
    sd2 = back.state_dict()                                   # (the flat buffers also hold alignment padding: compare per parameter)
    assert back.t == 7 and back.lr == sd["param_groups"][0]["lr"]
    for i in sd["state"]:
        assert torch.equal(sd2["state"][i]["exp_avg"], sd["state"][i]["exp_avg"])
        assert torch.equal(sd2["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4), T_max=10)
    sched.load_state_dict(tr.opt.scheduler_state_dict())
    assert sched.last_epoch == 3 and sched.T_max == 10


def test_train_metric_keys_are_the_reference_six():
    assert tuple("train/" + k for k in TA.TRAIN_KEYS) == ("train/loss", "train/loss_ce", "train/loss_sym", "train/loss_ibs",
                                                         "train/loss_ibs_cls", "train/loss_ibs_contrastive")   # train_art.py:248-255


def test_window_shard_cache_is_rebuilt_when_the_data_settings_change(tmp_path):
    rng = np.random.default_rng(0)
    eeg = tmp_path / "eeg"
    eeg.mkdir()
    items = []
    for i in range(5):
        for who in (1, 2):
            np.savetxt(eeg / f"r{i}_{who}.csv", rng.standard_normal((4, 96)).astype(np.float32), delimiter=",", fmt="%.6e")
        items.append({"player1": f"r{i}_1", "player2": f"r{i}_2", "class": "A" if i % 2 else "B"})
    (tmp_path / "meta.json").write_text(json.dumps(items))
    cfg = {"data": {"metadata_path": str(tmp_path / "meta.json"), "eeg_base_path": str(eeg), "label2id": {"A": 0, "B": 1},
                    "window_size": 32, "stride": 32, "train_test_split": 0.4, "random_seed": 1, "max_samples": None}}
    out = tmp_path / "shards"
    d1 = TA.prepare_shards(cfg, out)
    n1 = json.loads((d1["train"] / "index.json").read_text())["count"] + json.loads((d1["test"] / "index.json").read_text())["count"]
    assert n1 == 5 * 3 and (out / "READY").read_text() == TA.shard_fingerprint(cfg)
    stamp = (d1["train"] / "index.json").stat().st_mtime_ns
    TA.prepare_shards(cfg, out)                               # same settings: cache hit
    assert (d1["train"] / "index.json").stat().st_mtime_ns == stamp
    cfg["data"]["stride"] = 16                                # changed settings: stale windows must not be reused
    assert TA.shard_fingerprint(cfg) != (out / "READY").read_text()
    d2 = TA.prepare_shards(cfg, out)
    n2 = json.loads((d2["train"] / "index.json").read_text())["count"] + json.loads((d2["test"] / "index.json").read_text())["count"]
    assert n2 == 5 * 5
