"""Training / evaluation loop for the HIP DualEEGTransformer — the MI355X counterpart of the reference's
4_Experiments/scripts/train_art.py (train_epoch :142-255, evaluate :258-314, main :317-514).

Same command line (`python train_art.py --config <yaml>`, the form run_experiments.py:322-326 spawns) and same
YAML schema (4_Experiments/configs/dual_eeg_transformer.yaml: ablation / model / data / training / system / wandb).
Differences, all additive:
  * the step runs natively: forward + backward + clip(1.0) + AdamW are HIP kernels over flat buffers
    (no autograd graph); the auxiliary batch-level losses (off by default, yaml :96-101) are evaluated on the [B, d]
    outputs and their gradients w.r.t. cls1 / cls2 / ibs_token are injected into the HIP backward;
  * one process per GPU when launched under torchrun: per-rank sample sharding r::world, bucketed RCCL
    all-reduce overlapped with backward (ddp.py);
  * `data.synthetic: true` (or a missing EEG directory) trains on the class-conditional synthetic windows of
    data.py — the reference's CSVs are not distributed with it;
  * wandb is optional (imported only when training.report_to == ['wandb'] and the module exists).
"""
from __future__ import annotations

import argparse
import logging
import math
import os
import sys
from pathlib import Path
from typing import Any, Dict

import numpy as np
import torch
import torch.distributed as dist
import yaml

if __package__ in (None, ""):
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    __package__ = "eyegaze_multimodal_amd"

from . import DualEEGTransformer, HipAdamW  # noqa: E402
from .data import WindowShards, build_window_shards, synth_windows  # noqa: E402
from .ddp import GradAllReducer, broadcast_params, bucket_ranges, shard_indices  # noqa: E402

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
logger = logging.getLogger("train_art")


def load_config(path: str) -> Dict[str, Any]:
    with open(path, "r", encoding="utf-8") as f:
        return yaml.safe_load(f)


def build_model(config: Dict[str, Any], compute_dtype=None) -> DualEEGTransformer:
    """Constructor call of train_art.py:360-385, key for key."""
    ab = config.get("ablation", {})
    m, d, t = config["model"], config["data"], config["training"]
    return DualEEGTransformer(
        in_channels=m["in_channels"], num_classes=m["num_labels"], d_model=m["d_model"], num_layers=m["num_layers"],
        num_heads=m["num_heads"], d_ff=m["d_ff"], dropout=t["dropout"], max_len=d["window_size"] // 4,
        conv_kernel_size=m["conv_kernel_size"], conv_stride=m["conv_stride"], conv_layers=m["conv_layers"],
        sampling_rate=d["sampling_rate"], use_spectrogram=ab.get("use_spectrogram", True),
        spec_n_fft=m.get("spec_n_fft", 128), spec_hop_length=m.get("spec_hop_length", 64),
        spec_freq_bins=m.get("spec_freq_bins", 64), use_robust_ibs=(ab.get("ibs_mode", "robust") == "robust"),
        use_ibs=ab.get("use_ibs", True), use_cross_attention=ab.get("use_cross_attention", True),
        ibs_instance_norm=ab.get("ibs_instance_norm", True), ibs_feature_type=ab.get("ibs_feature_type", "all"),
        compute_dtype=compute_dtype or config.get("system", {}).get("compute_dtype"))


def macro_metrics(y_true: np.ndarray, y_pred: np.ndarray) -> Dict[str, float]:
    """accuracy + macro precision / recall / F1 with zero_division=0 (the sklearn calls at train_art.py:301-304)."""
    labels = np.union1d(y_true, y_pred)
    P, R, F = [], [], []
    for c in labels:
        tp = float(np.sum((y_pred == c) & (y_true == c)))
        fp = float(np.sum((y_pred == c) & (y_true != c)))
        fn = float(np.sum((y_pred != c) & (y_true == c)))
        p = tp / (tp + fp) if tp + fp > 0 else 0.0
        r = tp / (tp + fn) if tp + fn > 0 else 0.0
        P.append(p)
        R.append(r)
        F.append(2 * p * r / (p + r) if p + r > 0 else 0.0)
    return {"eval/accuracy": float(np.mean(y_true == y_pred)), "eval/precision": float(np.mean(P)),
            "eval/recall": float(np.mean(R)), "eval/f1": float(np.mean(F))}


class Trainer:
    def __init__(self, config: Dict[str, Any], device: torch.device, rank: int = 0, world: int = 1, compute_dtype=None):
        self.config, self.device, self.rank, self.world = config, device, rank, world
        seed = config["system"]["seed"]
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.model = build_model(config, compute_dtype).to(device)
        t = config["training"]
        self.opt = HipAdamW(self.model, lr=t["learning_rate"], weight_decay=t["weight_decay"])
        self.epochs = t["num_train_epochs"]
        self.step_no = 0
        self.reducer = None
        ab = config.get("ablation", {})
        self.has_ibs = ab.get("use_ibs", True)
        self.lam_ibs_cls = t.get("lambda_ibs_cls", 0.5) if (t.get("use_ibs_cls_loss", True) and self.has_ibs) else 0.0
        self.aux = dict(sym=t.get("use_sym_loss", False), ibs=t.get("use_ibs_loss", False) and self.has_ibs,
                        contrastive=t.get("use_ibs_contrastive", True) and self.has_ibs)
        self.lams = dict(sym=t.get("lambda_sym", 0.1), ibs=t.get("lambda_ibs", 0.1),
                         contrastive=t.get("lambda_ibs_contrastive", 0.3))

    def _engine(self, B, T):
        eng = self.model.engine(B, T, self.device)
        if self.world > 1 and (self.reducer is None or self.reducer.g is not self.model._flat.grad):
            fp = self.model._flat
            broadcast_params(fp.flat)
            rg = bucket_ranges(fp.names, fp.offsets, fp.total, self.model.cfg.num_layers, self.model.cfg.use_cross_attention)
            self.reducer = GradAllReducer(fp.grad, rg)
        return eng

    def train_step(self, eeg1, eeg2, labels) -> Dict[str, float]:
        """One optimiser step (train_art.py:171-222).  Returns device scalars (no host sync here)."""
        self.model.train()
        B, _, T = eeg1.shape
        eng = self._engine(B, T)
        self.step_no += 1
        gs = 1.0 / self.world
        self.opt.begin_step(eng, seed=self.config["system"]["seed"] * 7919 + self.step_no * self.world + self.rank, grad_scale=gs)
        hook = self.reducer.on_segment if self.reducer else None
        need_aux = any(self.aux.values())
        zero = torch.zeros((), device=self.device)
        if not need_aux:
            # the whole forward + backward is ONE registered operator (ops.py: eyegaze::dual_eeg_train_step)
            self.model._on_segment = hook
            fp = self.model._flat
            loss_ce, loss_ibs = torch.ops.eyegaze.dual_eeg_train_step(eeg1, eeg2, labels, fp.flat, fp.grad,
                                                                     float(self.lam_ibs_cls if self.has_ibs else 0.0),
                                                                     self.model._op_handle)
            if self.reducer:
                self.reducer.finish()
            self.opt.step(eng)
            li = loss_ibs if (self.has_ibs and self.lam_ibs_cls != 0.0) else zero
            return {"loss": (loss_ce + self.lam_ibs_cls * li).detach(), "loss_ce": loss_ce, "loss_sym": zero, "loss_ibs": zero,
                    "loss_ibs_cls": li, "loss_ibs_contrastive": zero}
        eng.forward(eeg1, eeg2, labels, train=True)
        one = torch.ones(1, device=self.device)
        kw = {}
        # the six entries train_epoch accumulates (train_art.py:224-229), disabled terms stay 0 as there (:183-186)
        losses = {"loss": None, "loss_ce": eng.a["loss"].reshape(()), "loss_sym": zero, "loss_ibs": zero, "loss_ibs_cls": zero,
                  "loss_ibs_contrastive": zero}
        if self.has_ibs:
            kw["gloss_ibs"] = one * self.lam_ibs_cls
            if self.lam_ibs_cls != 0.0:
                losses["loss_ibs_cls"] = eng.a["ibs_loss"].reshape(())
        if need_aux:
            # batch-level losses on [B,d] outputs: evaluated with torch ops, their gradients enter the HIP backward
            cls1 = eng.a["cls1"].clone().requires_grad_(True)
            cls2 = eng.a["cls2"].clone().requires_grad_(True)
            ibs = eng.a["ibs_pool_f"].clone().requires_grad_(True) if self.has_ibs else None
            aux = 0.0
            if self.aux["sym"]:
                losses["loss_sym"] = self.model.compute_symmetry_loss(cls1, cls2)
                aux = aux + self.lams["sym"] * losses["loss_sym"]
            if self.aux["ibs"]:
                losses["loss_ibs"] = self.model.compute_ibs_alignment_loss(ibs, cls1, cls2)
                aux = aux + self.lams["ibs"] * losses["loss_ibs"]
            if self.aux["contrastive"]:
                losses["loss_ibs_contrastive"] = self.model.compute_ibs_contrastive_loss(ibs, labels)
                aux = aux + self.lams["contrastive"] * losses["loss_ibs_contrastive"]
            if torch.is_tensor(aux) and aux.requires_grad:
                aux.backward()
                kw.update(gcls1=cls1.grad, gcls2=cls2.grad, gibs_token=(ibs.grad if ibs is not None else None))
        eng.backward(gloss=one, on_segment=hook, **kw)
        if self.reducer:
            self.reducer.finish()
        self.opt.step(eng)
        losses["loss"] = (losses["loss_ce"] + self.lams["sym"] * losses["loss_sym"] + self.lams["ibs"] * losses["loss_ibs"] +
                          self.lam_ibs_cls * losses["loss_ibs_cls"] + self.lams["contrastive"] * losses["loss_ibs_contrastive"])
        return {k: v.detach() for k, v in losses.items()}

    @torch.no_grad()
    def evaluate(self, batches) -> Dict[str, float]:
        """evaluate() of train_art.py:258-314: eval forward, argmax, macro metrics."""
        self.model.eval()
        preds, labs, tot, n = [], [], 0.0, 0
        for eeg1, eeg2, labels in batches:
            out = self.model(eeg1, eeg2, labels)
            tot += float(out["loss"])
            n += 1
            preds.append(torch.argmax(out["logits"], dim=-1).cpu().numpy())
            labs.append(labels.cpu().numpy())
        empty = np.zeros(0, dtype=np.int64)
        yp, yt = np.concatenate(preds or [empty]), np.concatenate(labs or [empty])   # a rank's shard may hold no batch
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (yp, yt, tot, n))
            yp = np.concatenate([g[0] for g in gathered])
            yt = np.concatenate([g[1] for g in gathered])
            tot, n = sum(g[2] for g in gathered), sum(g[3] for g in gathered)
        m = macro_metrics(yt, yp) if len(yt) else {"eval/accuracy": 0.0, "eval/precision": 0.0, "eval/recall": 0.0, "eval/f1": 0.0}
        return {"eval/loss": tot / max(n, 1), **m}


def split_items(items, test_size: float, seed: int):
    """Item-level train/test split of the metadata rows (reference: HF `Dataset.train_test_split(test_size, seed)`,
    train_art.py:93-109 — its stratified attempt raises on a string column and falls back to the plain split).  Uses the
    same library call when `datasets` is importable so that the split is the reference's; otherwise a seeded permutation
    (documented difference)."""
    try:
        from datasets import Dataset
        sp = Dataset.from_list(list(items)).train_test_split(test_size=test_size, seed=seed)
        return list(sp["train"]), list(sp["test"])
    except ImportError:
        perm = np.random.default_rng(seed).permutation(len(items))
        n_test = int(np.ceil(len(items) * test_size))
        return [items[i] for i in perm[n_test:]], [items[i] for i in perm[:n_test]]


def shard_fingerprint(config: Dict[str, Any]) -> str:
    """Hash of everything the cached windows depend on: a re-run with other data settings must not train on stale shards."""
    import hashlib
    import json
    d = config["data"]
    meta = Path(d["metadata_path"])
    stt = meta.stat() if meta.exists() else None
    key = {k: d.get(k) for k in ("window_size", "stride", "max_samples", "train_test_split", "random_seed", "label2id",
                                 "eeg_base_path", "metadata_path")}
    key["metadata_stat"] = [stt.st_size, int(stt.st_mtime)] if stt else None
    return hashlib.sha256(json.dumps(key, sort_keys=True, default=str).encode()).hexdigest()[:16]


def prepare_shards(config: Dict[str, Any], out_dir: Path, rank: int = 0, world: int = 1, timeout_s: float = 24 * 3600.0) -> Dict[str, Path]:
    """CSV recordings -> windowed shards, once per data configuration.  Rank 0 builds and then writes `<out_dir>/READY`
    holding the fingerprint; the other ranks poll that file (no collective: a long CSV conversion must not run into the
    process group's timeout).  Shards built under another fingerprint are rebuilt."""
    import json
    import shutil
    import time
    d = config["data"]
    dirs = {"train": out_dir / "train", "test": out_dir / "test"}
    fp = shard_fingerprint(config)
    ready = out_dir / "READY"

    def is_ready():
        return ready.exists() and ready.read_text().strip() == fp and all((p / "index.json").exists() for p in dirs.values())

    if rank == 0 and not is_ready():
        if ready.exists():
            ready.unlink()
        for p in dirs.values():
            if p.exists():
                shutil.rmtree(p)              # stale windows of another configuration
        items = json.loads(Path(d["metadata_path"]).read_text())
        if d.get("max_samples"):
            items = items[: d["max_samples"]]
        train_items, test_items = split_items(items, d["train_test_split"], d["random_seed"])
        for name, its in (("train", train_items), ("test", test_items)):
            idx = build_window_shards(its, d["eeg_base_path"], d["label2id"], dirs[name], d["window_size"], d["stride"])
            logger.info(f"{name}: {len(its)} recordings -> {idx['count']} windows in {len(idx['shards'])} shards")
        out_dir.mkdir(parents=True, exist_ok=True)
        ready.write_text(fp)
    t0 = time.time()
    while not is_ready():
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: window shards under {out_dir} did not become ready")
        time.sleep(0.5)
    return dirs


def _batches(x1, x2, y, bs, device, idx):
    """every sample, the tail batch included (the reference's test DataLoader keeps it, train_art.py:343-350)"""
    for i in range(0, len(idx), bs):
        j = idx[i:i + bs]
        yield x1[j].to(device), x2[j].to(device), y[j].to(device)


TRAIN_KEYS = ("loss", "loss_ce", "loss_sym", "loss_ibs", "loss_ibs_cls", "loss_ibs_contrastive")


def checkpoint_dict(tr: "Trainer", epoch: int, config, *, best_f1=None, metrics=None, periodic: bool = False) -> Dict[str, Any]:
    """Key sets of the reference's two torch.save calls (train_art.py:469-475 best model, :482-489 periodic checkpoint)."""
    ck = {"epoch": epoch, "model_state_dict": tr.model.state_dict(), "optimizer_state_dict": tr.opt.state_dict()}
    if periodic:
        ck["scheduler_state_dict"] = tr.opt.scheduler_state_dict()
        ck["metrics"] = metrics
    else:
        ck["best_f1"] = best_f1
    ck["config"] = config
    return ck


def main(args):
    config = load_config(args.config)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train_art.py (HIP) needs an MI355X: there is no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    tr = Trainer(config, device, rank, world, args.dtype)
    d, t = config["data"], config["training"]
    C, T, ncls = config["model"]["in_channels"], d["window_size"], config["model"]["num_labels"]
    eeg_dir = Path(d.get("eeg_base_path", ""))
    use_csv = not d.get("synthetic", False) and eeg_dir.exists() and any(eeg_dir.glob("*.csv"))
    shards = None
    if use_csv:
        shards = prepare_shards(config, Path(t["output_dir"]) / "window_shards", rank, world)   # file-based wait, no collective
    else:
        n_total = d.get("max_samples") or d.get("synthetic_samples", 2048)
        logger.info(f"synthetic class-conditional windows: n={n_total} C={C} T={T} classes={ncls}")
        x1, x2, y = synth_windows(n_total, C, T, ncls, fs=d["sampling_rate"], seed=d["random_seed"])
        n_test = int(round(n_total * d["train_test_split"]))
        perm = np.random.default_rng(d["random_seed"]).permutation(n_total)
        test_idx, train_idx = perm[:n_test], perm[n_test:]
    bs, ebs = t["per_device_train_batch_size"], t["per_device_eval_batch_size"]
    out_dir = Path(t["output_dir"])
    if rank == 0:
        out_dir.mkdir(parents=True, exist_ok=True)
    logger.info(f"Model created with {sum(p.numel() for p in tr.model.parameters())} parameters")
    wandb = None
    if t.get("report_to") == ["wandb"] and rank == 0:
        try:
            import wandb  # type: ignore
            wandb.init(project=config["wandb"]["project"], name=config["wandb"]["run_name"], tags=config["wandb"]["tags"],
                       notes=config["wandb"]["notes"], config=config)
        except ImportError:
            logger.warning("wandb is not installed; continuing without it")
            wandb = None
    best_f1, best_epoch = 0.0, 0
    for epoch in range(tr.epochs):
        tr.opt.set_epoch(epoch, tr.epochs)           # CosineAnnealingLR stepped per epoch (train_art.py:409,494)
        sums, nb = {}, 0
        if shards is not None:
            # one process: every window, tail batch included, as the reference's DataLoader (train_art.py:334-341);
            # data parallel: every rank must take the same number of steps, so the ragged tail is dropped
            train_ld = WindowShards(shards["train"], bs, device, rank, world, shuffle=True, seed=d["random_seed"],
                                    preprocessing=d.get("enable_preprocessing", False), drop_last=world > 1)
            train_ld.set_epoch(epoch)
            train_iter = ((b["eeg1"], b["eeg2"], b["labels"]) for b in train_ld)
        else:
            order = np.random.default_rng(d["random_seed"] + epoch).permutation(train_idx)
            gbs = bs * world
            stop = len(order) if world == 1 else len(order) - gbs + 1
            train_iter = ((x1[m].to(device), x2[m].to(device), y[m].to(device)) for m in
                          (order[i:i + gbs][list(shard_indices(len(order[i:i + gbs]), rank, world))] for i in range(0, stop, gbs)))
        for e1, e2, lab in train_iter:
            losses = tr.train_step(e1, e2, lab)
            for k, v in losses.items():
                sums[k] = sums.get(k, 0.0) + v.float()
            nb += 1
        train_metrics = {f"train/{k}": float(sums.get(k, 0.0)) / max(nb, 1) for k in TRAIN_KEYS}   # train_art.py:248-255
        if shards is not None:
            test_ld = WindowShards(shards["test"], ebs, device, rank, world, preprocessing=d.get("enable_preprocessing", False))
            ev = tr.evaluate((b["eeg1"], b["eeg2"], b["labels"]) for b in test_ld)
        else:
            my_test = test_idx[list(shard_indices(len(test_idx), rank, world))]
            ev = tr.evaluate(_batches(x1, x2, y, min(ebs, max(1, len(my_test))), device, my_test))
        metrics = {**train_metrics, **ev, "epoch": epoch + 1}
        if rank == 0:
            logger.info(" ".join(f"{k}: {v:.4f}" for k, v in metrics.items() if k != "epoch") + f" (epoch {epoch + 1}/{tr.epochs})")
            if wandb:
                wandb.log(metrics)
            if ev["eval/f1"] > best_f1:
                best_f1, best_epoch = ev["eval/f1"], epoch + 1
                torch.save(checkpoint_dict(tr, epoch + 1, config, best_f1=best_f1), out_dir / "best_model.pt")
            if (epoch + 1) % t["save_every_n_epochs"] == 0:
                torch.save(checkpoint_dict(tr, epoch + 1, config, metrics=metrics, periodic=True),
                           out_dir / f"checkpoint-epoch-{epoch + 1}.pt")
    if rank == 0:
        logger.info(f"Training completed! Best F1: {best_f1:.4f} at epoch {best_epoch}")
    # final evaluation on the best model (train_art.py:503-512); every rank loads it so that the sharded evaluate() agrees
    final = None
    best_path = out_dir / "best_model.pt"
    if world > 1:
        dist.barrier()
    if best_path.exists():
        ck = torch.load(best_path, map_location="cpu", weights_only=False)   # our own file, written above
        tr.model.load_state_dict(ck["model_state_dict"])
        if shards is not None:
            test_ld = WindowShards(shards["test"], ebs, device, rank, world, preprocessing=d.get("enable_preprocessing", False))
            final = tr.evaluate((b["eeg1"], b["eeg2"], b["labels"]) for b in test_ld)
        else:
            my_test = test_idx[list(shard_indices(len(test_idx), rank, world))]
            final = tr.evaluate(_batches(x1, x2, y, min(ebs, max(1, len(my_test))), device, my_test))
        if rank == 0:
            logger.info("Final Test Metrics: " + " ".join(f"{k}: {v:.4f}" for k, v in final.items()))
            if wandb:
                wandb.log({f"final/{k}": v for k, v in final.items()})
    if wandb:
        wandb.finish()
    if world > 1:
        dist.destroy_process_group()
    return final


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Train Dual EEG Transformer (MI355X HIP engine)")
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--checkpoint", type=str)
    ap.add_argument("--dtype", type=str, default=None, choices=[None, "bf16", "fp16", "f32"])
    main(ap.parse_args())
