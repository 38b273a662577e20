"""Synthetic window generators for throughput and accuracy runs.  Real EEG CSVs are absent
from the reference tree; the reference's own synthesiser is 1_Data/processed/two_EEG_fusion.py:31-49
(sine mixture + Gaussian noise).  `synth_windows` restates that recipe and makes it class-conditional so that
validation accuracy is learnable: stream 2 shares a class-dependent fraction of stream 1's components."""
from __future__ import annotations

import numpy as np
import torch


def zscore_window(x: np.ndarray) -> np.ndarray:
    """per-window global z-score, population std (1_Data/processed/dual_eeg_dataset.py:201-202)"""
    return ((x - x.mean()) / (x.std() + 1e-8)).astype(np.float32)


def _sines(rng, C, T, fs, ncomp):
    t = np.arange(T, dtype=np.float32) / fs
    f = rng.uniform(1.0, 40.0, size=(C, ncomp, 1)).astype(np.float32)
    a = rng.uniform(0.1, 1.0, size=(C, ncomp, 1)).astype(np.float32)
    p = rng.uniform(0.0, 2 * np.pi, size=(C, ncomp, 1)).astype(np.float32)
    return (a * np.sin(2 * np.pi * f * t + p)).sum(1).astype(np.float32)


def synth_windows(n: int, C: int = 8, T: int = 1024, num_classes: int = 3, fs: float = 256.0, noise_std: float = 0.1,
                  seed: int = 0):
    """Returns eeg1, eeg2 [n,C,T] f32 (z-scored per window) and labels [n] i64.
    class c: stream 2 = coupling[c] * (stream 1 delayed by lag[c]) + (1-coupling[c]) * independent mixture, and both
    streams carry a class-specific rhythm (6 / 11 / 20 / 31 / 43 Hz, random phase per window) on half of the channels."""
    rng = np.random.default_rng(seed)
    coupling = np.linspace(0.0, 0.9, num_classes)
    lags = [0, 7, 19, 3, 11][:num_classes]
    marker = [6.0, 11.0, 20.0, 31.0, 43.0][:num_classes]
    tt = np.arange(T, dtype=np.float32) / fs
    x1 = np.zeros((n, C, T), np.float32)
    x2 = np.zeros((n, C, T), np.float32)
    y = rng.integers(0, num_classes, size=n)
    for i in range(n):
        a = _sines(rng, C, T, fs, 3) + rng.normal(0, noise_std, (C, T)).astype(np.float32)
        b = _sines(rng, C, T, fs, 3) + rng.normal(0, noise_std, (C, T)).astype(np.float32)
        c = int(y[i])
        m = (0.8 * np.sin(2 * np.pi * marker[c] * tt + rng.uniform(0, 2 * np.pi))).astype(np.float32)
        a[::2] += m
        b[1::2] += m
        x1[i] = zscore_window(a)
        x2[i] = zscore_window(coupling[c] * np.roll(a, lags[c], axis=1) + (1 - coupling[c]) * b)
    return torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(y.astype(np.int64))


def randn_windows(B: int, C: int, T: int, seed: int, num_classes: int = 3, device="cpu"):
    """Throughput-run batch: randn(seed) + per-window z-score, labels cyclic."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, B, C, T, generator=g)
    x = (x - x.mean(dim=(2, 3), keepdim=True)) / (x.std(dim=(2, 3), unbiased=False, keepdim=True) + 1e-8)
    labels = torch.arange(B) % num_classes
    return x[0].contiguous().to(device), x[1].contiguous().to(device), labels.to(device)


# ------------------------------------------------------------------------------------------------------
# Windowed shards.  The reference's DualEEGDataset re-reads BOTH full CSV recordings for every
# window it returns (1_Data/processed/dual_eeg_dataset.py:176-183); here each recording is read once, cut into the
# same windows (same enumeration order as `_prepare_windows`, :62-120) and stored raw as `[n, 2, C, T]` f32 `.npy`
# shards.  Batches are gathered from memory-mapped shards into pinned staging memory, copied to the device on a
# side stream, and normalised there by `eg_window_normalize` (the reference normalises on the host, :194-202).
# ------------------------------------------------------------------------------------------------------
import json  # noqa: E402
from pathlib import Path  # noqa: E402
from typing import Dict, Iterable, List, Optional, Sequence  # noqa: E402


def orient_recording(eeg: np.ndarray):
    """file contents -> ((C, T) f32, columns in the file).  Rows are channels unless there are more rows than columns
    (dual_eeg_dataset.py:132-139).  The column count AS STORED is what the reference's window enumeration uses
    (:86-92 reads one row and takes its width), so a transposed file yields no windows there — kept."""
    ncols = int(eeg.shape[1])
    if eeg.shape[0] > eeg.shape[1]:
        eeg = eeg.T
    return np.ascontiguousarray(eeg, dtype=np.float32), ncols


def load_recording(path):
    import pandas as pd
    return orient_recording(pd.read_csv(path, header=None).values)


def build_window_shards(items: Sequence[Dict], eeg_base_path, label2id: Dict[str, int], out_dir, window_size: int = 1024,
                        stride: int = 256, shard_windows: int = 4096, recordings: Optional[Dict[str, np.ndarray]] = None) -> Dict:
    """Cut every (player1, player2) pair of `items` into sliding windows and write shards + index.

    items        sequence of {'player1': name, 'player2': name, 'class': label} (the reference's metadata rows)
    recordings   optional name -> 2-D array holding a file's contents, instead of `<eeg_base_path>/<name>.csv`
    Window enumeration = the reference's: per item in order, `(min_len - W) // stride + 1` windows starting at
    `i * stride`, items with a missing file or `min_len < W` skipped; channels truncated to the pair's minimum."""
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    base = Path(eeg_base_path) if eeg_base_path is not None else None
    cache: Dict[str, np.ndarray] = {}

    def get(name):
        if recordings is not None:
            return orient_recording(recordings[name]) if name in recordings else None
        if name not in cache:
            p = base / f"{name}.csv"
            cache[name] = load_recording(p) if p.exists() else None
        return cache[name]

    meta: List[Dict] = []
    labels: List[int] = []
    shards: List[Dict] = []
    buf: List[np.ndarray] = []
    C_all = None

    def flush():
        if not buf:
            return
        arr = np.stack(buf)
        name = f"windows_{len(shards):05d}.npy"
        np.save(out / name, arr)
        shards.append({"file": name, "count": int(arr.shape[0])})
        buf.clear()

    for idx, item in enumerate(items):
        ra, rb = get(item["player1"]), get(item["player2"])
        if ra is None or rb is None:
            continue
        (a, na), (b, nb_) = ra, rb
        min_len = min(na, nb_)
        if min_len < window_size:
            continue
        C = min(a.shape[0], b.shape[0])
        if C_all is None:
            C_all = C
        elif C != C_all:
            raise ValueError(f"item {idx}: {C} channels, earlier items have {C_all}")
        for w in range((min_len - window_size) // stride + 1):
            s, e = w * stride, w * stride + window_size
            buf.append(np.stack([a[:C, s:e], b[:C, s:e]]))
            meta.append({"dataset_idx": idx, "start": s, "end": e, "player1": item["player1"], "player2": item["player2"],
                         "class": item["class"]})
            labels.append(int(label2id[item["class"]]))
            if len(buf) == shard_windows:
                flush()
        if recordings is None:
            cache.clear()
    flush()
    np.save(out / "labels.npy", np.asarray(labels, dtype=np.int64))
    index = {"window_size": window_size, "stride": stride, "channels": C_all, "count": len(meta), "shards": shards,
             "windows": meta, "label2id": label2id}
    (out / "index.json").write_text(json.dumps(index))
    return index


class WindowShards:
    """Per-rank batch iterator over windowed shards; yields dicts shaped like the reference's `collate_fn` output
    (dual_eeg_dataset.py:236-248): {'eeg1','eeg2': f32 [B,C,T] on `device`, 'labels': i64 [B], 'dataset_idx': list}.

    preprocessing=False -> per-window z-score (yaml default, :201-202); True -> CAR + per-channel z-score (:142-168; the
    reference's band-pass step there is a TODO that does nothing).  Rank r of `world` takes indices r::world of each
    epoch's (optionally shuffled) order, as `ddp.shard_indices` does for the synthetic runs."""

    def __init__(self, root, batch_size: int, device, rank: int = 0, world: int = 1, shuffle: bool = False, seed: int = 0,
                 preprocessing: bool = False, drop_last: bool = False):
        from . import _lib as L
        L.lib()  # the normalisation has no CPU fallback
        self.root = Path(root)
        self.index = json.loads((self.root / "index.json").read_text())
        self.labels = torch.from_numpy(np.load(self.root / "labels.npy"))
        self.maps = [np.load(self.root / s["file"], mmap_mode="r") for s in self.index["shards"]]
        counts = np.array([s["count"] for s in self.index["shards"]], dtype=np.int64)
        self.starts = np.concatenate([[0], np.cumsum(counts)])
        self.n, self.C, self.T = int(self.index["count"]), int(self.index["channels"]), int(self.index["window_size"])
        self.B, self.device, self.rank, self.world = batch_size, torch.device(device), rank, world
        self.shuffle, self.seed, self.mode, self.drop_last = shuffle, seed, 1 if preprocessing else 0, drop_last
        self.epoch = 0
        # double-buffered pinned staging + device raw buffers; copies run on their own stream
        self.pinned = [torch.empty(batch_size, 2, self.C, self.T, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.raw = [torch.empty(batch_size, 2, self.C, self.T, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(self.device)
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _order(self) -> np.ndarray:
        idx = np.arange(self.n)
        if self.shuffle:
            idx = np.random.default_rng(self.seed + self.epoch).permutation(self.n)
        return idx[self.rank::self.world]

    def __len__(self):
        if self.drop_last:      # the same count on every rank (a DDP step needs all of them)
            return (self.n // self.world) // self.B
        m = len(range(self.rank, self.n, self.world))
        return (m + self.B - 1) // self.B

    def _gather(self, ids: np.ndarray, slot: int):
        dst = self.pinned[slot].numpy()
        which = np.searchsorted(self.starts, ids, side="right") - 1
        for j, (i, s) in enumerate(zip(ids, which)):
            dst[j] = self.maps[s][i - self.starts[s]]

    def _stage(self, ids: np.ndarray, slot: int):
        self.consumed[slot].synchronize()          # the normalise kernel that last read raw[slot] has finished
        self._gather(ids, slot)
        with torch.cuda.stream(self.copy_stream):
            self.raw[slot][: len(ids)].copy_(self.pinned[slot][: len(ids)], non_blocking=True)
            self.copied[slot].record(self.copy_stream)

    def __iter__(self):
        from ._lib import call, ptr
        order = self._order()
        nb = len(self)
        chunks = [order[i * self.B:(i + 1) * self.B] for i in range(nb)]
        if not chunks:
            return
        self._stage(chunks[0], 0)
        for k, ids in enumerate(chunks):
            slot = k & 1
            if k + 1 < nb:
                self._stage(chunks[k + 1], slot ^ 1)      # host gather + H2D of the next batch overlap this batch's compute
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self.copied[slot])
            n = len(ids)
            eeg1 = torch.empty(n, self.C, self.T, dtype=torch.float32, device=self.device)
            eeg2 = torch.empty_like(eeg1)
            call("eg_window_normalize", ptr(self.raw[slot]), ptr(eeg1), ptr(eeg2), n, self.C, self.T, self.mode, cur.cuda_stream)
            self.consumed[slot].record(cur)
            yield {"eeg1": eeg1, "eeg2": eeg2, "labels": self.labels[ids].to(self.device, non_blocking=True),
                   "dataset_idx": [self.index["windows"][int(i)]["dataset_idx"] for i in ids]}
