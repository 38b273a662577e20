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
