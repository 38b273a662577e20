"""Optional token families in front of the temporal tokens: spectrogram tokens (D:40-135) and
inter-stream synchrony (IBS) tokens (D:473-911).  Called by the engine through the module hooks."""
from __future__ import annotations

from . import _lib as L


def _unsupported(model):
    c = model.cfg
    if c.use_spectrogram or c.use_ibs:
        raise L.EgError("spectrogram / IBS token kernels are not built into this library version")


def pack(model, eng):
    _unsupported(model)


def forward(model, eng, eeg1, eeg2, train):
    _unsupported(model)


def backward(model, eng, dseq):
    _unsupported(model)
