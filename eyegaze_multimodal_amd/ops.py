"""PyTorch operator registration of the HIP path (torch.library): the module seam of the reference
(3_Models/backbones/dual_eeg_transformer.py:1110-1253, called from 4_Experiments/scripts/train_art.py:178) is served by
REGISTERED operators, visible to the dispatcher / profiler / torch.compile graph capture as `eyegaze::*`:

  eyegaze::dual_eeg_forward(Tensor eeg1, Tensor eeg2, Tensor? labels, Tensor[] params, int handle, bool train) -> Tensor[]
      the whole DualEEGTransformer.forward on the MI355X engine; outputs in the fixed order OUTPUT_KEYS (absent entries are
      empty tensors); autograd formula registered with torch.library.register_autograd (gradients for `params` come from the
      HIP backward over the flat gradient buffer)
  eyegaze::dual_eeg_train_step(Tensor eeg1, Tensor eeg2, Tensor labels, Tensor flat_params, Tensor(a!) flat_grads,
                               float lambda_ibs_cls, int handle) -> Tensor[]
      the native training step without an autograd graph (train_art.py:175-220): forward in train mode + backward of
      loss_ce + lambda_ibs_cls * loss_ibs_cls into the flat gradient buffer; returns [loss_ce, loss_ibs_cls]
  eyegaze::clip_adamw_step(Tensor flat_params, Tensor flat_grads, Tensor exp_avg, Tensor exp_avg_sq, int handle) -> ()
      clip_grad_norm_(max_norm) + AdamW of train_art.py:221-222 on the flat buffers (mutates all but the gradients)

`handle` names the Python-side engine owner (a module instance): operators carry tensors and scalars only.
Both operators are CUDA(HIP)-only: there is no CPU kernel behind them, calling them with host tensors raises.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Sequence

import torch

OUTPUT_KEYS = ("logits", "cls1", "cls2", "ibs_logits", "ibs_token", "loss_ce", "loss_ibs_cls")
_OWNERS: Dict[int, "weakref.ReferenceType"] = {}
_NEXT = [1]


def register_owner(obj) -> int:
    h = _NEXT[0]
    _NEXT[0] += 1
    _OWNERS[h] = weakref.ref(obj)
    return h


def owner(handle: int):
    ref = _OWNERS.get(int(handle))
    obj = ref() if ref is not None else None
    if obj is None:
        raise RuntimeError(f"eyegaze operator called with a stale module handle {handle}")
    return obj


@torch.library.custom_op("eyegaze::dual_eeg_forward", mutates_args=(), device_types="cuda")
def dual_eeg_forward(eeg1: torch.Tensor, eeg2: torch.Tensor, labels: Optional[torch.Tensor], params: Sequence[torch.Tensor],
                     handle: int, train: bool) -> List[torch.Tensor]:
    model = owner(handle)
    eng = model.engine(eeg1.shape[0], eeg1.shape[2], eeg1.device)
    out = model._run_forward(eng, eeg1, eeg2, labels, train)
    return [out[k] if k in out else eeg1.new_empty(0) for k in OUTPUT_KEYS]   # (outputs of an operator may not alias each other)


@dual_eeg_forward.register_fake
def _(eeg1, eeg2, labels, params, handle, train):
    model = owner(handle)
    B, d, nc = eeg1.shape[0], model.cfg.d_model, model.cfg.num_classes
    e = lambda: eeg1.new_empty(0)
    ibs, lab = model.cfg.use_ibs, labels is not None
    return [eeg1.new_empty(B, nc), eeg1.new_empty(B, d), eeg1.new_empty(B, d),
            eeg1.new_empty(B, nc) if ibs else e(), eeg1.new_empty(B, d) if ibs else e(),
            eeg1.new_empty(()) if lab else e(), eeg1.new_empty(()) if (lab and ibs) else e()]


def _setup(ctx, inputs, output):
    eeg1, eeg2, labels, params, handle, train = inputs
    model = owner(handle)
    ctx.handle, ctx.fwd_id, ctx.nparams = handle, model._fwd_count, len(params)
    ctx.shape = (eeg1.shape[0], eeg1.shape[2], eeg1.device)
    ctx.needs = [p.requires_grad for p in params]


def _backward(ctx, gouts):
    model = owner(ctx.handle)
    grads = model._run_backward(ctx.shape, ctx.fwd_id, dict(zip(OUTPUT_KEYS, gouts)))
    return None, None, None, [g if need else None for g, need in zip(grads, ctx.needs)], None, None


torch.library.register_autograd("eyegaze::dual_eeg_forward", _backward, setup_context=_setup)


@torch.library.custom_op("eyegaze::dual_eeg_train_step", mutates_args=("flat_grads",), device_types="cuda")
def dual_eeg_train_step(eeg1: torch.Tensor, eeg2: torch.Tensor, labels: torch.Tensor, flat_params: torch.Tensor,
                        flat_grads: torch.Tensor, lambda_ibs_cls: float, handle: int) -> List[torch.Tensor]:
    model = owner(handle)
    eng = model.engine(eeg1.shape[0], eeg1.shape[2], eeg1.device)
    fp = model._flat
    if flat_params.data_ptr() != fp.flat.data_ptr() or flat_grads.data_ptr() != fp.grad.data_ptr():
        raise RuntimeError("eyegaze::dual_eeg_train_step works on the module's own flat parameter / gradient buffers")
    eng.forward(eeg1, eeg2, labels, train=True)
    one = torch.ones(1, device=eeg1.device)
    ibs = model.cfg.use_ibs
    eng.backward(gloss=one, gloss_ibs=(one * lambda_ibs_cls if ibs else None), on_segment=getattr(model, "_on_segment", None))
    return [eng.a["loss"].reshape(()).clone(), (eng.a["ibs_loss"].reshape(()).clone() if ibs else eeg1.new_zeros(()))]


@dual_eeg_train_step.register_fake
def _(eeg1, eeg2, labels, flat_params, flat_grads, lambda_ibs_cls, handle):
    return [eeg1.new_empty(()), eeg1.new_empty(())]


@torch.library.custom_op("eyegaze::clip_adamw_step", mutates_args=("flat_params", "exp_avg", "exp_avg_sq"), device_types="cuda")
def clip_adamw_step(flat_params: torch.Tensor, flat_grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                    handle: int) -> None:
    opt = owner(handle)
    opt._native_step(flat_params, flat_grads, exp_avg, exp_avg_sq)
