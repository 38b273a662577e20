"""
TEST INFRASTRUCTURE — golden vectors for the LOOP of the reference's multimodal logit-fusion script
(4_Experiments/scripts/train_multimodal_fuzzy_fusion.py).  Runs ONLY in the build container.

The script cannot be imported as a module here: its top level loads early_fusion_vit.py (needs `timm`) and
multimodal_dataset.py (needs `torchvision`), both absent offline -- and they stay absent, nothing stands in for them.  The loop
itself depends on neither.  This generator therefore parses the reference file with `ast` and executes ONLY these definitions
of it, unmodified, in a namespace that holds the very imports the script makes for them (torch, numpy, sklearn.metrics, tqdm,
LambdaLR, AdamW, GradScaler, autocast -- all present here):

    MultimodalFusionModel (:106-179)   get_linear_warmup_cosine_scheduler (:197-214)   compute_metrics (:217-233)
    train_one_epoch (:395-543)         and, out of train(), the statements that build encoder_lr / fusion_lr / param_groups /
                                       optimizer / steps_per_epoch / scheduler (:727-746)

What the fixture pins: the loss composition, clip, optimizer groups, per-step scheduler order and the epoch metrics of the
reference loop, run by the reference's own code.  What it does not pin: the ViT image branch (replaced by the in-tree 2-D CNN,
the image branch this repo defines for configs[4]) and the GradScaler branch (`use_amp` needs CUDA: the fp32 branch :474-502 is
the one executed; GradScaler itself is torch's published algorithm).
Every nn.Dropout instance gets p = 0 so that the train-mode loop is deterministic (all dropouts in the model files are modules).

Usage:  python oracle/make_golden_mm_loop.py     -> tests/golden/mm_loop.npz   (data only)
"""
from __future__ import annotations

import ast
import copy
import logging
import sys
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle.dual_eeg_oracle import ModelCfg, synthetic_state_dict  # noqa: E402
from oracle.make_golden import REF, _load  # noqa: E402

SCRIPT = REF / "4_Experiments" / "scripts" / "train_multimodal_fuzzy_fusion.py"
EEG_KW = dict(in_channels=8, num_classes=3, max_len=256, use_spectrogram=True, use_ibs=False, use_cross_attention=True,
              d_model=64, num_layers=2, num_heads=2, d_ff=128)
WEIGHT_SEED, GAZE_SEED, DATA_SEED = 777, 778, 31
N, BATCH, EPOCHS, WARMUP_EPOCHS = 16, 4, 2, 1
CONFIG = {"training": {"encoder_learning_rate": 2e-4, "fusion_learning_rate": 2e-3, "weight_decay": 0.01, "max_grad_norm": 1.0,
                       "lambda_aux_img": 0.3, "lambda_aux_eeg": 0.3, "lambda_reg": 0.1, "warmup_epochs": WARMUP_EPOCHS,
                       "epochs": 3, "fp16": False},
          "fusion": {"mode": "full", "temp_reg_min": 0.5, "temp_reg_max": 5.0, "eps_temp": 0.1},
          "data": {"class_names": ["Single", "Competition", "Cooperation"]}}


def reference_definitions():
    """executes the loop's definitions of the reference script (see the module docstring); returns the namespace and train()'s
    optimizer / scheduler statements as a code object"""
    from sklearn.metrics import accuracy_score, confusion_matrix, precision_recall_fscore_support
    from torch.amp import GradScaler, autocast
    from torch.optim import AdamW
    from torch.optim.lr_scheduler import LambdaLR
    from torch.utils.data import DataLoader
    from tqdm import tqdm
    ns = dict(torch=torch, nn=nn, F=F, np=np, Dict=Dict, Optional=Optional, Tuple=Tuple, DataLoader=DataLoader, AdamW=AdamW,
              LambdaLR=LambdaLR, GradScaler=GradScaler, autocast=autocast, accuracy_score=accuracy_score,
              precision_recall_fscore_support=precision_recall_fscore_support, confusion_matrix=confusion_matrix,
              tqdm=(lambda it, **kw: tqdm(it, disable=True, **kw)), logger=logging.getLogger("reference_loop"), __name__="reference_loop")
    tree = ast.parse(SCRIPT.read_text(encoding="utf-8"))
    want = {"MultimodalFusionModel", "get_linear_warmup_cosine_scheduler", "compute_metrics", "train_one_epoch"}
    got = set()
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in want:
            exec(compile(ast.Module(body=[node], type_ignores=[]), str(SCRIPT), "exec"), ns)
            got.add(node.name)
    assert got == want, want - got
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "train")
    targets = ("encoder_lr", "fusion_lr", "param_groups", "optimizer", "steps_per_epoch", "scheduler")
    stmts = [s for s in main.body if isinstance(s, ast.Assign) and len(s.targets) == 1 and isinstance(s.targets[0], ast.Name)
             and s.targets[0].id in targets]
    assert [s.targets[0].id for s in stmts] == list(targets), [s.targets[0].id for s in stmts]
    return ns, compile(ast.Module(body=stmts, type_ignores=[]), str(SCRIPT), "exec")


class GazeTorch(nn.Module):
    """the in-tree 2-D CNN image branch (eyegaze_multimodal_amd/image_encoder.py: parameter containers `cnn`, `head`) as a plain
    torch module: forward = oracle.multimodal_oracle.image_logits"""

    def __init__(self, d_model: int, num_classes: int):
        super().__init__()
        from eyegaze_multimodal_amd.image_encoder import _CNN
        self.cnn = _CNN(d_model)
        self.head = nn.Linear(2 * d_model, num_classes)

    def forward(self, img1, img2):
        from oracle.multimodal_oracle import image_logits
        return image_logits(self, img1, img2)


def build_models(model_mod, fz_mod):
    cfg = ModelCfg(**EEG_KW)
    eeg = model_mod.DualEEGTransformer(**EEG_KW)
    eeg.load_state_dict(synthetic_state_dict(cfg, WEIGHT_SEED))
    gaze = GazeTorch(EEG_KW["d_model"], 3)
    from oracle.multimodal_oracle import synthetic_gaze_state
    gaze.load_state_dict(synthetic_gaze_state(gaze, GAZE_SEED))
    fusion = fz_mod.FuzzyGatingFusion(num_classes=3, mode="full", eps_temp=0.1)
    return gaze, eeg, fusion


def batches():
    from eyegaze_multimodal_amd.train_multimodal_fuzzy_fusion import synth_multimodal
    img1, img2, x1, x2, y = synth_multimodal(N, 8, 1024, 64, 16, 3, seed=DATA_SEED)
    return [{"img1": img1[i:i + BATCH], "img2": img2[i:i + BATCH], "eeg1": x1[i:i + BATCH], "eeg2": x2[i:i + BATCH],
             "labels": y[i:i + BATCH]} for i in range(0, N, BATCH)]


def sample(t: torch.Tensor, k: int = 16):
    return t.detach().reshape(-1)[:k].double().numpy()


def main():
    torch.set_num_threads(4)
    sys.path.insert(0, str(REF / "3_Models" / "backbones"))
    model_mod = _load("dual_eeg_transformer", REF / "3_Models" / "backbones" / "dual_eeg_transformer.py")
    fz_mod = _load("fuzzy_gating_fusion", REF / "3_Models" / "fusion" / "fuzzy_gating_fusion.py")
    ns, opt_code = reference_definitions()
    gaze, eeg, fusion = build_models(model_mod, fz_mod)
    model = ns["MultimodalFusionModel"](gaze_encoder=gaze, eeg_encoder=eeg, fusion_module=fusion)
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    train_loader = batches()
    config = copy.deepcopy(CONFIG)
    scope = dict(ns, model=model, config=config, train_loader=train_loader)
    exec(opt_code, scope)                     # the reference's own param_groups / AdamW / scheduler construction (:727-746)
    optimizer, scheduler = scope["optimizer"], scope["scheduler"]
    lrs, gnorm = [], []

    def record(opt, args, kwargs):            # what each optimizer.step() of the reference loop runs with
        lrs.append([g["lr"] for g in opt.param_groups])
        gnorm.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for g in opt.param_groups for p in g["params"]
                                          if p.grad is not None))))
    optimizer.register_step_pre_hook(record)
    blob: Dict[str, np.ndarray] = {}
    for epoch in range(EPOCHS):
        metrics = ns["train_one_epoch"](model, train_loader, optimizer, scheduler, None, torch.device("cpu"), epoch, config,
                                        use_amp=False)
        for k in ("loss", "loss_ce", "loss_aux_img", "loss_aux_eeg", "loss_reg", "alpha_mean", "alpha_std", "accuracy", "f1"):
            blob[f"epoch{epoch}/{k}"] = np.float64(metrics[k])
    blob["lrs"] = np.asarray(lrs, dtype=np.float64)                # [steps, 3 groups]: the lr each optimizer.step() ran with
    blob["clipped_grad_norm"] = np.asarray(gnorm, dtype=np.float64)   # global L2 norm at optimizer.step() (after the clip)
    final = model.state_dict()
    names, dn, fn, ds = [], [], [], []
    for k, v in final.items():
        if not v.dtype.is_floating_point or k.endswith("spectrogram_generator.window") or k.endswith("c_reliable"):
            continue
        names.append(k)
        dn.append(float((v - init[k]).double().norm()))
        fn.append(float(v.double().norm()))
        ds.append(sample(v - init[k]))
    blob["param_names"] = np.asarray(names)
    blob["delta_norm"] = np.asarray(dn)
    blob["final_norm"] = np.asarray(fn)
    blob["delta_head"] = np.stack([np.pad(d, (0, 16 - len(d))) for d in ds])     # first 16 elements of (final - initial) per tensor
    for k, v in final.items():
        if k.startswith("fusion.") and v.dtype.is_floating_point:
            blob["final/" + k] = v.detach().double().numpy()
    blob["meta"] = np.asarray(repr(dict(eeg_kw=EEG_KW, weight_seed=WEIGHT_SEED, gaze_seed=GAZE_SEED, data_seed=DATA_SEED, n=N,
                                        batch=BATCH, epochs=EPOCHS, config=CONFIG)))
    out = REPO / "tests" / "golden" / "mm_loop.npz"
    np.savez_compressed(out, **blob)
    print(f"wrote {out} ({out.stat().st_size} bytes); steps={len(lrs)} lrs[0]={lrs[0]} lrs[-1]={lrs[-1]}")
    for e in range(EPOCHS):
        print({k.split('/')[1]: float(v) for k, v in blob.items() if k.startswith(f"epoch{e}/")})


if __name__ == "__main__":
    main()
