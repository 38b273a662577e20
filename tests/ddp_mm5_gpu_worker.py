"""Worker for tests/test_gpu_ddp_mm5.py: one rank of a 2-rank data-parallel run of the multimodal logit-fusion step
(BASELINE configs[4]) on ONE GPU (gloo carries the collectives; the driver's multi-GPU runs use RCCL).  Real engines, real
MultimodalReducers on the side stream, real clip + per-group AdamW + loss-scaling kernels.
  phase A  f32, dropout off, two steps: the replicas equal a single-process run on the global batch;
  phase B  fp16 with dynamic loss scaling, three steps, rank 1 ALONE overflows in step 1: both ranks skip that step, back the
           scale off, and hold bit-identical parameters afterwards."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.distributed as dist

from eyegaze_multimodal_amd import DualEEGTransformer
from eyegaze_multimodal_amd.ddp import shard_indices
from eyegaze_multimodal_amd.fuzzy_gating_fusion import FuzzyGatingFusion
from eyegaze_multimodal_amd.image_encoder import GazeCNNEncoder
from eyegaze_multimodal_amd.train_multimodal_fuzzy_fusion import MultimodalFusionModel, MultimodalTrainer, synth_multimodal

KW = dict(in_channels=8, num_classes=3, max_len=256, use_spectrogram=True, use_ibs=False, use_cross_attention=True,
          d_model=64, num_layers=2, num_heads=2, d_ff=128)
HP = dict(encoder_lr=2e-4, fusion_lr=2e-3, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=2, total_steps=10)


def make(dtype, seed):
    torch.manual_seed(seed)
    return MultimodalFusionModel(GazeCNNEncoder(num_classes=3, d_model=64, compute_dtype=dtype),
                                 DualEEGTransformer(**KW, compute_dtype=dtype), FuzzyGatingFusion(num_classes=3, mode="full"))


def flats(tr, model):
    return [model.eeg_encoder._flat.flat, model.gaze_encoder._flat.flat, tr.fus.flat]


def identical(ts):
    ok = True
    for t_ in ts:
        other = t_.clone()
        dist.broadcast(other, src=0)
        ok = ok and bool(torch.equal(other, t_))
    return ok


def main():
    out = Path(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    GB, F_, W_ = 16, 64, 16
    batch = synth_multimodal(GB, 8, 1024, F_, W_, 3, seed=5)
    mine = list(shard_indices(GB, rank, world))
    d = lambda t_: t_.to(dev)
    res = {"rank": rank}

    # ---- phase A: f32, deterministic, against the single-process global-batch run ----
    model = make("f32", 100 + rank)                      # different init per rank: the trainer's broadcast must fix that
    tr = MultimodalTrainer(model, dev, **HP)
    for step in range(2):
        o = tr.train_step(*(d(t_[mine]) for t_ in batch), dropout=False)
    torch.cuda.synchronize()
    res["A_active"] = bool(tr.red is not None and tr.red.active)
    res["A_identical"] = identical(flats(tr, model))
    if rank == 0:
        ref = make("f32", 100)                           # rank 0's initialisation = what the broadcast distributed
        tr1 = MultimodalTrainer(ref, dev, **HP)
        tr1._reducers = lambda *a: None                  # single-process reference: no exchange
        ref.eeg_encoder._flat.ensure(dev)
        ref.gaze_encoder._flat.ensure(dev)
        start = [f.clone() for f in flats(tr1, ref)]
        for step in range(2):
            tr1.train_step(*(d(t_) for t_ in batch), dropout=False)
        torch.cuda.synchronize()
        num = sum(float((a - b).double().pow(2).sum()) for a, b in zip(flats(tr, model), flats(tr1, ref)))
        den = sum(float((a - b).double().pow(2).sum()) for a, b in zip(flats(tr1, ref), start))
        res["A_param_rel_err_after_2_steps"] = (num / den) ** 0.5
        res["A_moved"] = den ** 0.5

    # ---- phase B: fp16, rank 1 alone overflows in step 1 ----
    model = make("fp16", 7)
    tr = MultimodalTrainer(model, dev, **HP)
    scales, skipped = [], []
    for step in range(3):
        b = [d(t_[mine]).clone() for t_ in batch]
        if step == 1 and rank == 1:
            b[0] = b[0] * 1e30                           # fp16 image rows overflow -> non-finite logits and gradients on this rank
        tr.train_step(*b)
        st = model.eeg_encoder.engine(len(mine), 1024, dev).read_state()
        scales.append(st.loss_scale)
        skipped.append(int(st.skipped))
    torch.cuda.synchronize()
    st = model.eeg_encoder.engine(len(mine), 1024, dev).read_state()
    res.update(B_identical=identical(flats(tr, model)), B_scales=scales, B_skipped=skipped, B_opt_steps=int(st.opt_steps),
               B_finite=bool(all(torch.isfinite(f).all() for f in flats(tr, model))))
    (out / f"rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
