"""Worker for tests/test_gpu_ddp.py: one rank of a 2-rank data-parallel run on ONE GPU (gloo carries the collectives; the
driver's multi-GPU runs use RCCL).  Real engine, real bucketed reducer on the side stream, real clip + AdamW kernels."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.distributed as dist

from eyegaze_multimodal_amd import DualEEGTransformer, HipAdamW
from eyegaze_multimodal_amd.data import randn_windows
from eyegaze_multimodal_amd.ddp import GradAllReducer, broadcast_params, bucket_ranges, shard_indices


def main():
    out = Path(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    kw = dict(in_channels=8, num_classes=3, max_len=256, num_layers=2, use_spectrogram=False, use_ibs=False, use_cross_attention=True)
    torch.manual_seed(100 + rank)                      # different init per rank: broadcast must fix that
    model = DualEEGTransformer(**kw, compute_dtype="f32").to(dev)
    model.eval()                                       # deterministic step (no dropout): comparable with one big batch
    GB = 16
    x1, x2, y = randn_windows(GB, 8, 1024, seed=5, num_classes=3)
    mine = list(shard_indices(GB, rank, world))
    eng = model.engine(len(mine), 1024, dev)
    fp = model._flat
    broadcast_params(fp.flat)
    start = fp.flat.clone()
    opt = HipAdamW(model, lr=1e-3, weight_decay=0.01)
    red = GradAllReducer(fp.grad, bucket_ranges(fp.names, fp.offsets, fp.total, 2, True))
    one = torch.ones(1, device=dev)
    seen = []
    for step in range(2):
        opt.begin_step(eng, seed=step, grad_scale=red.grad_scale)
        eng.forward(x1[mine].to(dev), x2[mine].to(dev), y[mine].to(dev), train=False)
        eng.backward(gloss=one, on_segment=lambda n: (seen.append(n), red.on_segment(n)))
        red.finish()
        if step == 0:
            torch.cuda.synchronize()
            g0 = (fp.grad * red.grad_scale).clone()    # mean over ranks of the per-rank mean-loss gradients
        opt.step(eng)
    torch.cuda.synchronize()
    # all ranks must hold identical parameters
    mineflat = fp.flat.clone()
    other = mineflat.clone()
    dist.broadcast(other, src=0)
    same = bool(torch.equal(other, mineflat))
    res = {"rank": rank, "same_params_as_rank0": same, "segments": seen[: len(seen) // 2], "moved": float((mineflat - start).abs().max())}
    if rank == 0:
        # single-process reference: the whole global batch in one engine, same starting parameters
        ref = DualEEGTransformer(**kw, compute_dtype="f32").to(dev)
        ref.eval()
        ref._flat.ensure(dev)
        ref._flat.flat.copy_(start)
        e2 = ref.engine(GB, 1024, dev)
        e2.set_state(seed=0, lr=0.0, step=1)
        e2.forward(x1.to(dev), x2.to(dev), y.to(dev), train=False)
        e2.backward(gloss=one)
        torch.cuda.synchronize()
        gref = ref._flat.grad
        res["grad_rel_err"] = float((g0 - gref).norm() / gref.norm())
        o2 = HipAdamW(ref, lr=1e-3, weight_decay=0.01)
        for step in range(2):
            o2.begin_step(e2, seed=step)
            e2.forward(x1.to(dev), x2.to(dev), y.to(dev), train=False)
            e2.backward(gloss=one)
            o2.step(e2)
        torch.cuda.synchronize()
        res["param_rel_err_after_2_steps"] = float((ref._flat.flat - mineflat).norm() / (mineflat - start).norm())
    (out / f"rank{rank}.json").write_text(json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
