"""CPU-only, world_size 2 over gloo: the data-parallel gradient exchange of ddp.py.
  * buckets tile the flat gradient buffer exactly and follow the backward's segment order;
  * summed all-reduce * grad_scale == the gradient of the global batch (checked with the CPU oracle as the
    per-rank gradient provider: the HIP engine cannot run without a GPU);
  * rank r takes samples r::world."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eyegaze_multimodal_amd import DualEEGTransformer
from eyegaze_multimodal_amd.ddp import GradAllReducer, bucket_ranges, shard_indices
from oracle import dual_eeg_oracle as O

KW = dict(in_channels=8, max_len=256, num_classes=3, d_model=64, num_layers=2, num_heads=2, d_ff=128,
          use_spectrogram=False, use_ibs=False, use_cross_attention=True)
SEGMENTS = ["heads", "cross", "encoder.norm", "layer1", "layer0", "tokens", "conv1", "frontend"]  # order Engine.backward emits


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(0)
    model = DualEEGTransformer(**KW)
    fp = model._flat
    fp.ensure(torch.device("cpu"))
    cfg = O.ModelCfg(**KW)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    Bg = 4
    x1, x2 = torch.randn(Bg, 8, 1024, generator=g), torch.randn(Bg, 8, 1024, generator=g)
    labels = torch.tensor([0, 1, 2, 1])
    mine = list(shard_indices(Bg, rank, world))

    def grads(idx):
        P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out = O.forward(x1[idx], x2[idx], P, cfg, labels[idx])
        out["loss_ce"].backward()
        return P
    P = grads(mine)
    for n, p in zip(fp.names, fp.params):
        o = fp.offsets[n]
        fp.grad[o:o + p.numel()] = P[n].grad.reshape(-1)
    ranges = bucket_ranges(fp.names, fp.offsets, fp.total, KW["num_layers"], True)
    red = GradAllReducer(fp.grad, ranges)
    for s in SEGMENTS:
        red.on_segment(s)
    red.finish()
    got = fp.grad * red.grad_scale
    if rank == 0:
        Pfull = grads(list(range(Bg)))
        ref = torch.zeros_like(got)
        for n, p in zip(fp.names, fp.params):
            o = fp.offsets[n]
            ref[o:o + p.numel()] = Pfull[n].grad.reshape(-1)
        q.put((float((got - ref).abs().max()), float(ref.abs().max()), sorted(ranges.items(), key=lambda kv: kv[1]), fp.total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_global_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, ranges, total = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err < 1e-5 * max(1.0, scale), (err, scale)
    # buckets tile [0, total) with no gap / overlap
    pos = 0
    for name, (b, e) in ranges:
        assert b == pos and e > b, (name, b, e, pos)
        pos = e
    assert pos == total


def test_bucket_order_matches_registration_order():
    torch.manual_seed(0)
    model = DualEEGTransformer(in_channels=8, max_len=256, use_spectrogram=True, use_ibs=True)
    fp = model._flat
    r = bucket_ranges(fp.names, fp.offsets, fp.total, 6, True)
    assert r["frontend"][0] == 0 and r["heads"][1] == fp.total
    # the front end is three buckets in registration order: cls_token + conv-0 (reduced last), conv-1, then every token generator /
    # extra head / position parameter registered before the encoder
    for n in fp.names:
        if n.startswith(("temporal_conv.convs.0", "cls_token")):
            assert r["frontend"][0] <= fp.offsets[n] < r["frontend"][1], n
        if n.startswith("temporal_conv.convs.1"):
            assert r["conv1"][0] <= fp.offsets[n] < r["conv1"][1], n
        if n.startswith(("spectrogram", "ibs_", "pos_embed")):
            assert r["tokens"][0] <= fp.offsets[n] < r["tokens"][1], n
    assert r["frontend"][1] == r["conv1"][0] and r["conv1"][1] == r["tokens"][0] and r["tokens"][1] == r["layer0"][0]
    assert r["layer5"][0] == fp.offsets["encoder.layers.5.mha.q_proj.weight"]


def test_shard_indices_partition():
    idx = [list(shard_indices(2048, r, 8)) for r in range(8)]
    assert all(len(i) == 256 for i in idx)
    assert sorted(sum(idx, [])) == list(range(2048))
