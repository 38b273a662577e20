"""Data-parallel gradient exchange (no reference counterpart): one process per GPU, gradients
summed with torch.distributed all-reduce (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The flat gradient buffer is cut into contiguous buckets that follow the order in which the backward pass
finishes them (heads -> cross-attention -> encoder layers L-1..0 -> tokens -> conv-1 -> front-end), so each bucket's all-reduce
is enqueued on a side stream as soon as its kernels are queued and overlaps the rest of the backward.
The sum is turned into the mean by `grad_scale = 1/world_size` inside the clip/AdamW kernels.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def bucket_ranges(names: List[str], offsets: Dict[str, int], total: int, num_layers: int,
                  use_cross: bool) -> Dict[str, Tuple[int, int]]:
    """segment name (as emitted by Engine.backward) -> [begin, end) in floats of the flat buffer."""
    def first(prefix):
        return min(offsets[n] for n in names if n.startswith(prefix))
    # front end, in registration order: cls_token, conv-0 | conv-1 (6.5 MB of the 7 MB) | token generators, extra heads, positions.
    # Engine.backward releases "tokens" before the strided-conv backward starts and "conv1" right after conv-1's weight gradient
    # -- ahead of its four backward-data phases and conv-0's gradient -- so only "frontend" (cls_token + conv-0, 0.2 MB) is left to
    # reduce once backward has ended.
    marks = [("frontend", 0)]
    c1 = [n for n in names if n.startswith("temporal_conv.convs.1.")]
    if c1 and first("temporal_conv.convs.1.") > 0:
        after = [offsets[n] for n in names if offsets[n] > max(offsets[m] for m in c1)]
        if after and min(after) < first("encoder.layers.0."):
            marks += [("conv1", first("temporal_conv.convs.1.")), ("tokens", min(after))]
    for l in range(num_layers):
        marks.append((f"layer{l}", first(f"encoder.layers.{l}.")))
    marks.append(("encoder.norm", first("encoder.norm.")))
    if use_cross:
        marks.append(("cross", first("cross_attn.")))
    marks.append(("heads", first("symmetric_fusion.")))
    out = {}
    for i, (name, beg) in enumerate(marks):
        end = marks[i + 1][1] if i + 1 < len(marks) else total
        out[name] = (beg, end)
    return out


class GradAllReducer:
    def __init__(self, flat_grad: torch.Tensor, ranges: Dict[str, Tuple[int, int]], group=None, force: bool = False):
        self.g, self.ranges, self.group = flat_grad, ranges, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()   # exercise the collective path even with one rank (rehearsal)
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(flat_grad.device) if self.cuda else None
        self.works = []

    def on_segment(self, name: str):
        """Called by Engine.backward right after the kernels producing `name`'s gradients were enqueued."""
        if (self.world == 1 and not self.force) or name not in self.ranges:
            return
        b, e = self.ranges[name]
        if e <= b:
            return
        view = self.g[b:e]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.g.device))
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Makes the compute stream wait for every bucket (no host sync on GPU)."""
        for w in self.works:
            w.wait()
        self.works = []
        if self.cuda:
            torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def broadcast_params(flat: torch.Tensor, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=0, group=group)


def shard_indices(n: int, rank: int, world: int) -> range:
    """rank r takes samples r::world of each global batch (SURVEY §8e partitioning)."""
    return range(rank, n, world)
