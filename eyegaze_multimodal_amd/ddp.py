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


def coalesce_groups(ranges: Dict[str, Tuple[int, int]], num_layers: int) -> List[List[str]]:
    """Segments that travel as ONE collective.  Engine.backward finishes the encoder's weight gradients in two pieces (layers
    L-1 .. L/2 with the cross block, then layers L/2-1 .. 0), so finer buckets than that only add collectives: twelve per step cost
    0.36 ms with ONE rank before a byte crossed xGMI (round-2 VERDICT item 6).  Four remain, each contiguous in the flat buffer and
    released when its last segment is queued:
      upper   = layers L/2 .. L-1 | encoder.norm | cross | heads      (released at the split layer: reduces under the lower layers)
      lower   = tokens | layers 0 .. L/2-1                            (released after the positional / token-generator gradients)
      conv1   = conv-1's 6.5 MB                                        (reduces under conv-1's backward-data phases and conv-0)
      frontend = cls_token + conv-0 (0.2 MB), once backward has ended"""
    h = num_layers // 2
    upper = [f"layer{l}" for l in range(h, num_layers)] + ["encoder.norm", "cross", "heads"]
    lower = ["tokens"] + [f"layer{l}" for l in range(0, h)]
    groups = [[n for n in g if n in ranges] for g in (upper, lower, ["conv1"], ["frontend"])]
    groups = [g for g in groups if g]
    for g in groups:                        # a group must be one contiguous range of the flat buffer
        spans = sorted(ranges[n] for n in g)
        if any(spans[i][1] != spans[i + 1][0] for i in range(len(spans) - 1)):
            return [[n] for n in ranges]    # unexpected registration order: fall back to one collective per segment
    rest = [n for n in ranges if not any(n in g for g in groups)]
    return groups + [[n] for n in rest]


class GradAllReducer:
    def __init__(self, flat_grad: torch.Tensor, ranges: Dict[str, Tuple[int, int]], group=None, force: bool = False,
                 comm_stream=None, groups: Optional[List[List[str]]] = None):
        """comm_stream: share one side stream between the reducers of several flat buffers (the multimodal model has three).
        groups: lists of segment names reduced by one collective, released when the last of them arrives (coalesce_groups)."""
        self.g, self.ranges, self.group = flat_grad, ranges, group
        self.groups = groups or [[n] for n in ranges]
        self._group_of = {n: i for i, g in enumerate(self.groups) for n in g}
        self._arrived = [0] * len(self.groups)
        self.collectives = 0                 # all-reduces issued so far (tests / bench report it)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()   # exercise the collective path even with one rank (rehearsal)
        self.cuda = flat_grad.is_cuda
        self.comm_stream = (comm_stream or torch.cuda.Stream(flat_grad.device)) if self.cuda else None
        self.works = []

    def on_segment(self, name: str):
        """Called by Engine.backward right after the kernels producing `name`'s gradients were enqueued."""
        if (self.world == 1 and not self.force) or name not in self.ranges:
            return
        gi = self._group_of[name]
        self._arrived[gi] += 1
        if self._arrived[gi] < len(self.groups[gi]):
            return                           # the group's collective waits for its last segment
        self._arrived[gi] = 0
        b = min(self.ranges[n][0] for n in self.groups[gi])
        e = max(self.ranges[n][1] for n in self.groups[gi])
        if e <= b:
            return
        self.collectives += 1
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


class MultimodalReducers:
    """The gradient exchange of the three-parameter-set multimodal model (train_multimodal_fuzzy_fusion.py: gaze encoder, EEG
    encoder, fusion scalars -- one optimizer over three param groups in the reference, :727-737): one GradAllReducer per flat
    gradient buffer on ONE shared side stream, released in the order the step finishes them -- fusion scalars (ready after the
    [B, K] autograd graph), the image branch (ready after its short backward, reduces under the EEG backward), then the EEG
    encoder's buckets as Engine.backward emits them.  `sync_flag` makes the fp16 overflow decision collective: every rank's
    eg_clip_coef sees the same reduced gradients, so the flags already agree by construction; the MAX all-reduce of the one
    found_inf word keeps the replicas in lock-step even if a backend ever returned rank-dependent roundings."""

    def __init__(self, eeg_fp, gaze_fp, fus_fp, num_layers: int, use_cross: bool, group=None, force: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or force)
        whole = lambda fp: {"all": (0, fp.total)}
        self.eeg = self.gaze = self.fusion = None
        stream = None
        if eeg_fp is not None and eeg_fp.grad is not None:
            rng = bucket_ranges(eeg_fp.names, eeg_fp.offsets, eeg_fp.total, num_layers, use_cross)
            self.eeg = GradAllReducer(eeg_fp.grad, rng, group, force, groups=coalesce_groups(rng, num_layers))
            stream = self.eeg.comm_stream
        if gaze_fp is not None and gaze_fp.grad is not None:
            self.gaze = GradAllReducer(gaze_fp.grad, whole(gaze_fp), group, force, comm_stream=stream)
            stream = stream or self.gaze.comm_stream
        self.fusion = GradAllReducer(fus_fp.grad, whole(fus_fp), group, force, comm_stream=stream)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def on_fusion(self):
        self.fusion.on_segment("all")

    def on_gaze(self):
        if self.gaze is not None:
            self.gaze.on_segment("all")

    @property
    def on_eeg_segment(self):
        return self.eeg.on_segment if (self.eeg is not None and self.active) else None

    def finish(self):
        for r in (self.fusion, self.gaze, self.eeg):
            if r is not None:
                r.finish()

    def sync_flag(self, state_dev: torch.Tensor):
        """MAX over ranks of eg_step_state.found_inf (word 9), stream-ordered between eg_clip_coef and the AdamW kernels."""
        if self.active:
            dist.all_reduce(state_dev[9:10], op=dist.ReduceOp.MAX, group=self.group)

    def broadcast(self, *flats):
        for f in flats:
            if f is not None:
                broadcast_params(f, self.group)


def broadcast_params(flat: torch.Tensor, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=0, group=group)


def shard_indices(n: int, rank: int, world: int) -> range:
    """rank r takes samples r::world of each global batch (SURVEY §8e partitioning)."""
    return range(rank, n, world)
