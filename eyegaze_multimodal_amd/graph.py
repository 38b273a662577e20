"""hipGraph replay of one training step.  The step is a fixed sequence of ~300 short launches; issuing them from
Python costs more host time than the GPU needs to run them, so the sequence is captured once per (engine, mode)
and replayed.  Everything that changes between steps is read by the kernels from device memory: the inputs live
in static buffers and the dropout seed / learning rate / bias corrections in the device-resident eg_step_state.

The backward is captured as one graph PER SEGMENT (heads, cross, encoder.norm, layer L-1 .. 0, front-end) so that the
data-parallel all-reduce of a segment's gradient bucket can still be enqueued (eagerly, on the side stream) right
after that segment's graph is launched (ddp.GradAllReducer.on_segment)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _lib


class GraphedStep:
    def __init__(self, eng, opt, train: bool = True, reducer=None):
        self.eng, self.opt, self.train, self.reducer = eng, opt, train, reducer
        dev = eng.device
        self.x1 = torch.zeros(eng.B, eng.C, eng.T, device=dev)
        self.x2 = torch.zeros(eng.B, eng.C, eng.T, device=dev)
        self.labels = torch.zeros(eng.B, dtype=torch.int64, device=dev)
        self.one = torch.ones(1, device=dev)
        self.graphs: List[Tuple[List[str], torch.cuda.CUDAGraph]] = []  # (segments finished by this graph, graph)
        self.captured = False

    def _capture(self):
        eng, opt = self.eng, self.opt
        side = torch.cuda.Stream(eng.device)
        side.wait_stream(torch.cuda.current_stream(eng.device))
        graphs = self.graphs
        with torch.cuda.stream(side):
            cur = {"g": torch.cuda.CUDAGraph(), "names": [], "calls": _lib.CALLS}
            cur["g"].capture_begin()

            def cut(reopen=True):
                cur["g"].capture_end()
                graphs.append((cur["names"], cur["g"]))
                if reopen:
                    cur["g"], cur["names"], cur["calls"] = torch.cuda.CUDAGraph(), [], _lib.CALLS
                    cur["g"].capture_begin()

            eng.forward(self.x1, self.x2, self.labels, train=self.train)

            def seg(name):
                if _lib.CALLS == cur["calls"] and graphs:   # nothing launched since the last cut: same graph finished it
                    graphs[-1][0].append(name)
                    return
                cur["names"].append(name)
                cut()

            # Without a reducer the whole step is ONE graph: segment cuts only exist so that a bucket's all-reduce can be enqueued
            # between them.  (Round 2 always cut -- 12 graph launches per step, and the segment callback also switched the engine to
            # the two-piece weight-gradient launch of data-parallel runs: replay measured 4.22 ms against 3.77 ms eager.)
            eng.backward(gloss=self.one, gloss_ibs=(self.one if getattr(eng.cfg, "use_ibs", False) else None),
                         on_segment=(seg if self.reducer is not None else None))
            cur["names"].append("optimizer")
            opt.step(eng)
            cut(reopen=False)
        torch.cuda.current_stream(eng.device).wait_stream(side)
        self.captured = True

    def run(self, x1: torch.Tensor, x2: torch.Tensor, labels: torch.Tensor):
        """Caller has already published this step's scalars (opt.begin_step)."""
        self.x1.copy_(x1, non_blocking=True)
        self.x2.copy_(x2, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        if not self.captured:
            self._capture()
        red = self.reducer
        for names, g in self.graphs:
            if red is not None and "optimizer" in names:
                red.finish()
            g.replay()
            if red is not None:
                for name in names:
                    if name != "optimizer":
                        red.on_segment(name)
