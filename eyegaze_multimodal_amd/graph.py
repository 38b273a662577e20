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


class GraphedStep:
    def __init__(self, eng, opt, train: bool = True, reducer=None):
        self.eng, self.opt, self.train, self.reducer = eng, opt, train, reducer
        dev = eng.device
        self.x1 = torch.zeros(eng.B, eng.C, eng.T, device=dev)
        self.x2 = torch.zeros(eng.B, eng.C, eng.T, device=dev)
        self.labels = torch.zeros(eng.B, dtype=torch.int64, device=dev)
        self.one = torch.ones(1, device=dev)
        self.graphs: List[Tuple[str, torch.cuda.CUDAGraph]] = []
        self.captured = False

    def _capture(self):
        eng, opt = self.eng, self.opt
        side = torch.cuda.Stream(eng.device)
        side.wait_stream(torch.cuda.current_stream(eng.device))
        graphs = self.graphs
        with torch.cuda.stream(side):
            cur = {"g": torch.cuda.CUDAGraph(), "name": "forward"}
            cur["g"].capture_begin()

            def cut(next_name):
                cur["g"].capture_end()
                graphs.append((cur["name"], cur["g"]))
                if next_name is not None:
                    cur["g"], cur["name"] = torch.cuda.CUDAGraph(), next_name
                    cur["g"].capture_begin()

            eng.forward(self.x1, self.x2, self.labels, train=self.train)
            cut("bwd")
            order = []

            def seg(name):
                order.append(name)
                cur["name"] = name        # the graph just captured produced segment `name`
                cut("bwd")

            eng.backward(gloss=self.one, on_segment=seg)
            # the last cut() opened an empty capture for the optimiser: fill it
            cur["name"] = "optimizer"
            opt.step(eng)
            cut(None)
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
        for name, g in self.graphs:
            if name == "optimizer" and red is not None:
                red.finish()
            g.replay()
            if red is not None and name not in ("forward", "optimizer"):
                red.on_segment(name)
