"""Diagnostic (manual, GPU box): what plain streaming kernels (torch element-wise ops, graph-replayed) take for the HBM traffic
of the encoder's products at the benchmark size -- the floor the GEMM launches are held against in DESIGN.md."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

M = 33280


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        g.capture_begin()
        for _ in range(reps):
            fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


bf = torch.bfloat16
x = torch.randn(M, 256, device="cuda").to(bf)
r = torch.randn(M, 256, device="cuda").to(bf)
y = torch.empty(M, 256, device="cuda", dtype=bf)
h = torch.randn(M, 1024, device="cuda").to(bf)
q = torch.empty(M, 768, device="cuda", dtype=bf)
h2 = torch.empty(M, 1024, device="cuda", dtype=bf)
cases = [
    ("out-proj traffic: read 17 + 17 MB, write 17 MB (y = x + r)", lambda: torch.add(x, r, out=y), 51.1),
    ("q|k|v traffic: read 17 MB, write 51 MB (q = cat(x, x, x))", lambda: torch.cat([x, x, x], dim=1, out=q), 68.2),
    ("FFN-1 traffic: read 17 MB, write 68 MB (h2 = cat(x x 4))", lambda: torch.cat([x, x, x, x], dim=1, out=h2), 85.2),
    ("FFN-2 traffic: read 68 + 17 MB, write 17 MB (y = h[:, :256] + ... 4 slices + r)",
     lambda: torch.add(torch.add(torch.add(h[:, :256], h[:, 256:512]), torch.add(h[:, 512:768], h[:, 768:])), r, out=y), None),
    ("copy 68 MB -> 68 MB", lambda: h2.copy_(h), 136.3),
    ("LayerNorm-bwd traffic: read 17 + 17, write 17 + 17 MB", lambda: (torch.add(x, r, out=y), torch.sub(x, r, out=r)), None),
]
for name, fn, mb in cases:
    t = timed(fn)
    extra = f"  {mb / t:5.2f} TB/s" if mb else ""
    print(f"{name}: {t:6.1f} us{extra}", flush=True)
