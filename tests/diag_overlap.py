"""Diagnostic: do latency-bound GEMM chains on two HIP streams overlap?"""
import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from eyegaze_multimodal_amd import _lib as L
from eyegaze_multimodal_amd._lib import GemmDesc, GemmTNDesc, call, ptr, rowmap
dev = "cuda"
M, N, K = 33280, 256, 256
td = torch.bfloat16
A = torch.randn(M, K, device=dev).to(td); W = torch.randn(N, K, device=dev).to(td); Cc = torch.zeros(M, N, device=dev, dtype=td)
d = GemmDesc(); d.A, d.W, d.C = ptr(A), ptr(W), ptr(Cc); d.a, d.c = rowmap(K), rowmap(N); d.r = d.c; d.p = d.c
d.M, d.N, d.K, d.ldw, d.act, d.dtype = M, N, K, K, 0, L.EG_BF16
dY = torch.randn(M, 1024, device=dev).to(td); X = torch.randn(M, 256, device=dev).to(td)
part = torch.zeros(64 * 1024 * 256, device=dev)
t = GemmTNDesc(); t.dY, t.X, t.partial = ptr(dY), ptr(X), ptr(part); t.y, t.x = rowmap(1024), rowmap(256)
t.M, t.N, t.K, t.splits, t.dtype = M, 1024, 256, 48, L.EG_BF16
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run(nA, nB, reps=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_event(e0); sB.wait_event(e0)
    for _ in range(reps):
        for _ in range(nA): call("eg_gemm_nt", C.byref(d), sA.cuda_stream)
        for _ in range(nB): call("eg_gemm_tn", C.byref(t), sB.cuda_stream)
    torch.cuda.current_stream().wait_stream(sA); torch.cuda.current_stream().wait_stream(sB)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
run(4, 4)
a = run(8, 0); b = run(0, 8); ab = run(8, 8)
print(f"NT x8 alone {a:.1f} us, TN x8 alone {b:.1f} us, both streams {ab:.1f} us (sum {a+b:.1f})")
