"""Diagnostic (manual, GPU box): time eg_gemm_nt / eg_gemm_tn for the step's shapes."""
import sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from eyegaze_multimodal_amd import _lib as L
from eyegaze_multimodal_amd._lib import GemmDesc, GemmTNDesc, call, ptr, rowmap
dev = "cuda"

def bench_nt(M, N, K, residual=False, bias=True, act=0, reps=30, dtype=L.EG_BF16, row_tile=0, ln=0):
    td = torch.bfloat16 if dtype == L.EG_BF16 else torch.float32
    A = torch.randn(M, K, device=dev).to(td); W = torch.randn(N, K, device=dev).to(td); Cc = torch.zeros(M, N, device=dev, dtype=td)
    b = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev).to(td)
    d = GemmDesc(); d.A, d.W, d.C = ptr(A), ptr(W), ptr(Cc)
    d.bias = ptr(b) if bias else None; d.residual = ptr(R) if residual else None
    d.a, d.c = rowmap(K), rowmap(N); d.r = d.c; d.p = d.c
    d.M, d.N, d.K, d.ldw, d.act, d.dtype = M, N, K, K, act, dtype
    assert not row_tile, 'the row-complete tile was removed in round 3'
    if ln:
        gm = torch.ones(N, device=dev); Y = torch.zeros(M, N, device=dev, dtype=td); S = torch.zeros(M, 2, device=dev)
        raise SystemExit('the LayerNorm-epilogue tile was removed in round 3')
    for _ in range(5): call("eg_gemm_nt", C.byref(d), 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call("eg_gemm_nt", C.byref(d), 0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    es = 2 if dtype == L.EG_BF16 else 4
    byt = (M * K + M * N * (2 if residual else 1)) * es
    print(f"NT row={row_tile} ln={ln} M={M:6d} N={N:5d} K={K:5d} res={int(residual)} : {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s  {byt/us/1e6:6.2f} TB/s  blocks={((M+127)//128)*((N+127)//128)}")

SECTIONS = sys.argv[1:] or ["rounds", "rowtile", "fixed", "nscale"]

if "rounds" in SECTIONS:
    print("-- round quantisation: N=1024 K=256, 768 resident workgroups (256 CUs x 3) --")
    for mt in (48, 96, 97, 144, 192, 193, 260, 288):
        bench_nt(mt * 128, 1024, 256, act=1)
    print("-- same, N=256 --")
    for mt in (96, 192, 260, 384, 385, 520):
        bench_nt(mt * 128, 256, 256, residual=True)
if "rowtile" in SECTIONS:
    print("-- row-complete tile vs 128x128 tile, N=256 --")
    for K in (256, 768, 1024):
        bench_nt(33280, 256, K, residual=True)
        bench_nt(33280, 256, K, residual=True, row_tile=1)
        bench_nt(33280, 256, K, residual=True, ln=1)
if "fixed" in SECTIONS:
    print("-- fixed cost vs per-iteration cost (N=256, plain epilogue) --")
    for M in (128, 2048, 16640, 33280):
        for K in (128, 256, 512, 1024):
            bench_nt(M, 256, K, residual=False, bias=True)
if "nscale" in SECTIONS:
    print("-- N scaling at K=256 --")
    for N in (128, 256, 512, 768, 1024):
        bench_nt(33280, N, 256)
    bench_nt(32768, 256, 6400)
    bench_nt(8192, 8192, 8192)
