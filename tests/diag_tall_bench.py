"""Diagnostic (manual, GPU box): the tall tile against the wide tile on the conv shapes; EYEGAZE_TALL_DBG bits switch parts of
the K loop off (1 no activation DMA refill, 2 a quarter of the MFMAs, 4 no weight refill) to find the limiting pipe."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import ctypes as C
import torch
from eyegaze_multimodal_amd import _lib as L
from eyegaze_multimodal_amd._lib import GemmDesc, call, ptr, rowmap
from tests.diag_ffn_bench import timed
from tests.test_gpu_ops import dev_state

for M, K in [(32768, 6400), (35840, 1792)]:
    A = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    W = (torch.randn(256, K, device="cuda") * 0.05).to(torch.bfloat16)
    Wf = torch.zeros_like(W).view(-1)
    call("eg_frag_order_rows", ptr(W), ptr(Wf), K, K, 1, 0)
    out = torch.zeros(M, 256, device="cuda", dtype=torch.bfloat16)
    st = dev_state()
    for tall in (0, 1):
        d = GemmDesc(); d.A, d.W, d.C, d.state = ptr(A), ptr(W), ptr(out), ptr(st)
        d.a, d.c = rowmap(K), rowmap(256); d.r = d.c; d.p = d.c
        d.M, d.N, d.K, d.ldw, d.dtype = M, 256, K, K, L.EG_BF16
        d.W_frag = ptr(Wf) if tall else None
        t = timed(lambda s=0: call("eg_gemm_nt", C.byref(d), s), reps=10)
        print(f"dbg={os.environ.get('EYEGAZE_TALL_DBG','0')} M={M} K={K} {'tall' if tall else 'wide'}: {t:7.1f} us  {2*M*256*K/t/1e6:7.1f} TFLOP/s", flush=True)
