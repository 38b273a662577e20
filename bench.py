#!/usr/bin/env python3
"""bench.py — train samples/s of the dual-stream window classifier step on N MI355X of one node.

A step = forward (train mode, dropout on) + backward + gradient all-reduce (N > 1) + clip_grad_norm_(1.0) + AdamW
over one batch of synthetic [B=256, C=8, T=1024] window pairs per GPU, inputs resident in HBM.
Workload = BASELINE.json configs[1] (two 1-D-conv streams, concat fusion, bf16, batch 256) unless --workload says
otherwise.  Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (ctor overrides, description)
    "cfg2": (dict(use_spectrogram=False, use_ibs=False, use_cross_attention=False),
             "BASELINE configs[1]: two 1-D-conv streams + 6-layer Siamese encoder, concat fusion (no cross-attention)"),
    "cfg3": (dict(use_spectrogram=False, use_ibs=False, use_cross_attention=True),
             "BASELINE configs[2]/[3]: + bidirectional cross-stream attention fusion"),
}
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def cpu_baseline(kw, C, T, seconds_budget=25.0):
    """The CPU oracle (oracle/dual_eeg_oracle.py, kind 'port') timed on this host: train-mode forward + backward +
    clip + AdamW on a bounded sample (B=32) of the same synthetic workload."""
    from oracle import dual_eeg_oracle as O
    from eyegaze_multimodal_amd.data import randn_windows
    cores = min(16, len(os.sched_getaffinity(0)))  # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    cfg = O.ModelCfg(in_channels=C, max_len=T // 4, **kw)
    sd = O.synthetic_state_dict(cfg, seed=1)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    Bc = 32
    x1, x2, labels = randn_windows(Bc, C, T, seed=1234, num_classes=cfg.num_classes)
    state = {}

    def step(i):
        for p in params.values():
            p.grad = None
        out = O.forward(x1, x2, params, cfg, labels, train=True)
        out["loss_ce"].backward()
        with torch.no_grad():
            O.clip_and_adamw({k: p.data for k, p in params.items()}, {k: p.grad for k, p in params.items()}, state, step=i + 1)
    step(0)
    t0 = time.perf_counter()
    n = 0
    while n < 2 or (time.perf_counter() - t0 < seconds_budget / 2 and n < 8):
        step(n + 1)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(Bc / dt, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fwd+bwd+clip+AdamW, train mode, B={Bc} windows of the same synthetic workload, {n} timed steps after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=256, help="windows pairs per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the HIP path has no CPU fallback"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from eyegaze_multimodal_amd import DualEEGTransformer, HipAdamW
    from eyegaze_multimodal_amd.data import randn_windows
    from eyegaze_multimodal_amd.ddp import GradAllReducer, broadcast_params, bucket_ranges

    C, T, B = 8, 1024, args.batch
    kw, desc = WORKLOADS[args.workload]
    kw = dict(kw, num_classes=3)
    torch.manual_seed(42)
    model = DualEEGTransformer(in_channels=C, max_len=T // 4, compute_dtype=args.dtype, **kw).to(dev)
    model.train()
    eng = model.engine(B, T, dev)
    fp = model._flat
    broadcast_params(fp.flat)
    opt = HipAdamW(model, lr=1e-4, weight_decay=0.01)
    x1, x2, labels = randn_windows(B, C, T, seed=1234 + rank, num_classes=3, device=dev)
    ranges = bucket_ranges(fp.names, fp.offsets, fp.total, model.cfg.num_layers, model.cfg.use_cross_attention)
    reducer = GradAllReducer(fp.grad, ranges) if world > 1 else None
    one = torch.ones(1, device=dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    from eyegaze_multimodal_amd.graph import GraphedStep
    # N > 1: eager launches (same speed: the step is GPU-bound) keep the RCCL buckets out of graph capture
    graphed = None if (args.eager or world > 1) else GraphedStep(eng, opt, train=True, reducer=reducer)

    def step(i, probe=None, eager=False):
        opt.begin_step(eng, seed=1000 + i, grad_scale=(reducer.grad_scale if reducer else 1.0))
        if graphed is not None and not eager:
            graphed.run(x1, x2, labels)
            return
        eng.probes = {"conv1_fwd": probe} if probe else {}
        eng.forward(x1, x2, labels, train=True)
        eng.backward(gloss=one, on_segment=(reducer.on_segment if reducer else None))
        if reducer:
            reducer.finish()
        opt.step(eng)

    step(0, eager=True)  # first step eagerly: lazy workspace allocation and one-time kernel attributes
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    probe_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, probe=evs[i])  # HIP events on the launch stream; read after the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if graphed is not None:
        # HIP events cannot be recorded inside a replayed graph: time the probed launch live right after the timed
        # region (same buffers, same stream), eager, events on the launch stream
        torch.cuda.synchronize()
        for a, b in evs[:8]:
            step(args.warmup + args.steps, probe=(a, b), eager=True)
        torch.cuda.synchronize()
        evs = evs[:8]
    probe_ms = [a.elapsed_time(b) for a, b in evs]
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    loss = float(eng.a["loss"])
    assert loss == loss, "loss is NaN"

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        # dominant-kernel roofline: conv-1 as an MFMA GEMM, M = 2B*T2 rows, N = d, K = 25*d  
        M, N, K = eng.NB * eng.T2, model.cfg.d_model, eng.k * model.cfg.d_model
        flops = 2.0 * M * N * K
        kms = sum(probe_ms) / len(probe_ms)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        achieved = flops / (kms * 1e-3) / 1e12
        out = {
            "metric": "train samples/sec (gaze+EEG windows)", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "batch_per_gpu": B, "global_batch": world * B, "C": C, "T": T,
                       "seq_len": eng.S, "d_model": model.cfg.d_model, "layers": model.cfg.num_layers,
                       "step": "fwd(train,dropout)+bwd+allreduce+clip+AdamW", "parallelism": f"dp{world}",
                       "final_loss": round(loss, 5)},
            "roofline": {"kernel": "gemm_nt_kernel<bf16> (conv-1 forward: strided Conv1d as MFMA GEMM)" if args.dtype == "bf16"
                         else "gemm_nt_kernel<f32>", "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                         "launch_ms": round(kms, 4), "algorithmic_flops_per_launch": flops},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(kw, C, T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
