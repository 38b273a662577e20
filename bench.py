#!/usr/bin/env python3
"""bench.py — train samples/s of the dual-stream window classifier step on N MI355X of one node.

A step = forward (train mode, dropout on) + backward + gradient all-reduce (N > 1) + clip_grad_norm_(1.0) + AdamW
over one batch of synthetic [B=256, C=8, T=1024] window pairs per GPU, inputs resident in HBM.
Workload = BASELINE.json configs[2]/[3] (two 1-D-conv streams + bidirectional cross-stream attention fusion, bf16, batch 256
per GPU: the configuration the 1/2/4/8-GPU metric is quoted on) unless --workload says otherwise.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.

Ranks: under torchrun (WORLD_SIZE set) this process IS one rank.  Started plainly with --gpus N > 1 it becomes a launcher:
it spawns N fresh rank processes (before touching the GPU itself -- a process that has initialised HIP is never re-executed),
relays rank 0's JSON line and exits with the worst child status.
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
    "cfg5": (dict(use_spectrogram=True, use_ibs=False, use_cross_attention=True),
             "BASELINE configs[4]: + STFT image -> 2-D CNN tokens per channel (third modality), cross-attention fusion"),
    "a5": (dict(use_spectrogram=True, use_ibs=True, use_robust_ibs=True, use_cross_attention=True),
           "reference default flags (A5): spectrogram tokens + 42 inter-stream synchrony tokens + cross-attention, loss_ce + loss_ibs_cls"),
    "mm5": (dict(use_spectrogram=True, use_ibs=False, use_cross_attention=True),
            "BASELINE configs[4]: three modalities -- gaze image (in-tree 2-D CNN branch) + EEG windows with STFT->2-D-CNN tokens, "
            "cross-attention fusion -- joined by FuzzyGatingFusion on the logits; the whole multimodal step "
            "(train_multimodal_fuzzy_fusion.py:395-543: fused + auxiliary CE + temperature regulariser, one global clip, "
            "per-group AdamW, warm-up + cosine per step, dynamic loss scaling in fp16)"),
    "a5c32": (dict(use_spectrogram=True, use_ibs=True, use_robust_ibs=True, use_cross_attention=True, in_channels=32),
              "the reference's default yaml (4_Experiments/configs/dual_eeg_transformer.yaml:38-53): A5 flags at in_channels = 32 "
              "(S = 139, 1024-wide synchrony rows)"),
}
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0     # HBM3E, MI355X_MICROARCH.md (about 6.3 TB/s is reachable by a streaming kernel)


def plan_ranks(gpus: int, env: dict, share_gpu: bool = False, port: int = 0):
    """Environment of every child rank the launcher starts (one process per GPU, rendezvous on 127.0.0.1).
    Returns [] when this process must run as a rank itself (N == 1, or a launcher such as torchrun already set WORLD_SIZE)."""
    if gpus <= 1 or "WORLD_SIZE" in env:
        return []
    if not port:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    out = []
    for r in range(gpus):
        e = dict(env)
        e.update(WORLD_SIZE=str(gpus), RANK=str(r), LOCAL_RANK=str(0 if share_gpu else r), LOCAL_WORLD_SIZE=str(gpus),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out.append(e)
    return out


def launch_ranks(plans, argv=None, timeout_s: float = None, poll_s: float = 0.2) -> int:
    """Starts one fresh interpreter per rank with this script's own arguments; rank 0's stdout is relayed verbatim.
    The children are POLLED: the first rank that exits non-zero (import error, GPU fault, failed rendezvous) ends the run -- its
    siblings, which would otherwise sit in init_process_group / a collective until the collective timeout, are terminated (they
    are this launcher's own fresh children, addressed by PID) and that status is returned.  A launcher-level timeout
    (EYEGAZE_LAUNCH_TIMEOUT seconds, default 1500) bounds the whole run the same way."""
    import subprocess
    argv = sys.argv[1:] if argv is None else argv
    if timeout_s is None:
        timeout_s = float(os.environ.get("EYEGAZE_LAUNCH_TIMEOUT", "1500"))
    procs = [subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=e,
                              stdout=(None if i == 0 else subprocess.DEVNULL)) for i, e in enumerate(plans)]
    t0, rc, live = time.monotonic(), 0, list(procs)
    while live and rc == 0:
        for p_ in list(live):
            st = p_.poll()
            if st is not None:
                live.remove(p_)
                if st != 0:
                    rc = abs(st) or 1
        if live and rc == 0:
            if time.monotonic() - t0 > timeout_s:
                rc = 124
                print(f"bench.py launcher: ranks still running after {timeout_s:.0f} s, terminating them", file=sys.stderr)
                break
            time.sleep(poll_s)
    for p_ in live:                              # a failed or timed-out run: end what is left, by PID
        p_.terminate()
    for p_ in live:
        try:
            p_.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p_.kill()
            p_.wait()
    return rc


def granted_cores():
    """CPU cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256 cores in the
    affinity mask but grants a 16-CPU quota per GPU: 256 threads on a 16-CPU quota run the oracle several times SLOWER than 16)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    return min(aff, quota) if quota else aff, {"affinity": aff, "cgroup_quota": quota, "cpu_count": os.cpu_count()}


def cpu_baseline(kw, C, T, Bc=256, seconds_budget=30.0):
    """The CPU oracle (oracle/dual_eeg_oracle.py, kind 'port') timed on this host: train-mode forward + backward +
    clip + AdamW on the same synthetic workload at the same batch (B = 256, fp32, all granted cores,
    1 warm-up step, then up to 3 timed steps inside a bounded time budget)."""
    from oracle import dual_eeg_oracle as O
    from eyegaze_multimodal_amd.data import randn_windows
    cores, avail = granted_cores()                  # every core this process is granted (all cores, count stated)
    torch.set_num_threads(cores)
    cfg = O.ModelCfg(in_channels=C, max_len=T // 4, **kw)
    sd = O.synthetic_state_dict(cfg, seed=1)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x1, x2, labels = randn_windows(Bc, C, T, seed=1234, num_classes=cfg.num_classes)
    state = {}

    def step(i):
        for p in params.values():
            p.grad = None
        out = O.forward(x1, x2, params, cfg, labels, train=True)
        (out["loss_ce"] + (out["loss_ibs_cls"] if "loss_ibs_cls" in out else 0.0)).backward()   # lambda_ibs_cls = 1 (yaml :98)
        with torch.no_grad():
            O.clip_and_adamw({k: p.data for k, p in params.items()}, {k: p.grad for k, p in params.items()}, state, step=i + 1)
    tw = time.perf_counter()
    step(0)
    tw = time.perf_counter() - tw
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (n < 3 and (time.perf_counter() - t0) + tw * 1.1 < seconds_budget):
        step(n + 1)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(Bc / dt, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "cores_available": avail,
            "kind": "port",
            "sample": f"oracle fwd+bwd+clip+AdamW, train mode, B={Bc} windows of the same synthetic workload, {n} timed steps after 1 warm-up"}


ROUTES = {0: "gemm_nt_kernel", 1: "gemm_nt_wide_kernel", 2: "rs_gemm_kernel", 3: "gemm_nt_row_kernel", 4: "ffn_chain_kernel",
          8: "attn_block_fwd_kernel"}


def roofline_from_probes(probes, nsteps_probed, dtype):
    """Groups the HIP-event-bracketed eg_gemm_nt / eg_ffn_chain launches by the kernel that served them; returns the roofline of
    the group with the most GPU time and the per-kernel list."""
    peak = PEAK_F32_TFLOPS if dtype == "f32" else PEAK_BF16_TFLOPS
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    groups = {}
    for a_, b_, f_, nb_, shape_, route_ in probes:
        g_ = groups.setdefault(route_, {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        g_["n"] += 1
        g_["ms"] += a_.elapsed_time(b_)
        g_["flops"] += f_
        g_["bytes"] += nb_
    by_kernel = [{"kernel": f"{ROUTES.get(r_, r_)}<{dtype}>", "launches_per_step": g_["n"] // nsteps_probed,
                  "ms_per_step": round(g_["ms"] / nsteps_probed, 4), "launch_ms": round(g_["ms"] / g_["n"], 5),
                  "GB/s": round(g_["bytes"] / g_["ms"] / 1e6, 1), "TFLOP/s": round(g_["flops"] / g_["ms"] / 1e9, 1)}
                 for r_, g_ in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])]
    r_, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    # the dominant kernel's launches split by the roofline that binds each (its products straddle the ridge: the encoder's
    # K <= 1024 products sit below it, the strided convolutions above): the aggregate fraction mixes the two
    split = {"hbm": {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}, "mfma": {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}}
    for a_, b_, f_, nb_, shape_, route_ in probes:
        if route_ == r_:
            c_ = split["hbm" if f_ / max(nb_, 1.0) < ridge else "mfma"]
            c_["n"] += 1
            c_["ms"] += a_.elapsed_time(b_)
            c_["flops"] += f_
            c_["bytes"] += nb_
    by_bound = [{"bound": k_, "launches_per_step": c_["n"] // nsteps_probed, "ms_per_step": round(c_["ms"] / nsteps_probed, 4),
                 "GB/s": round(c_["bytes"] / c_["ms"] / 1e6, 1), "TFLOP/s": round(c_["flops"] / c_["ms"] / 1e9, 1),
                 "frac": round((c_["bytes"] / c_["ms"] / 1e6 / PEAK_HBM_GBS) if k_ == "hbm" else (c_["flops"] / c_["ms"] / 1e9 / peak), 4)}
                for k_, c_ in split.items() if c_["n"]]
    kms, flops, nbytes = dom["ms"] / dom["n"], dom["flops"] / dom["n"], dom["bytes"] / dom["n"]
    tflops, gbs = flops / (kms * 1e-3) / 1e12, nbytes / (kms * 1e-3) / 1e9
    hbm_bound = (flops / max(nbytes, 1.0)) < ridge
    return {"kernel": f"{ROUTES.get(r_, r_)}<{dtype}> ({dom['n'] // nsteps_probed} launches per step)",
            "bound": "hbm" if hbm_bound else "mfma", "achieved": round(gbs if hbm_bound else tflops, 2),
            "peak": PEAK_HBM_GBS if hbm_bound else peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": round((gbs / PEAK_HBM_GBS) if hbm_bound else (tflops / peak), 4), "traffic": None,
            "launch_ms": round(kms, 5), "algorithmic_bytes_per_launch": round(nbytes),
            "algorithmic_flops_per_launch": round(flops), "flop_per_byte": round(flops / max(nbytes, 1.0), 1),
            "ridge_flop_per_byte": round(ridge, 1),
            "mfma": {"achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4)},
            "hbm": {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)},
            "dominant_by_bound": by_bound, "eg_gemm_nt_by_kernel": by_kernel}


def attach_traffic(roof, workload, dtype, B):
    """`roofline.traffic` = HBM bytes per launch of the dominant kernel from the PMC counters.  FETCH_SIZE and WRITE_SIZE cannot share
    a rocprofv3 pass (and never ride along with a timed run), so the value is NOT measured by this run: it is read from the newest
    committed collection of this command (profiles/collect_r03.sh: separate --pmc passes, KiB -> B, FETCH x2 per the gfx950
    correction), and `traffic_source` says which file, collected when, for which kernel."""
    import datetime
    if B != 256:
        return
    for rnd in ("r03", "r02"):
        f = REPO / "profiles" / f"{rnd}_pmc_{workload}_{dtype}.json"
        if not f.exists():
            continue
        recs = json.loads(f.read_text())
        for rec in (recs if isinstance(recs, list) else [recs]):
            if roof["kernel"].startswith(rec.get("kernel", "?")):
                roof["traffic"] = rec.get("hbm_bytes_per_launch")
                roof["traffic_source"] = {"file": f"profiles/{f.name}", "collected": rec.get("collected") or
                                          datetime.date.fromtimestamp(f.stat().st_mtime).isoformat(), "kernel": rec.get("kernel"),
                                          "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command "
                                                 "(not measured by this run)"}
                return
    roof["traffic_source"] = None


def f32_leg(ctor, B, C, T, dev, steps=5, warmup=2):
    """The reference's own precision (train_art.py:170-222 runs fp32) beside the headline dtype: a few timed steps of the same
    step in compute_dtype='f32' (exact-fp32 MFMA path), so the driver's record carries it."""
    from eyegaze_multimodal_amd import HipAdamW
    from eyegaze_multimodal_amd.data import randn_windows
    torch.manual_seed(42)
    model = ctor("f32").to(dev)
    model.train()
    eng = model.engine(B, T, dev)
    opt = HipAdamW(model, lr=1e-4, weight_decay=0.01)
    x1, x2, labels = randn_windows(B, C, T, seed=1234, num_classes=3, device=dev)
    one = torch.ones(1, device=dev)

    def step(i):
        opt.begin_step(eng, seed=2000 + i)
        eng.forward(x1, x2, labels, train=True)
        eng.backward(gloss=one, gloss_ibs=(one if model.cfg.use_ibs else None))
        opt.step(eng)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss = float(eng.a["loss"])
    assert loss == loss
    del eng, opt, model
    torch.cuda.empty_cache()
    return {"dtype": "f32", "value": round(B / dt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt, 4), "steps": steps,
            "warmup": warmup, "note": "same step at the reference's precision (exact-fp32 MFMA path), secondary to the headline dtype"}


def run_mm5(args, dev, world=1, rank=0, use_dist=False):
    """BASELINE configs[4]: the multimodal logit-fusion step of train_multimodal_fuzzy_fusion.py at B = 256 per GPU; with more
    than one rank the three gradient sets are exchanged by ddp.MultimodalReducers (weak scaling: every rank its own batch)."""
    import gc
    from eyegaze_multimodal_amd.train_multimodal_fuzzy_fusion import build_from_config, synth_multimodal
    B, C, T, F_, W_ = args.batch, 8, 1024, 64, 16
    config = {"data": {"window_size": T, "num_classes": 3},
              "eeg_encoder": {"in_channels": C, "use_spectrogram": True, "use_ibs": False, "use_cross_attention": True},
              "gaze_encoder": {"d_model": 256}, "fusion": {"mode": "full"},
              "training": {"encoder_learning_rate": 1e-4, "fusion_learning_rate": 1e-3, "weight_decay": 0.01, "max_grad_norm": 1.0,
                           "epochs": 10, "steps_per_epoch": 1000, "warmup_epochs": 1, "fp16": args.dtype == "fp16",
                           "compute_dtype": args.dtype}}
    tr = build_from_config(config, dev)
    tr.force_dist = bool(args.force_dist)
    img1, img2, x1, x2, y = (t_.to(dev) for t_ in synth_multimodal(B, C, T, F_, W_, 3, seed=1234 + rank))
    probes = []
    tr.train_step(img1, img2, x1, x2, y)                      # lazy workspaces, one-time kernel attributes
    engines = tr._engines(B, T, F_, W_)
    for e in engines:
        e.probe_all = probes
    tr.train_step(img1, img2, x1, x2, y)
    launches_per_step = len(probes)
    probes.clear()
    n_probe_steps = min(4, max(1, args.steps // 20), args.steps)
    stream0 = torch.cuda.current_stream(dev)
    pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_probe_steps * launches_per_step)]
    for a_, b_ in pool:
        a_.record(stream0)
        b_.record(stream0)
    for e in engines:
        e.probe_all, e.probe_pool = None, pool
    torch.cuda.synchronize()
    gc.collect()    # BEFORE the warm-up: a collection between warm-up and timed region leaves the GPU idle for tens of ms
    gc.disable()
    for _ in range(args.warmup):
        tr.train_step(img1, img2, x1, x2, y)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        for e in engines:
            e.probe_all = probes if i >= args.steps - n_probe_steps else None
        out = tr.train_step(img1, img2, x1, x2, y)
    for e in engines:
        e.probe_all = None
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    loss = float(out["loss"])
    assert loss == loss, "loss is NaN"
    eeg = engines[0]
    if rank == 0:
        res = {"metric": "train samples/sec (gaze+EEG windows)", "value": round(world * B * args.steps / elapsed, 2), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"mm5: {WORKLOADS['mm5'][1]}", "batch_per_gpu": B, "global_batch": world * B, "C": C, "T": T,
                          "image": [F_, W_], "seq_len": eeg.S, "d_model": 256, "layers": 6,
                          "step": "fwd(image CNN + EEG, train, dropout) + fuzzy fusion + losses + bwd"
                                  + (" + 3-set gradient all-reduce" if use_dist else "") + " + global clip + per-group AdamW"
                                  + (" + loss scaling" if args.dtype == "fp16" else ""),
                          "parallelism": f"dp{world}", "final_loss": round(loss, 5)},
               "roofline": roofline_from_probes(probes, n_probe_steps, args.dtype)}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline_mm5(tr, C, T, F_, W_, Bc=args.cpu_batch)
        emit(res)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline_mm5(tr, C, T, F_, W_, Bc=256, seconds_budget=30.0):
    """The CPU restatement of the multimodal step (oracle/multimodal_oracle.py::Stepper, kind 'port') on this host."""
    import copy
    from oracle import dual_eeg_oracle as O
    from oracle.multimodal_oracle import Stepper
    from eyegaze_multimodal_amd.train_multimodal_fuzzy_fusion import synth_multimodal
    cores, avail = granted_cores()
    torch.set_num_threads(cores)
    m = tr.model
    gaze = copy.deepcopy(m.gaze_encoder).cpu().float()
    eeg_sd = {k: v.detach().cpu().float().clone() if v.dtype.is_floating_point else v.detach().cpu().clone()
              for k, v in m.eeg_encoder.state_dict().items()}
    fus_sd = {k: v.detach().cpu().clone() for k, v in m.fusion.state_dict().items()}
    cfg = O.ModelCfg(in_channels=C, num_classes=3, max_len=T // 4, use_spectrogram=True, use_ibs=False, use_cross_attention=True)
    st = Stepper(gaze, cfg, eeg_sd, fus_sd, "full", tr.encoder_lr, tr.fusion_lr, tr.wd, tr.max_norm, tr.lams, tr.treg,
                 tr.warmup_steps, tr.total_steps)
    batch = synth_multimodal(Bc, C, T, F_, W_, 3, seed=1234)
    tw = time.perf_counter()
    st.step(*batch)
    tw = time.perf_counter() - tw
    t0, n = time.perf_counter(), 0
    while n < 1 or (n < 3 and (time.perf_counter() - t0) + tw * 1.1 < seconds_budget):
        st.step(*batch)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(Bc / dt, 3), "unit": "samples/s", "cores": cores, "cores_available": avail, "kind": "port",
            "sample": f"multimodal oracle step (image CNN + EEG oracle + fuzzy fusion + clip + AdamW), eval-mode dropout, B={Bc}, "
                      f"{n} timed steps after 1 warm-up"}


_REAL_STDOUT = None


def protect_stdout():
    """The contract is ONE JSON line on stdout, but libraries write to file descriptor 1 on their own (RCCL prints a version
    banner at communicator creation, gloo a connection note).  Everything this process and its libraries print goes to stderr
    from here on; `emit` writes the result line to the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        sys.stdout = os.fdopen(os.dup(2), "w")


def emit(obj):
    line = json.dumps(obj)
    out = _REAL_STDOUT or sys.stdout
    out.write(line + "\n")
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=256, help="windows pairs per GPU")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "f32"],
                    help="compute dtype (default bf16; fp16 for --workload mm5, as BASELINE configs[4] states)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay captured hipGraphs instead of eager launches")
    ap.add_argument("--eager", action="store_true", help="(default) accepted for compatibility")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0")
    ap.add_argument("--probe-dump", default=None, help="write the per-launch gemm_nt timings (shape, us, TFLOP/s, GB/s) to this file")
    ap.add_argument("--h2d", default="none", choices=["none", "sync", "overlap"],
                    help="PCIe-inclusive variant (never the headline value): copy the batch from pinned host memory every step, "
                         "on the compute stream (sync) or double-buffered on a copy stream (overlap)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: run the RCCL path with a single rank")
    ap.add_argument("--cpu-batch", type=int, default=256, help="batch of the CPU baseline leg (SURVEY 8d: 256)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the secondary f32 measurement (the reference's precision)")
    ap.add_argument("--plan-only", action="store_true",
                    help="rehearsal without a GPU: form the process group, report world/rank and exit (CPU test of the launcher)")
    args = ap.parse_args()
    if args.dtype is None:
        args.dtype = "fp16" if args.workload == "mm5" else "bf16"

    plans = plan_ranks(args.gpus, os.environ, share_gpu=args.share_gpu)
    if plans:                                  # launcher: nothing below runs in this process, the GPU stays untouched
        sys.exit(launch_ranks(plans))
    protect_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    if world != max(args.gpus, 1) and "WORLD_SIZE" in os.environ and args.gpus > 1:
        raise SystemExit(f"--gpus {args.gpus} disagrees with WORLD_SIZE={world}")
    use_dist = world > 1 or args.force_dist
    if args.plan_only:
        if use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
            tot = torch.ones(1)
            dist.all_reduce(tot)
            assert int(tot) == world
            dist.barrier()
        if rank == 0:
            emit({"plan_only": True, "n_gpus": world, "workload": args.workload, "local_rank": local_rank})
        if use_dist:
            dist.destroy_process_group()
        return
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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

    if args.workload == "mm5":
        return run_mm5(args, dev, world, rank, use_dist)
    T, B = 1024, args.batch
    kw, desc = WORKLOADS[args.workload]
    kw = dict(kw, num_classes=3)
    C = kw.pop("in_channels", 8)
    torch.manual_seed(42)
    ctor = lambda dt_: DualEEGTransformer(in_channels=C, max_len=T // 4, compute_dtype=dt_, **kw)
    model = ctor(args.dtype).to(dev)
    model.train()
    eng = model.engine(B, T, dev)
    fp = model._flat
    broadcast_params(fp.flat)
    opt = HipAdamW(model, lr=1e-4, weight_decay=0.01)
    x1, x2, labels = randn_windows(B, C, T, seed=1234 + rank, num_classes=3, device=dev)
    ranges = bucket_ranges(fp.names, fp.offsets, fp.total, model.cfg.num_layers, model.cfg.use_cross_attention)
    from eyegaze_multimodal_amd.ddp import coalesce_groups
    reducer = (GradAllReducer(fp.grad, ranges, force=args.force_dist, groups=coalesce_groups(ranges, model.cfg.num_layers))
               if use_dist else None)
    one = torch.ones(1, device=dev)

    from eyegaze_multimodal_amd.graph import GraphedStep
    # default: every kernel is launched eagerly so that HIP events can bracket the dominant kernel's launches INSIDE the
    # timed region (events cannot be recorded inside a replayed graph).  The step is GPU-bound, so eager == graph speed;
    # --graph replays captured hipGraphs instead (then the probe runs right after the timed region).
    graphed = GraphedStep(eng, opt, train=True, reducer=reducer) if (args.graph and world == 1) else None

    host = None
    if args.h2d != "none":
        host = [t_.cpu().pin_memory() for t_ in (x1, x2, labels)]
        dbuf = [[torch.empty_like(x1), torch.empty_like(x2), torch.empty_like(labels)] for _ in range(2)]
        copy_stream = torch.cuda.Stream(dev)
        copied = [torch.cuda.Event() for _ in range(2)]
        used = [torch.cuda.Event() for _ in range(2)]

        def stage(slot):            # host -> device of one batch on the copy stream
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(used[slot])
                for d_, h_ in zip(dbuf[slot], host):
                    d_.copy_(h_, non_blocking=True)
                copied[slot].record(copy_stream)
        for ev in used:
            ev.record(torch.cuda.current_stream(dev))
        if args.h2d == "overlap":
            stage(0)

    def step(i, probe=False, eager=False):
        nonlocal x1, x2, labels
        if host is not None:
            slot = i & 1
            if args.h2d == "sync":
                for d_, h_ in zip(dbuf[slot], host):
                    d_.copy_(h_, non_blocking=True)
            else:
                torch.cuda.current_stream(dev).wait_event(copied[slot])
                stage(slot ^ 1)     # the next batch travels while this one computes
            x1, x2, labels = dbuf[slot]
        _step(i, probe, eager)
        if host is not None:
            used[i & 1].record(torch.cuda.current_stream(dev))

    def _step(i, probe=False, eager=False):
        opt.begin_step(eng, seed=1000 + i, grad_scale=(reducer.grad_scale if reducer else 1.0))
        if graphed is not None and not eager:
            graphed.run(x1, x2, labels)
            return
        eng.probe_all = probes if probe else None
        eng.forward(x1, x2, labels, train=True)
        eng.backward(gloss=one, gloss_ibs=(one if model.cfg.use_ibs else None),
                     on_segment=(reducer.on_segment if reducer else None))
        eng.probe_all = None
        if reducer:
            reducer.finish()
        opt.step(eng)

    probes = []
    step(0, probe=True, eager=True)  # first step eagerly: lazy workspace allocation, one-time kernel attributes, launch count
    launches_per_step = len(probes)
    probes.clear()
    # the timed region's probe events are created (and recorded once, which is what materialises a hipEvent) HERE, outside it:
    # inside, a probed launch costs two hipEventRecord calls and the host stays ahead of the GPU
    # a probed step costs the host ~0.6 ms of hipEventRecord calls (the GPU idles meanwhile), so few steps are probed: 4 of 100, 1 of 20
    n_probe_steps = 0 if graphed is not None else min(4, max(1, args.steps // 20), args.steps)
    stream0 = torch.cuda.current_stream(dev)
    pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            for _ in range(n_probe_steps * launches_per_step)]
    for a_, b_ in pool:
        a_.record(stream0)
        b_.record(stream0)
    eng.probe_pool = pool
    torch.cuda.synchronize()
    # no cyclic-GC pause inside the timed region: a step allocates a few hundred short-lived ctypes descriptors and event
    # tuples, and a generation-2 collection over the process' tensors costs tens of milliseconds -- invisible in 100 steps,
    # a doubling of ms/step in a 20-step run.  Collected BEFORE the warm-up steps, so that the GPU does not sit idle between
    # the warm-up and the timed region.
    import gc
    gc.collect()
    gc.disable()
    for i in range(args.warmup):
        step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # the probed steps are the LAST of the region: the host is then several steps ahead of the GPU, so the ~0.6 ms of
        # hipEventRecord calls of a probed step hide behind queued work (as the first step after the synchronize they idled the GPU)
        step(args.warmup + i, probe=(i >= args.steps - n_probe_steps))  # events on the launch stream, read after the region
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if graphed is not None:
        for i in range(4):
            step(args.warmup + args.steps + i, probe=True, eager=True)
        torch.cuda.synchronize()
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    loss = float(eng.a["loss"])
    assert loss == loss, "loss is NaN"

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        # eg_gemm_nt (every forward product -- strided convs, q/k/v/out, heads -- and every backward-data product) routes a launch
        # to one of its kernels (eg_gemm_nt_route); eg_ffn_chain is the feed-forward pair.  HIP events bracket EVERY such launch
        # of the timed region's last steps on the launch stream; launches are grouped by the kernel they ran, and `roofline`
        # describes the group with the most GPU time = the step's dominant kernel (rocprofv3 --kernel-trace --stats of this command
        # under profiles/ agrees).  Per launch (averaged over that kernel's launches of a step): algorithmic FLOPs = 2*M*N*K,
        # algorithmic bytes = each distinct operand/output element once (Engine._gemm_bytes).  With d_model = 256 the products
        # sit BELOW the ridge (FLOP/B < peak_flops/peak_bw), so the binding roofline is HBM; the MFMA fraction is reported beside it.
        nsteps_probed = (n_probe_steps if graphed is None else 4)
        roof = roofline_from_probes(probes, nsteps_probed, args.dtype)
        # HBM traffic per launch of the dominant kernel comes from the separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE
        # cannot share a pass, profiles/collect_r02.sh); the file names the kernel it was measured on.
        attach_traffic(roof, args.workload, args.dtype, B)
        per_step = len(probes) // nsteps_probed
        if args.probe_dump:
            rows = []
            for j in range(per_step):
                sel = probes[j::per_step]
                us = 1e3 * sum(a.elapsed_time(b) for a, b, *_ in sel) / len(sel)
                _, _, f, nb, shape, route_ = sel[0]
                rows.append({"launch": j, "kernel": ROUTES.get(route_, route_), "M": shape[0], "N": shape[1], "K": shape[2],
                             "us": round(us, 2), "tflops": round(f / us / 1e6, 1), "gbs": round(nb / us / 1e3, 1)})
            Path(args.probe_dump).write_text(json.dumps(rows, indent=0))
        out = {
            "metric": "train samples/sec (gaze+EEG windows)", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "batch_per_gpu": B, "global_batch": world * B, "C": C, "T": T,
                       "seq_len": eng.S, "d_model": model.cfg.d_model, "layers": model.cfg.num_layers,
                       "step": "fwd(train,dropout)+bwd+allreduce+clip+AdamW" + ("" if args.h2d == "none" else f" + per-step H2D ({args.h2d})"),
                       "parallelism": f"dp{world}",
                       "final_loss": round(loss, 5)},
            "roofline": roof,
        }
        if world == 1 and args.dtype != "f32" and not args.no_f32_leg and not args.no_cpu_baseline and args.h2d == "none":
            out["f32"] = f32_leg(ctor, B, C, T, dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(kw, C, T, Bc=args.cpu_batch)
        emit(out)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
