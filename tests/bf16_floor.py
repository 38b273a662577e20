"""Storage-format floor of the 16-bit compute modes.  CPU only, no GPU, no reference needed.

Question: how far from the reference's fp32 logits does a path land that does ALL arithmetic in fp32 but keeps its GEMM weights and
every activation it stores between kernels in bf16 (or fp16) -- i.e. the HIP path's storage format with perfect kernels?
Method: the CPU oracle (pinned to the reference by tests/golden) with oracle.dual_eeg_oracle.STORAGE_ROUND rounding at exactly the
sites where the HIP engine writes compute-dtype tensors and the packed GEMM weights rounded once, on the same 512
window pairs per BASELINE config as tests/test_gpu_logits512.py, against the reference logits in tests/golden/logits512.npz.

Usage:  python tests/bf16_floor.py [bf16|fp16] -> profiles/r03_<dtype>_storage_floor.json
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from eyegaze_multimodal_amd.data import randn_windows  # noqa: E402
from oracle import dual_eeg_oracle as O  # noqa: E402
from tests.helpers import GOLDEN, load_golden  # noqa: E402

NAMES = ["cfg1_a1_2class", "cfg2_concat", "cfg3_xattn", "cfg5_a2_spec", "a5_full"]
# parameters the engine re-casts to the compute dtype each step (Engine._pack_body / tokens.pack); everything else stays fp32:
# biases, LayerNorm / InstanceNorm affine, cls_token, the two [d -> num_classes] output layers
CAST_SUFFIXES = ("temporal_conv.convs.0.weight", "temporal_conv.convs.1.weight", "pos_embed.pos_embed.weight", "q_proj.weight",
                 "k_proj.weight", "v_proj.weight", "out_proj.weight", "ffn.linear1.weight", "ffn.linear2.weight",
                 "symmetric_fusion.proj.weight", "classifier.0.weight", "ibs_classifier.0.weight", "spec_conv.3.weight",
                 "proj.0.weight", "proj.3.weight", "bottleneck.0.weight", "bottleneck.3.weight", "type_embedding",
                 "ibs_generator.proj.0.weight", "ibs_generator.proj.3.weight")


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dt]
    rnd = lambda t_: t_.to(tdt).to(torch.float32)
    torch.set_num_threads(8)
    z = np.load(GOLDEN / "logits512.npz", allow_pickle=False)
    n, seed = int(z["n"]), int(z["seed"])
    res = {"dtype": dt, "n": n, "model": "fp32 arithmetic; GEMM weights and every stored activation rounded to " + dt +
           " (oracle.dual_eeg_oracle.STORAGE_ROUND at the HIP engine's store sites)", "configs": {}}
    for name in NAMES:
        _, kw, cfg, sd = load_golden(name)
        sd_r = {k: (rnd(v) if k.endswith(CAST_SUFFIXES) else v) for k, v in sd.items()}
        x1, x2, _ = randn_windows(n, cfg.in_channels, 1024, seed=seed, num_classes=cfg.num_classes)
        ref = z[name + "/logits"]
        out = {}
        for label, state, hook in (("weights_only", sd_r, None), ("weights_and_activations", sd_r, lambda t_, site: rnd(t_)),
                                   ("fp32_oracle", sd, None)):
            O.STORAGE_ROUND = hook
            try:
                with torch.no_grad():
                    got = torch.cat([O.forward(x1[i:i + 64], x2[i:i + 64], state, cfg)["logits"] for i in range(0, n, 64)]).numpy()
            finally:
                O.STORAGE_ROUND = None
            err = np.abs(got - ref)
            decided = z[name + "/margin"] > 4e-2
            out[label] = {"max_abs_dlogit": float(err.max()), "p99_abs_dlogit": float(np.quantile(err, 0.99)),
                          "mean_abs_dlogit": float(err.mean()),
                          "argmax_agree_decided": float((got.argmax(-1)[decided] == z[name + "/argmax"][decided]).mean())}
        res["configs"][name] = out
        print(name, {k: round(v["max_abs_dlogit"], 6) for k, v in out.items()}, flush=True)
    hip = REPO / "profiles" / "r02_parity_table.json"
    res["note"] = ("max_abs_dlogit of 'weights_and_activations' is the floor a perfect-kernel implementation of this storage layout reaches; "
                   "the HIP path's measured maxima on the same samples are in profiles/r03_parity_table.json (bf16) and the gates of "
                   "tests/test_gpu_logits512.py are set from the two")
    outp = REPO / "profiles" / f"r03_{dt}_storage_floor.json"
    outp.write_text(json.dumps(res, indent=1))
    print("wrote", outp)


if __name__ == "__main__":
    main()
