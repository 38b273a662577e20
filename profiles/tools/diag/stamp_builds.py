"""Diagnostic builds with in-kernel s_memtime stamps (NOT part of the product: they are generated from the product sources into
scratch/, built into scratch/libstamp_*.so and loaded by pointing eyegaze_multimodal_amd._lib.LIB_PATH at them).
  python profiles/tools/diag/stamp_builds.py build          # on the build box (hipcc, no GPU needed)
  python profiles/tools/diag/stamp_builds.py run-ffn|run-wide|run-attn   # on the MI355X
The stamps are per wave and relative to the wave's own start (s_memtime bases differ between XCDs); every stamp is
`s_memtime; s_waitcnt lgkmcnt(0)` fenced by sched_barriers (cdna_hip_programming.md, In-kernel stamps)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[3]
CS = ROOT / "eyegaze_multimodal_amd" / "csrc"
SCR = ROOT / "scratch"
HIPCC = "/opt/rocm/bin/hipcc"


def rep(s, a, b):
    assert a in s, a[:60]
    return s.replace(a, b, 1)


def gen_ffn():
    s = (CS / "ffn.hip").read_text()
    s = rep(s, '#include "common.h"', """#include "../eyegaze_multimodal_amd/csrc/common.h"
__device__ unsigned long long eg_stamps[416][4][8];
#define STAMPV(v_) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(v_) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int eg_debug_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eg_stamps), sizeof(eg_stamps)); }
""")
    s = rep(s, "  const int nch = p.F / FC;\n", "  const int nch = p.F / FC;\n  unsigned long long T0, Ta, Tb, Tc, Td, Te, acc_p1 = 0, acc_e1 = 0, acc_bar = 0, acc_st = 0, acc_p2 = 0, Tloop, Tend;\n  STAMPV(T0);\n")
    s = rep(s, "  for (int c = 0; c < nch; ++c) {\n    char* const hc = hb + (c & 1) * F_HT;", "  STAMPV(Tloop);\n  for (int c = 0; c < nch; ++c) {\n    STAMPV(Ta);\n    char* const hc = hb + (c & 1) * F_HT;")
    s = rep(s, "    // ---- epilogue 1 (MFMA layout: lane holds 4 consecutive hidden columns of row l15) -> 16-bit chunk image in LDS ----\n    // Vector-instruction", "    STAMPV(Tb); acc_p1 += Tb - Ta;\n    // ---- epilogue 1 (MFMA layout: lane holds 4 consecutive hidden columns of row l15) -> 16-bit chunk image in LDS ----\n    // Vector-instruction")
    s = rep(s, "    __syncthreads();        // chunk c is complete in LDS; nobody reads buffer (c+1)&1 (chunk c-1) any more\n", "    STAMPV(Tc); acc_e1 += Tc - Tb;\n    __syncthreads();\n    STAMPV(Td); acc_bar += Td - Tc;\n")
    s = rep(s, "    // ---- product 2: acc2[i][j] += sum_h W2[col][h] * H[row][h] over the chunk's 128 hidden columns ----\n", "    STAMPV(Te); acc_st += Te - Td;\n")
    s = rep(s, "      else if (c + 1 < nch) req_w2(c + 1, s - 2, s & 1);\n    }\n  }\n", "      else if (c + 1 < nch) req_w2(c + 1, s - 2, s & 1);\n    }\n    STAMPV(Ta); acc_p2 += Ta - Te;\n  }\n  STAMPV(Tend);\n")
    i = s.index("template <typename T>\nstatic int ffn_launch")
    k_end = s.rindex("}\n", 0, i)
    s = s[:k_end] + """  { unsigned long long Tz; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMPV(Tz);
    if (lane == 0 && blockIdx.x < 416) { unsigned long long* o = eg_stamps[blockIdx.x][wn];
      o[0] = Tloop - T0; o[1] = acc_p1; o[2] = acc_e1; o[3] = acc_bar; o[4] = acc_st; o[5] = acc_p2; o[6] = Tz - Tend; o[7] = Tz - T0; } }
""" + s[k_end:]
    (SCR / "ffn_stamp.hip").write_text(s)


def gen_wide():
    s = (CS / "widegemm.hip").read_text()
    s = rep(s, '#include "common.h"', '#include "../eyegaze_multimodal_amd/csrc/common.h"')
    s = rep(s, "#include <stdlib.h>\n", """#include <stdlib.h>
__device__ unsigned long long eg_stamps[208][8][8];
#define STAMP(i_) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); ts[i_] = t_; } while (0)
extern "C" int eg_debug_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eg_stamps), sizeof(eg_stamps)); }
""")
    s = rep(s, "  const int nk = p.K >> 6;\n", "  const int nk = p.K >> 6;\n  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n  STAMP(0);\n")
    s = rep(s, "  const int na = wave < 4 ? 3 : 2;", "  asm volatile(\"\" :: \"v\"(asrc[0]), \"v\"(asrc[1]), \"v\"(asrc[2]), \"v\"(wsrc[0]), \"v\"(wsrc[3]));\n  STAMP(7);\n  const int na = wave < 4 ? 3 : 2;")
    s = rep(s, "  issue(0, 0);\n  if (nk > 1) issue(1, 1);\n  int slot = 0;", "  issue(0, 0);\n  if (nk > 1) issue(1, 1);\n  STAMP(1);\n  int slot = 0;")
    s = rep(s, "    if (kt + 2 < nk) issue(kt + 2, slot == 0 ? 2 : slot - 1);", "    if (kt == 0) STAMP(2);\n    if (kt == 1) STAMP(6);\n    if (kt + 2 < nk) issue(kt + 2, slot == 0 ? 2 : slot - 1);")
    s = rep(s, "  asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\");       // every wave has left the ring: it becomes epilogue scratch\n", "  asm volatile(\"s_waitcnt lgkmcnt(0)\\n\\ts_barrier\" ::: \"memory\");\n  STAMP(3);\n")
    s = rep(s, "#pragma unroll\n  for (int i = 0; i < 5; ++i) {\n    const int m = m0 + 80 * wm + 16 * i + er;\n#pragma unroll\n    for (int j = 0; j < 4; ++j) *(f32x4*)(timg", "  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  STAMP(4);\n#pragma unroll\n  for (int i = 0; i < 5; ++i) {\n    const int m = m0 + 80 * wm + 16 * i + er;\n#pragma unroll\n    for (int j = 0; j < 4; ++j) *(f32x4*)(timg")
    s = rep(s, "      store8(p.C + coff + 8, v + 8);\n    }\n  }\n}", "      store8(p.C + coff + 8, v + 8);\n    }\n  }\n  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  STAMP(5);\n  if (lane == 0 && blockIdx.x < 208) {\n#pragma unroll\n    for (int i = 0; i < 8; ++i) eg_stamps[blockIdx.x][wave][i] = ts[i];\n  }\n}")
    (SCR / "widegemm_stamp.hip").write_text(s)


def gen_attn():
    s = (CS / "attention.hip").read_text()
    s = rep(s, '#include "common.h"', """#include "../eyegaze_multimodal_amd/csrc/common.h"
__device__ unsigned long long eg_stamps[2048][4][8];
#define STAMPV(v_) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(v_) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int eg_debug_stamps(void* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eg_stamps), sizeof(eg_stamps)); }
""")
    k0 = s.index("void attn_bwd1_kernel(")
    head, body = s[:k0], s[k0:]
    body = rep(body, "  constexpr int NKT = SP / 16;\n", "  constexpr int NKT = SP / 16;\n  unsigned long long T0, T1, T2, T3, T4, T5;\n  STAMPV(T0);\n")
    body = rep(body, "  rows_store<NCH>(doimg, rd, SP, first);\n", "  rows_store<NCH>(doimg, rd, SP, first);\n  STAMPV(T5);\n")
    body = rep(body, "  const uint32_t Sp2 = (uint32_t)((S + 1) & ~1);\n  __syncthreads();\n", "  const uint32_t Sp2 = (uint32_t)((S + 1) & ~1);\n  __syncthreads();\n  STAMPV(T1);\n")
    body = rep(body, "  // ---- dQ: role 0 finishes query tiles [0, NKT/2), role 1 the rest; each hands the other its partial of the other's tiles ----\n  __syncthreads();", "  STAMPV(T2);\n  __syncthreads();\n  STAMPV(T3);")
    e0 = body.index("// ------------------------------------------------------------------------------------------------\n// Exact-fp32 attention")
    k_end = body.rindex("}\n", 0, e0)
    body = body[:k_end] + """  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMPV(T4);
  if (lane == 0 && blockIdx.x < 2048) { unsigned long long* o = eg_stamps[blockIdx.x][wave];
    o[0] = T5 - T0; o[1] = T1 - T5; o[2] = T2 - T1; o[3] = T3 - T2; o[4] = T4 - T3; o[5] = T4 - T0; }
""" + body[k_end:]
    (SCR / "attention_stamp.hip").write_text(head + body)


def build():
    SCR.mkdir(exist_ok=True)
    gen_ffn()
    gen_wide()
    gen_attn()
    objs = sorted(str(o) for o in CS.glob("*.o"))
    for name, repl in (("ffn_stamp", "ffn.o"), ("widegemm_stamp", "widegemm.o"), ("attention_stamp", "attention.o")):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-I", str(ROOT / "include"), "-c",
                        str(SCR / f"{name}.hip"), "-o", str(SCR / f"{name}.o")], check=True)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(SCR / f"lib{name}.so"), str(SCR / f"{name}.o"),
                        *[o for o in objs if not o.endswith("/" + repl)]], check=True)
        print(SCR / f"lib{name}.so")


def run_ffn():
    import ctypes as C
    sys.path.insert(0, str(ROOT))
    from eyegaze_multimodal_amd import _lib as L
    L.LIB_PATH = (SCR / "libffn_stamp.so").resolve()
    import numpy as np
    import torch
    src = (ROOT / "tests" / "diag_ffn_bench.py").read_text().split("\nfor M, F, mode, p in")[0]
    ns = {"__file__": str(ROOT / "tests" / "diag_ffn_bench.py"), "__name__": "diag"}
    exec(compile(src, "diag_ffn_bench", "exec"), ns)
    names = ["prologue (A tile, touch)", "product 1", "epilogue 1", "barrier wait", "H store issue", "product 2", "epilogue 2 + drain", "lifetime"]
    for mode, p in (("fwd", 0.1), ("fwd", 0.0), ("bwd_bits", 0.0)):
        two, one, keep = ns["variants"](33280, 1024, mode, p)
        for _ in range(3):
            one(); torch.cuda.synchronize()
        buf = np.zeros((416, 4, 8), dtype=np.uint64)
        L.lib().eg_debug_stamps(C.c_void_p(buf.ctypes.data))
        t = buf.astype(np.int64)
        print(f"eg_ffn_chain M=33280 F=1024 {mode} p={p}: cycles per wave, summed over the 8 chunks (416 workgroups x 4 waves)")
        for i, n in enumerate(names):
            col = t[:, :, i]
            print(f"   {n:26s} median {int(np.median(col)):7d}  p10 {int(np.percentile(col, 10)):7d}  p90 {int(np.percentile(col, 90)):7d}")


def run_wide():
    import ctypes as C
    sys.path.insert(0, str(ROOT))
    from eyegaze_multimodal_amd import _lib as L
    L.LIB_PATH = (SCR / "libwidegemm_stamp.so").resolve()
    import numpy as np
    import torch
    from tests.test_gpu_ops import DEV, gemm_nt
    M, N = 33280, 256
    names = ["start", "issued 2 stages", "stage 0 visible", "ring left (barrier)", "epilogue operand landed", "end (stores drained)", "stage 1 visible", "addresses ready"]
    order = [0, 7, 1, 2, 6, 3, 4, 5]
    for K, resid in ((256, False), (256, True), (768, False)):
        g = torch.Generator().manual_seed(1)
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
        W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
        R = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV) if resid else None
        out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
        big = torch.zeros(64 * 1024 * 1024, device=DEV)
        for _ in range(3):
            big.add_(1.0); torch.cuda.synchronize()
            gemm_nt(A, W, M, N, K, L.EG_BF16, out=out, residual=R)
        buf = np.zeros((208, 8, 8), dtype=np.uint64)
        L.lib().eg_debug_stamps(C.c_void_p(buf.ctypes.data))
        t = buf.astype(np.int64)
        rel = t - t[:, :, :1]
        print(f"gemm_nt_wide M={M} N={N} K={K} residual={resid}: cycles since the wave's start (208 workgroups x 8 waves)")
        for i in order:
            col = rel[:, :, i]
            print(f"   {names[i]:26s} median {int(np.median(col)):7d}  p10 {int(np.percentile(col, 10)):7d}  p90 {int(np.percentile(col, 90)):7d}")


def run_attn():
    import ctypes as C
    sys.path.insert(0, str(ROOT))
    from eyegaze_multimodal_amd import _lib as L
    L.LIB_PATH = (SCR / "libattention_stamp.so").resolve()
    import numpy as np
    import torch
    from eyegaze_multimodal_amd._lib import call, ptr
    from tests.test_gpu_ops import DEV, dev_state
    NB, S, H, D = 512, 65, 8, 256
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(NB * S, 3 * D, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    ctx = torch.zeros(NB * S, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(NB, H, S, device=DEV)
    st = dev_state(seed=5)
    call("eg_attention_fwd", ptr(qkv), ptr(ctx), ptr(lse), NB, S, H, 0, L.EG_BF16, 0.1, 21, ptr(st), 0)
    dctx = (torch.randn(NB * S, D, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    dqkv = torch.zeros_like(qkv)
    for _ in range(3):
        call("eg_attention_bwd", ptr(qkv), ptr(ctx), ptr(dctx), ptr(lse), ptr(dqkv), NB, S, H, 0, L.EG_BF16, 0.1, 21, ptr(st), 0)
        torch.cuda.synchronize()
    buf = np.zeros((2048, 4, 8), dtype=np.uint64)
    L.lib().eg_debug_stamps(C.c_void_p(buf.ctypes.data))
    t = buf.astype(np.int64)
    names = ["request everything, store q, k, dO images", "delta = rowsum(dO * O), barrier", "key-tile loop", "wait at the exchange barrier", "dQ exchange + stores drained", "lifetime"]
    for role in (0, 1):
        print(f"attn_bwd1 NB={NB} S={S} p=0.1, waves of role {role} (key tiles {'0, 2, 4' if role == 0 else '1, 3'}): cycles per wave")
        for i, n in enumerate(names):
            col = t[:, role::2, i]
            print(f"   {n:44s} median {int(np.median(col)):7d}  p10 {int(np.percentile(col, 10)):7d}  p90 {int(np.percentile(col, 90)):7d}")


if __name__ == "__main__":
    {"build": build, "run-ffn": run_ffn, "run-wide": run_wide, "run-attn": run_attn}[sys.argv[1]]()
