"""pmc_by_kernel.json (profiles/tools/pmc_by_kernel.py) -> the per-kernel records bench.py reads for `roofline.traffic`:
a list of {"kernel": <name prefix as bench.py's ROUTES spell it>, "hbm_bytes_per_launch", "fetch...", "write...", "collected"}; one
record per kernel family, launch-weighted over its template instantiations.  Usage: pmc_for_bench.py <pmc_by_kernel.json> <out.json>"""
import datetime
import json
import sys

FAMILIES = ["gemm_nt_wide_kernel", "ffn_chain_kernel", "attn_block_fwd_kernel", "rs_gemm_kernel", "gemm_nt_kernel", "gemm_nt_row_kernel"]


def main():
    src, out = sys.argv[1:3]
    res = json.load(open(src))
    recs = []
    for fam in FAMILIES:
        sel = {k: v for k, v in res.items() if isinstance(v, dict) and k.startswith(fam + "<")}
        n = sum(v["launches_sampled"] for v in sel.values())
        if not n:
            continue
        mean = lambda key: sum(v[key] * v["launches_sampled"] for v in sel.values()) / n
        recs.append({"kernel": fam, "instantiations": sorted(sel), "launches_sampled": n,
                     "fetch_bytes_per_launch": round(mean("fetch_bytes_per_launch")),
                     "write_bytes_per_launch": round(mean("write_bytes_per_launch")),
                     "hbm_bytes_per_launch": round(mean("hbm_bytes_per_launch")),
                     "collected": datetime.date.today().isoformat(), "corrections": res.get("_corrections")})
    json.dump(recs, open(out, "w"), indent=1)
    for r in recs:
        print(r["kernel"], r["launches_sampled"], r["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
