"""Per-kernel HBM traffic of a whole step from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB, and FETCH_SIZE reports half of the
bytes of wide coalesced reads (x2).  Usage: pmc_by_kernel.py <fetch_dir> <write_dir> <out.json> [<kernel prefix> <dominant.json>]
With a kernel prefix (e.g. gemm_nt_wide_kernel) the launch-weighted mean over every instantiation of that kernel is written to
<dominant.json> in the form bench.py reads for `roofline.traffic`."""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    return list(csv.DictReader(open(f)))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, name in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        for r in load(d):
            if r["Counter_Name"] == name:
                k = short(r["Kernel_Name"])
                acc[k][name].append(float(r["Counter_Value"]))
                acc[k]["us_" + name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res = {}
    for k, v in acc.items():
        n = max(1, len(v["FETCH_SIZE"]))
        fetch = sum(v["FETCH_SIZE"]) / n * 1024 * 2
        write = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024
        us = sum(v["us_FETCH_SIZE"]) / n
        res[k] = {"launches_sampled": n, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                  "hbm_bytes_per_launch": round(fetch + write), "avg_launch_us_under_pmc": round(us, 2),
                  "total_us_sampled": round(us * n, 1)}
    res["_corrections"] = "FETCH_SIZE KiB x1024 x2 (gfx950 wide-read correction); WRITE_SIZE KiB x1024; separate --pmc passes"
    json.dump(res, open(out, "w"), indent=1)
    if len(sys.argv) >= 6:
        prefix, dom_out = sys.argv[4:6]
        sel = {k: v for k, v in res.items() if k.startswith(prefix)}
        n = sum(v["launches_sampled"] for v in sel.values())
        if n:
            mean = lambda key: sum(v[key] * v["launches_sampled"] for v in sel.values()) / n
            json.dump({"kernel": prefix, "instantiations": sorted(sel), "launches_sampled": n,
                       "fetch_bytes_per_launch": round(mean("fetch_bytes_per_launch")),
                       "write_bytes_per_launch": round(mean("write_bytes_per_launch")),
                       "hbm_bytes_per_launch": round(mean("hbm_bytes_per_launch")),
                       "corrections": res["_corrections"]}, open(dom_out, "w"), indent=1)
    for k, v in sorted(((k, v) for k, v in res.items() if not k.startswith("_")), key=lambda kv: -kv[1]["total_us_sampled"])[:16]:
        print(f"{k[:48]:48s} n={v['launches_sampled']:5d} us={v['avg_launch_us_under_pmc']:8.1f} fetch={v['fetch_bytes_per_launch']/1e6:8.1f}MB "
              f"write={v['write_bytes_per_launch']/1e6:8.1f}MB")


if __name__ == "__main__":
    main()
