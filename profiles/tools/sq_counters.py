"""Per-kernel means of SQ counters from several rocprofv3 --pmc passes (one directory per pass): which pipe a kernel is bound by.
Usage: sq_counters.py <out.json> <pass_dir> [<pass_dir> ...]"""
import collections
import csv
import glob
import json
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    out = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[2:]:
        fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[k]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": len(cs["us"])} for k, cs in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    names = sorted({c for cs in res.values() for c in cs if c not in ("us", "launches")})
    top = sorted(res.items(), key=lambda kv: -kv[1]["us"] * kv[1]["launches"])[:14]
    for k, cs in top:
        print(f"{k[:60]:60s} us={cs['us']:8.1f}")
        for c in names:
            if c in cs:
                print(f"      {c:36s} {cs[c]:16.0f}")


if __name__ == "__main__":
    main()
