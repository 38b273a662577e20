"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB, and FETCH_SIZE reports half of
the bytes of wide coalesced reads (x2).  Usage: pmc_summary.py <fetch_dir> <write_dir> <kernel-substring> <out.json>"""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    return list(csv.DictReader(open(f)))


def main():
    fetch_dir, write_dir, needle, out = sys.argv[1:5]
    acc = collections.defaultdict(list)
    for d, name in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        for r in load(d):
            if needle in r["Kernel_Name"] and r["Counter_Name"] == name:
                acc[name].append(float(r["Counter_Value"]))
                acc["us_" + name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n = len(acc["FETCH_SIZE"])
    fetch = sum(acc["FETCH_SIZE"]) / n * 1024 * 2
    write = sum(acc["WRITE_SIZE"]) / max(1, len(acc["WRITE_SIZE"])) * 1024
    res = {"kernel": needle, "launches_sampled": n, "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
           "hbm_bytes_per_launch": round(fetch + write), "avg_launch_us_under_pmc": round(sum(acc["us_FETCH_SIZE"]) / n, 2),
           "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 wide-read correction); WRITE_SIZE KiB x1024; separate --pmc passes"}
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()
