#!/usr/bin/env python3
"""Turn rocprofv3 output (gpurun_out/, scratch) into the small summaries committed under profiles/.

  python profiles/summarize.py stats   <*_kernel_stats.csv> <out.md> "<title>" "<command line>" ["<note>"]
  python profiles/summarize.py traffic <fetch *_counter_collection.csv> <write *_counter_collection.csv> <out.json>

`traffic`: per-kernel HBM bytes per launch from two separate `--pmc` passes (FETCH_SIZE and WRITE_SIZE do not fit one pass),
as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide (16 B / lane) coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16 B / lane stores.
"""
import csv
import json
import sys
from collections import defaultdict


def short(name: str) -> str:
    return name if len(name) <= 80 else name[:80]


def stats(src, dst, title, cmd, note=""):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nCommand (on the MI355X box): `{cmd}`\n\n")
        if note:
            f.write(note + "\n\n")
        f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                    f"{float(r['Percentage']):.2f} |\n")


def per_kernel(path, counter):
    # a dispatch may be split over several rows (one per counter instance / XCC): sum per dispatch first
    disp = defaultdict(float)
    name_of = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        disp[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name_of[r["Dispatch_Id"]] = r["Kernel_Name"]
    acc = defaultdict(list)
    for d, v in disp.items():
        acc[name_of[d]].append(v)
    return acc


def traffic(fetch_csv, write_csv, dst):
    fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in sorted(set(fe) | set(wr)):
        if "amid::" not in k and "amid_rt" not in k:
            continue
        f = fe.get(k, [0.0])
        w = wr.get(k, [0.0])
        fetch_b = 2.0 * 1024.0 * sum(f) / len(f)            # KiB -> bytes, x2: gfx950 correction for wide coalesced reads
        write_b = 1024.0 * sum(w) / len(w)
        out[short(k)] = {"launches": len(f), "fetch_bytes_per_launch": round(fetch_b), "write_bytes_per_launch": round(write_b),
                         "hbm_bytes_per_launch": round(fetch_b + write_b)}
    json.dump({"note": "FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE x 1024, averaged per launch; separate --pmc passes; "
                       "counts the memory-side requests of L2 (Infinity-Cache hits included)", "kernels": out}, open(dst, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(*sys.argv[2:])
    elif sys.argv[1] == "traffic":
        traffic(*sys.argv[2:])
    else:
        raise SystemExit(__doc__)
