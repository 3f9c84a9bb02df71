#!/usr/bin/env python3
"""Tabulate the rocprofv3 passes of tools/traffic_calib.sh: per calibration kernel, each counter's per-launch mean and — for FETCH_SIZE /
WRITE_SIZE — the ratio of the bytes it claims to the bytes the kernel is known to move.  The ratios are the pattern-specific corrections
tools/pmc_summary.py applies (MI355X_MICROARCH.md §HBM: only the 16-B-per-lane streaming read is calibrated there)."""
import csv
import glob
import re
import sys


def main():
    out = sys.argv[1]
    known = None
    for line in open(f"{out}/fetch.log", errors="replace"):
        m = re.search(r"bytes per launch: (\d+)", line)
        if m:
            known = int(m.group(1))
    acc = {}
    for path in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if not k.startswith("calib_") or k == "calib_fill":
                continue
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(f"known bytes per launch: {known}   (rocprofv3 --pmc, one pass per counter set; means over the launches of each kernel)")
    print(f"{'kernel':38s} {'FETCH_SIZE KiB':>15s} {'x1024/known':>12s} {'WRITE_SIZE KiB':>15s} {'x1024/known':>12s} {'RDREQ':>11s} {'RDREQ_32B':>10s} {'BUBBLE':>9s} "
          f"{'B/RDREQ':>8s} {'WRREQ':>11s} {'WRREQ_64B':>10s} {'B/WRREQ':>8s}")
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        f, w = c.get("FETCH_SIZE", float("nan")), c.get("WRITE_SIZE", float("nan"))
        rd, rd32, bub = c.get("TCC_EA0_RDREQ_sum", float("nan")), c.get("TCC_EA0_RDREQ_32B_sum", float("nan")), c.get("TCC_BUBBLE_sum", float("nan"))
        wr, wr64 = c.get("TCC_EA0_WRREQ_sum", float("nan")), c.get("TCC_EA0_WRREQ_64B_sum", float("nan"))
        is_read = "_read_" in k
        print(f"{k:38s} {f:15.0f} {f * 1024 / known:12.3f} {w:15.0f} {w * 1024 / known:12.3f} {rd:11.0f} {rd32:10.0f} {bub:9.0f} "
              f"{(known / rd if is_read and rd else float('nan')):8.1f} {wr:11.0f} {wr64:10.0f} {(known / wr if not is_read and wr else float('nan')):8.1f}")


if __name__ == "__main__":
    main()
