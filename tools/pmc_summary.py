#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile.sh into the two small files that get committed under profiles/:
  <round>_kernel_stats_<workload>_<dtype>.csv   — the cmx:: rows of the --kernel-trace --stats summary
  <round>_pmc_traffic[_<workload>]_<dtype>.json — HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes;
                                                   gfx950 correction: FETCH_SIZE counts 32-B units → ×2 vs the KiB it
                                                   claims, MI355X_MICROARCH.md §HBM), next to the algorithmic bytes."""
import csv
import glob
import json
import sys
from pathlib import Path

BYTES_PER_POINT = {"sb2006": 13, "sb2006_chen": 13, "cloud_diag": 9, "arg2000_columns": 29, "sb2006_column": 11, "icenuc": 5, "mp0m": 3, "mp1m": 11, "mp1m_lin": 11, "mp1m_column": 11, "mp1m_column_lin": 11, "arg2000": 9, "p3": 9,
                   "p3_split": 9, "p3_selfcol": 7, "sb2006_aos": 15, "sb2006_fields": 11, "mp2m_p3": 20}   # columns in + out


def source_digest():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", Path(__file__).resolve().parent.parent / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.source_digest()


def find(out, sub, suffix):
    hits = glob.glob(f"{out}/{sub}/**/*{suffix}", recursive=True)
    return hits[0] if hits else None


def valu_summary(wl, dt, n, out, rnd):
    """Per-kernel means of the SQ counters of the 4th pass (compute-bound workloads)."""
    path = find(out, "sq", "counter_collection.csv")
    dst = Path("gpurun_out/profiles")
    acc = {}
    for r in csv.DictReader(open(path)):
        if "cmx::" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0]
        d = acc.setdefault(k, {"vgpr": r["VGPR_Count"], "accum_vgpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"],
                               "scratch": r["Scratch_Size"], "counters": {}})
        d["counters"].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    # mean duration of each kernel from the --kernel-trace --stats pass of the same tools/profile.sh run (un-instrumented launches)
    avg_ns = {}
    stats = find(out, "kt", "kernel_stats.csv")
    if stats:
        for r in list(csv.reader(open(stats)))[1:]:
            if r and "cmx::" in r[0]:
                avg_ns[r[0].split("(")[0]] = float(r[3])
    res = {"round": rnd, "workload": wl, "dtype": dt, "points": n, "source_digest": source_digest(),
           "valu_issue_utilisation_formula": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel_avg_ns x 2.4 GHz): the share of "
                                             "one-wave64-VALU-instruction-per-4-cycles issue slots the kernel fills (kernel_avg_ns from the "
                                             "kernel-trace pass of the same profile run; 2.4 GHz = the spec clock).  Simple Float32 instructions "
                                             "issue in 2.4-2.9 cycles on this part (tools/valu_probe.hip, profiles/rNN_probe_valu.txt), so a Float32 "
                                             "kernel can exceed 1 here; valu_frac_of_fastest_issue = the same count x 2.4 cycles (the fastest measured "
                                             "rate) is <= 1 by construction and is what bench.py prints as roofline.frac",
           "kernels": {}}
    tot_insts = tot_ns = 0.0
    for k, d in acc.items():
        c = {name: sum(v) / len(v) for name, v in d["counters"].items()}
        d["counters"] = c
        if "SQ_INSTS_VALU" in c and n:
            d["valu_instructions_per_point"] = c["SQ_INSTS_VALU"] * 64 / n               # per-lane VALU instructions per grid point / state
        if c.get("SQ_WAVE_CYCLES"):
            d["valu_active_over_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
        if k in avg_ns and "SQ_INSTS_VALU" in c:
            d["kernel_avg_ns"] = avg_ns[k]
            d["valu_issue_utilisation"] = c["SQ_INSTS_VALU"] * 4 / (1024 * avg_ns[k] * 1e-9 * 2.4e9)
            d["valu_frac_of_fastest_issue"] = c["SQ_INSTS_VALU"] * 2.4 / (1024 * avg_ns[k] * 1e-9 * 2.4e9)
            tot_insts += c["SQ_INSTS_VALU"]; tot_ns += avg_ns[k]
        res["kernels"][k] = d
    if tot_ns:
        res["valu_issue_utilisation"] = tot_insts * 4 / (1024 * tot_ns * 1e-9 * 2.4e9)     # all kernels of one step together
        res["valu_frac_of_fastest_issue"] = tot_insts * 2.4 / (1024 * tot_ns * 1e-9 * 2.4e9)
    (dst / f"{rnd}_pmc_valu_{wl}_{dt}.json").write_text(json.dumps(res, indent=1))
    print(json.dumps(res))


def main():
    wl, dt, n, out, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    if wl in ("sb2006_column", "mp1m_column", "mp1m_column_lin"):
        n = n // 74 * 74          # bench.py rounds to whole 74-level columns
    if len(sys.argv) > 6 and sys.argv[6] == "valu":
        return valu_summary(wl, dt, n, out, rnd)
    dst = Path("gpurun_out/profiles")
    dst.mkdir(parents=True, exist_ok=True)
    summary = {"round": rnd, "workload": wl, "dtype": dt, "points": n, "source_digest": source_digest(),
               "command": f"tools/profile.sh {wl} {dt} {n}  (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE in separate passes, "
                          "--kernel-trace --stats in a third; PMC passes: bench.py --steps 5 --warmup 1; kernel-trace pass: --steps " + __import__("os").environ.get("KT_STEPS", "40") + " --warmup 5)"}
    stats = find(out, "kt", "kernel_stats.csv")
    if stats:
        rows = list(csv.reader(open(stats)))
        keep = [rows[0]] + [r for r in rows[1:] if r and "cmx::" in r[0]]
        with open(dst / f"{rnd}_kernel_stats_{wl}_{dt}.csv", "w", newline="") as f:
            csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
        main_row = max(keep[1:], key=lambda r: float(r[2]), default=None)
        if main_row:
            summary.update(kernel=main_row[0], calls=int(main_row[1]), avg_ns=float(main_row[3]), min_ns=float(main_row[5]))
    for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        path = find(out, sub, "counter_collection.csv")
        if not path:
            continue
        per_kernel = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == key and "cmx::" in r["Kernel_Name"] and not any(k in r["Kernel_Name"] for k in ("column_sums", "column_partials", "column_finish")):
                per_kernel.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
                summary.setdefault("vgpr", r["VGPR_Count"]); summary.setdefault("sgpr", r["SGPR_Count"])
                summary.setdefault("scratch", r["Scratch_Size"])
        if per_kernel:   # one step = one launch of each kernel of the workload: mean per kernel, summed over the kernels of the STEP.
            # A kernel launched fewer than half as often as the most frequent one belongs to the workload's set-up, not to its step (the 2M + P3
            # line computes its cached log λ column once, with p3_shape_kernel): rounds 1-3 added it in, which is where that line's "1.30 x the
            # algorithmic bytes" came from (profiles/r04_raw_requests_mp2m_p3_f64.txt).
            most = max(len(v) for v in per_kernel.values())
            setup = sorted(k for k, v in per_kernel.items() if 2 * len(v) < most)
            step = {k: v for k, v in per_kernel.items() if k not in setup}
            summary[f"{key}_KiB_per_launch_mean"] = sum(sum(v) / len(v) for v in step.values())
            summary[f"{key}_launches"] = sum(len(v) for v in step.values())
            if setup:
                summary["setup_kernels_excluded"] = [k[:80] for k in setup]
    if "FETCH_SIZE_KiB_per_launch_mean" in summary and "WRITE_SIZE_KiB_per_launch_mean" in summary:
        # gfx950 correction: every TCC→fabric read request is 128 B and FETCH_SIZE tallies it at 64 — checked on known byte counts in six access
        # patterns of this library, 4/8/16 B per lane and the 8-lanes-per-state reads of the collision kernels (profiles/r04_traffic_calibration.txt);
        # WRITE_SIZE is exact in all seven write patterns tried there
        fetch = summary["FETCH_SIZE_KiB_per_launch_mean"] * 1024 * 2
        write = summary["WRITE_SIZE_KiB_per_launch_mean"] * 1024
        algo = n * BYTES_PER_POINT[wl] * (4 if dt == "f32" else 8)
        summary.update(fetch_bytes_corrected=fetch, write_bytes=write, hbm_bytes_per_launch=fetch + write,
                       algorithmic_bytes_per_launch=algo, traffic_over_algorithmic=(fetch + write) / algo)
    tag = "" if wl == "sb2006" else f"_{wl}"
    (dst / f"{rnd}_pmc_traffic{tag}_{dt}.json").write_text(json.dumps(summary, indent=1))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
