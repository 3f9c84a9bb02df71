#!/usr/bin/env python3
"""Generate the measured tables and sentences of DESIGN.md from the committed evidence under profiles/ — so that the document cannot
quote a number the evidence does not hold ("DESIGN §5 misquotes its own evidence", "0.666-vs-0.707 discrepancies").

    tools/gen_design_tables.py [--round rNN]            print the generated blocks
    tools/gen_design_tables.py --write [--round rNN]    rewrite the blocks between the markers in DESIGN.md
    tools/gen_design_tables.py --check [--round rNN]    exit 1 if DESIGN.md's blocks differ from what the evidence generates

Markers in DESIGN.md:  <!-- BEGIN GENERATED <name> --> … <!-- END GENERATED <name> -->   with <name> ∈ {performance, parity}.
Sources (round rNN = the newest round that has a bench directory unless --round is given):
    profiles/bench_rNN/<workload>_<dtype>.json          bench.py lines (HIP-event mean of the timed launches, same session)
    profiles/rNN_kernel_stats_<workload>_<dtype>.csv    rocprofv3 --kernel-trace --stats (mean / min over ≥ 40 profiled launches)
    profiles/rNN_pmc_traffic[_<workload>]_<dtype>.json  FETCH_SIZE / WRITE_SIZE passes (HBM bytes over algorithmic bytes)
    profiles/rNN_pmc_valu_<workload>_<dtype>.json       SQ_INSTS_VALU pass (instructions per point, VALU issue utilisation)
    profiles/rNN_parity_report.json                     rows written by tests/parity.py during the -m gpu session
"""
import csv
import json
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
PROF = REPO / "profiles"
HBM_PEAK = 8000.0
ORDER = ["sb2006", "sb2006_chen", "sb2006_column", "sb2006_aos", "sb2006_fields", "mp0m", "cloud_diag", "icenuc", "mp1m", "mp1m_lin", "mp1m_column", "mp1m_column_lin", "arg2000", "arg2000_columns",
         "p3_split", "p3", "p3_selfcol", "mp2m_p3"]


def latest_round():
    rounds = sorted(int(m.group(1)) for p in PROF.glob("bench_r*") if (m := re.fullmatch(r"bench_r(\d+)", p.name)))
    return f"r{rounds[-1]:02d}" if rounds else None


def bench_lines(rnd):
    out = {}
    for p in sorted((PROF / f"bench_{rnd}").glob("*.json")):
        try:
            lines = [l for l in p.read_text().splitlines() if l.strip().startswith("{")]
            d = json.loads(lines[-1])
        except (OSError, ValueError, IndexError):
            continue
        out[p.stem] = d
    return out


def kernel_stats(rnd, wl, dt):
    """(sum of the mean durations of the cmx kernels of one step [ms], sum of the minima, calls of the main kernel) or None."""
    p = PROF / f"{rnd}_kernel_stats_{wl}_{dt}.csv"
    if not p.exists():
        return None
    rows = [r for r in list(csv.reader(open(p)))[1:] if r and "cmx::" in r[0] and not any(k in r[0] for k in ("column_sums", "column_partials", "column_finish"))]
    if not rows:
        return None
    main = max(rows, key=lambda r: float(r[2]))
    calls = int(float(main[1]))
    same = [r for r in rows if int(float(r[1])) == calls]      # the kernels launched once per step
    return sum(float(r[3]) for r in same) * 1e-6, sum(float(r[5]) for r in same) * 1e-6, calls


def load(p):
    try:
        return json.loads(p.read_text())
    except (OSError, ValueError):
        return None


def performance_block(rnd):
    lines = bench_lines(rnd)
    rows = []
    for key, d in lines.items():
        m = re.fullmatch(r"(.+)_(f32|f64)", key)
        if not m:
            continue
        wl, dt = m.groups()
        r = d["roofline"]
        n = d["config"]["points_per_gpu"]
        bpp = r.get("bytes_per_point") or r.get("hbm", {}).get("bytes_per_point")
        ks = kernel_stats(rnd, wl, dt)
        tr = load(PROF / f"{rnd}_pmc_traffic{'' if wl == 'sb2006' else '_' + wl}_{dt}.json")
        pv = load(PROF / f"{rnd}_pmc_valu_{wl}_{dt}.json")
        kms = r["kernel_ms"]
        rk = d.get("ranks_kernel_ms") or {}
        same = (rk.get("same_buffer") or [kms])[0]
        rot = (rk.get("rotating") or [None])[0]
        frac_b = n * bpp / (kms * 1e-3) / 1e9 / HBM_PEAK
        frac_p = n * bpp / (ks[0] * 1e-3) / 1e9 / HBM_PEAK if ks else None
        ipp = util = active = None
        if pv and pv.get("points") == n:
            insts = [k.get("counters", {}).get("SQ_INSTS_VALU") for k in pv["kernels"].values()]
            if insts and all(v is not None for v in insts):
                ipp = sum(insts) * 64 / n
            util = pv.get("valu_issue_utilisation")
            act = [k.get("counters", {}).get("SQ_ACTIVE_INST_VALU") for k in pv["kernels"].values()]
            if act and all(v is not None for v in act):
                active = sum(act) * 4 / 1024          # clocks per SIMD in which the VALU was executing (the counter ticks once per 4 clocks)
        cold = (d.get("cold") or {}).get("first5_ms")
        tel = d.get("telemetry") or {}
        pw = [w for w in (tel.get("package_power_w") or []) if w is not None]
        clk = (sum(tel["sclk_mhz"]) / len(tel["sclk_mhz"]), (sum(pw) / len(pw)) if pw else None) if tel.get("sclk_mhz") else None
        sus = tel.get("sustained_ms_per_step")
        sus = (sum(sus[1:]) / len(sus[1:])) if sus and len(sus) > 1 else None      # the first batch is the ramp
        rows.append((ORDER.index(wl) if wl in ORDER else 99, dt, wl, n, bpp, same, rot, d.get("value_uses"), cold[0] if cold else None, ks, frac_b, frac_p,
                     tr.get("traffic_over_algorithmic") if tr and tr.get("points") == n else None, ipp, util, r.get("bound", "hbm"), clk, sus,
                     (active / (clk[0] * 1e6 * ks[0] * 1e-3)) if active and clk and ks and dt == "f64" else None))
    rows.sort()
    fmt = lambda v, f: "—" if v is None else f % v  # noqa: E731
    out = [f"Measured in ONE session on one box (round {rnd}; `profiles/bench_{rnd}/`, `profiles/{rnd}_kernel_stats_*.csv`, "
           f"`profiles/{rnd}_pmc_*.json`).  same / rotating = HIP-event mean per launch of the two timed regions of `bench.py`: every launch on ONE "
           "buffer set, and launches cycling through 4 disjoint buffer sets (`--rotate 4`; the working set of the sub-millisecond lines then "
           "exceeds the 256-MiB Infinity Cache several times over, and every launch sweeps pages another launch touched last) — `value` uses the "
           "rotating time when it is more than 5 % slower.  first = the very first timed launch after the inputs are generated.  rocprof = mean (min) "
           "over the profiled launches of `rocprofv3 --kernel-trace --stats` in the same session (1000 timed launches of the sweep kernels plus bench.py's settle and warm-up launches, 6–10 for the P3 lines).  HBM frac = algorithmic bytes ÷ time ÷ 8 TB/s "
           "(bench: of the region `value` uses).  traffic = (2 × FETCH_SIZE + WRITE_SIZE) ÷ algorithmic bytes, per launch, kernels of the step only "
           "(calibration: `profiles/r04_traffic_calibration.txt`).  VALU frac = SQ_INSTS_VALU × 2.4 cycles ÷ (1024 SIMDs × rocprof mean × 2.4 GHz) — against "
           "the fastest measured issue rate of a wave64 VALU instruction (`profiles/r03_probe_valu.txt`), ≤ 1 by construction; 4-cycle slots = the same "
           "count against one instruction per 4 cycles per SIMD.  sclk / W = engine clock and package power read from the amdgpu sysfs files while the kernel loops "
           "(`telemetry` of the bench line; the spec clock is 2400 MHz, the package power cap 1400 W).  VALU busy at sclk = SQ_ACTIVE_INST_VALU × 4 clocks ÷ "
           "1024 SIMDs ÷ (sclk × rocprof mean): the share of the launch, at the clock the kernel actually ran at, in which the SIMDs' vector ALUs were executing "
           "(Float64 rows only: the counter ticks once per 4 clocks, exact for Float64 instructions, which occupy the ALU for at least 4; a Float32 instruction "
           "that issues in 2.4–2.9 clocks is charged a whole tick, so the same ratio runs up to 1.7 for the Float32 kernels and is not printed).  `bound` is what `bench.py` prices `roofline.frac` against.", "",
           "| workload | dtype | points | B/point | same ms | rotating ms | sustained ms | first ms | rocprof ms (min) | HBM frac bench | HBM frac rocprof | traffic / algorithmic | "
           "VALU instr / point | VALU frac | 4-cycle slots | sclk MHz / W | VALU busy at sclk | bound |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for _, dt, wl, n, bpp, same, rot, uses, first, ks, fb, fp, tro, ipp, util, bound, clk, sus, busy in rows:
        out.append(f"| `{wl}` | {dt} | {n:.3g} | {bpp} | {same:.3f} | {fmt(rot, '%.3f')}{' ←' if uses == 'rotating' else ''} | {fmt(sus, '%.3f')} | {fmt(first, '%.2f')} | " +
                   (f"{ks[0]:.3f} ({ks[1]:.3f})" if ks else "—") + f" | {fb:.3f} | {fmt(fp, '%.3f')} | {fmt(tro, '%.3f')} | {fmt(ipp, '%.0f')} | "
                   f"{fmt(util * 0.6 if util is not None else None, '%.2f')} | {fmt(util, '%.2f')} | " +
                   ("—" if not clk else f"{clk[0]:.0f}" + (f" / {clk[1]:.0f}" if clk[1] is not None else "")) + f" | {fmt(busy, '%.2f')} | {bound} |")
    # the cold probes of the driver's default line: what separates the clock / power ramp from address translation
    dflt = lines.get("default") or {}
    pr = (dflt.get("cold") or {}).get("probes")
    if pr:
        f5 = lambda v: "[" + ", ".join(f"{x:.2f}" for x in v) + "]"  # noqa: E731
        out += ["", f"Cold probes of the default line (`profiles/bench_{rnd}/default.json`, north star, Float32, kernel ms of five consecutive launches): the first five "
                f"launches of the process {f5(dflt['cold']['first5_ms'])}; the first visit of each other buffer set {f5(dflt['cold']['first_visit_other_sets_ms'])}; "
                f"the SAME buffers after one second of idle {f5(pr['after_1s_idle_same_buffers_ms'])}; FRESH buffers right after a busy period "
                f"{f5(pr['fresh_buffers_warm_clocks_ms'])}; control (steady state) {f5(pr['steady_same_buffers_ms'])}.  Same-buffer region "
                f"{dflt.get('same_buffer_ms_per_step', float('nan')):.4f} ms per step, rotating region {dflt.get('rotating_ms_per_step') or float('nan'):.4f} ms per step; `value` uses the "
                f"{dflt.get('value_uses', '?').replace('_', '-')} region."]
    return "\n".join(out)


def parity_block(rnd):
    d = load(PROF / f"{rnd}_parity_report.json")
    if not d:
        return f"(no `profiles/{rnd}_parity_report.json`)"
    out = [f"From `profiles/{rnd}_parity_report.json` (written by `tests/parity.py` during the `-m gpu` session of round {rnd}; "
           "`tools/gen_design_tables.py` prints this block, `tests/test_design_doc.py` checks it).  One line per kernel family and float type: "
           "output columns compared, points, the share OUTSIDE the plain relative bound (these pass only through the operand-scaled allowance of "
           "`assert_parity`, or through the test's own documented tolerance), points set aside, the smallest per-column fraction inside the plain "
           "bound over the random-state rows, the worst plain relative error among well-conditioned points, and how the rows are asserted "
           "(A = `assert_parity`: operand-scaled bound, fraction inside, well-conditioned plain bound; W = test-specific tolerance AND the "
           "well-conditioned plain bound; T = test-specific tolerance only — the reason is in the row's `note`).", "",
           "| family | dtype | columns | points | outside plain bound | set aside | min fraction inside | worst well-conditioned | asserted |",
           "|---|---|---|---|---|---|---|---|---|"]
    # two tests append their rows themselves (their metrics differ from assert_parity's): attribute them by what they say they are
    for r in d["rows"]:
        if not r.get("family"):
            r["family"] = ("2M + P3 fused entry (f2)" if r["what"].startswith("2M+P3 fused") else
                           "1-moment LinearizedAverage (a2 / f1)" if r["what"].startswith("1M LinearizedAverage") else None)
    fams = []
    for r in d["rows"]:
        f = r.get("family") or "unattributed"
        if f not in fams:
            fams.append(f)
    kind = lambda r: "A" if r.get("asserted", "operand").startswith("operand") else ("W" if "well-conditioned" in r.get("asserted", "") else "T")  # noqa: E731
    for f in sorted(fams):
        for ft in ("f64", "f32"):
            rows = [r for r in d["rows"] if (r.get("family") or "unattributed") == f and r["ft"] == ft]
            if not rows:
                continue
            pts = sum(r["n"] for r in rows)
            outside = sum(r["n_outside"] for r in rows)
            typ = [r for r in rows if "degenerate" not in r["what"]] or rows
            out.append(f"| {f} | {ft} | {len(rows)} | {pts:,} | {outside:,} ({outside / max(pts, 1):.1e}) | {sum(r['n_excluded'] for r in rows):,} | "
                       f"{min(r['frac_within'] for r in typ):.4f} | {max(r['worst_wellcond'] for r in rows):.2e} | {'/'.join(sorted({kind(r) for r in rows}))} |")
    out.append("")
    for ft, tol in (("f64", "1e-6"), ("f32", "1e-3")):
        s = d["summary"].get(ft)
        if not s:
            continue
        rows = [r for r in d["rows"] if r["ft"] == ft]
        bound = [r for r in rows if kind(r) in "AW"]
        worst = max(bound, key=lambda r: r["worst_wellcond"])
        # the "degenerate" sets place their states ON the cancellation points (q_v = q_sat, T = T_freeze ± 0.01 K, contents at ϵ): their plain
        # fraction is reported separately — the statement about typical states comes from the random-state rows
        adv = [r for r in rows if "degenerate" in r["what"] and kind(r) == "A"]
        typ = [r for r in rows if "degenerate" not in r["what"] and kind(r) == "A"] or rows
        low = min(typ, key=lambda r: r["frac_within"])
        out.append(f"* **{'Float64' if ft == 'f64' else 'Float32'}** (plain bound {tol}): {s['rows']} output columns, {s['points']:,} compared points, "
                   f"{s['outside_plain_bound']:,} outside the plain relative bound ({s['outside_plain_bound'] / max(s['points'], 1):.2e} of them), "
                   f"{s['excluded_near_branch']:,} set aside (next to a genuine discontinuity of the scheme, on another root of a solver, or — LinearizedAverage "
                   f"Float32 rows — below the difference-quotient floor); smallest per-column fraction inside among the `assert_parity` rows "
                   f"{low['frac_within']:.4f} (`{low['what']}` `{low['output']}`)" +
                   (f"; in the adversarial degenerate-state sets {min(r['frac_within'] for r in adv):.4f}" if adv else "") +
                   f"; worst well-conditioned point of the rows asserted at the plain bound {worst['worst_wellcond']:.2e} (`{worst['what']}` `{worst['output']}`).")
    bad = [r for r in d["rows"] if kind(r) in "AW" and r["worst_wellcond"] > r["rtol"]]
    out.append("")
    out.append(f"Rows whose worst well-conditioned point exceeds the tolerance: {len(bad)}" +
               ("." if not bad else ": " + "; ".join(f"`{r['what']}` `{r['output']}` {r['worst_wellcond']:.2e}" for r in bad) + "."))
    tonly = [r for r in d["rows"] if kind(r) == "T" and r["worst_wellcond"] > r["rtol"]]
    if tonly:
        groups = {}
        for r in tonly:      # one entry per (test, reason): the counts inside a reason differ between the variants of one test
            key = (re.sub(r"\s+", " ", re.sub(r"\b(from_state=\w+|f32|f64)\b", "", r["what"])).strip(), re.sub(r"^\d+ of \d+ ", "some ", r.get("note", "")))
            groups[key] = max(groups.get(key, 0.0), r["worst_wellcond"])
        out.append("Rows asserted at a test-specific tolerance only (T) whose worst point exceeds the plain bound, with the reason the test states: " +
                   "; ".join(f"`{k[0]}` {w:.1e} — {k[1] or 'see the test'}" for k, w in groups.items()) + ".")
    set_aside = [r for r in d["rows"] if r.get("frac_below_difference_quotient_floor", 0) > 0]
    if set_aside:
        w = max(set_aside, key=lambda r: r["frac_below_difference_quotient_floor"])
        out.append(f"LinearizedAverage Float32 rows set points aside below the difference-quotient floor eps·q/Δt (tests/test_mp1m_linearized.py): "
                   f"at most {w['frac_below_difference_quotient_floor']:.3f} of the states (`{w['what']}` `{w['output']}`).")
    return "\n".join(out)


def floor_block(rnd):
    """§4.2: the Float64 instruction floor (tools/f64_floor.py --markdown: census on the counting type, dynamic function costs, measured instructions)."""
    r = subprocess.run([sys.executable, str(REPO / "tools" / "f64_floor.py"), "--round", rnd, "--markdown"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return r.stdout.strip()


def wrap_prose(text, width=150):
    """Prose lines of a generated block wrapped at `width`; table rows (|…|) and list continuations stay whole lines / get a two-space hanging indent."""
    import textwrap
    out = []
    for line in text.split("\n"):
        if line.startswith("|") or len(line) <= width:
            out.append(line)
        elif line.startswith("* "):
            out += textwrap.wrap(line, width, subsequent_indent="  ", break_long_words=False, break_on_hyphens=False)
        else:
            out += textwrap.wrap(line, width, break_long_words=False, break_on_hyphens=False)
    return "\n".join(out)


BLOCKS = {"performance": performance_block, "parity": parity_block, "floor": floor_block}


def main():
    args = sys.argv[1:]
    rnd = args[args.index("--round") + 1] if "--round" in args else latest_round()
    doc_path = REPO / "DESIGN.md"
    gen = {name: wrap_prose(fn(rnd)) for name, fn in BLOCKS.items()}
    if "--write" not in args and "--check" not in args:
        for name, text in gen.items():
            print(f"<!-- BEGIN GENERATED {name} -->\n{text}\n<!-- END GENERATED {name} -->\n")
        return 0
    doc = doc_path.read_text()
    stale = []
    for name, text in gen.items():
        pat = re.compile(rf"(<!-- BEGIN GENERATED {name} -->\n)(.*?)(\n<!-- END GENERATED {name} -->)", re.S)
        m = pat.search(doc)
        if not m:
            stale.append(f"{name}: markers missing")
            continue
        if m.group(2) != text:
            stale.append(name)
            doc = doc[:m.start(2)] + text + doc[m.end(2):]
    if "--write" in args:
        doc_path.write_text(doc)
        print("rewrote: " + (", ".join(stale) if stale else "nothing (current)"))
        return 0
    if stale:
        print("DESIGN.md generated blocks are stale: " + ", ".join(stale) + f"  (run tools/gen_design_tables.py --write --round {rnd})")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
