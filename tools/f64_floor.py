#!/usr/bin/env python3
"""The Float64 instruction floor of the three streaming kernels, from three committed measurements:

  census     what the SOURCE asks for per point — tools/census/f64_census.cpp instantiates the device point functions (csrc/cmx_sb2006.hpp, cmx_mp1m.hpp,
             cmx_arg.hpp) on a counting value type that takes the Float64 code paths: calls of each elementary function, fused multiply-adds, multiplies,
             adds, min / max, compares (the gates are selects, so the tally does not depend on the state);
  cost       DYNAMIC VALU instructions of each elementary function of csrc/cmx_lean_f64.hpp — per-dispatch SQ_INSTS_VALU / SQ_WAVES of
             cmx_lean_eval_literal_f64 minus the identity launch (tools/lean_cost_run.py under rocprofv3 --pmc; profiles/<round>_lean_cost_counters.csv);
  measured   SQ_INSTS_VALU per point of the production kernel (profiles/<round>_pmc_valu_<workload>_f64.json, tools/profile.sh).

floor = Σ calls × cost + fma + mul + add + min/max + compares.  Not in the floor (the kernel's overhead above it): the selects behind the gates (two
v_cndmask_b32 per Float64 select), the NaN poison, address arithmetic, literal / SGPR moves, v_readlane of spilled kernel constants.

    python tools/f64_floor.py [--round r06] [--markdown]"""
import csv
import json
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tools"))
WORKLOAD = {"sb2006_f64": "sb2006", "mp1m_f64": "mp1m", "arg2000_f64": "arg2000"}
ARITH = ("fma", "mul", "add", "minmax", "cmp")
# census name → cost name (Math<double> maps div to rcp + mul; rcp_nz1 is lean::rcp_finite)
COST_OF = {"exp2": "exp2", "exp2_fin": "exp2_fin", "log2": "log2", "rcp": "rcp", "rcp_nz": "rcp_nz", "rcp_nz1": "rcp_nz1", "sqrt": "sqrt", "sqrt_pos": "sqrt_pos",
           "rsqrt": "rsqrt", "rsqrt_pos": "rsqrt_pos", "pow_m34_pos": "pow_m34_pos", "erfc": "erfc", "expm1": "expm1", "log1p": "log1p"}


def census():
    src = REPO / "tools" / "census" / "f64_census.cpp"
    with tempfile.TemporaryDirectory() as d:
        exe = Path(d) / "f64_census"
        subprocess.run(["g++", "-std=c++2a", "-O1", "-DCMX_HOST_BUILD=1", f"-I{REPO / 'include'}", "-o", str(exe), str(src)], check=True, cwd=src.parent)
        out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    res = {}
    for line in out.splitlines():
        name, *kv = line.split()
        res[name] = {k: int(v) for k, v in (x.split("=") for x in kv)}
    return res


def costs(rnd):
    import lean_cost_run as L
    path = REPO / "profiles" / f"{rnd}_lean_cost_counters.csv"
    if not path.exists():      # the functions' costs change only when cmx_lean_f64.hpp does: the latest committed measurement stands
        path = sorted((REPO / "profiles").glob("r*_lean_cost_counters.csv"))[-1]
    rows = list(csv.DictReader(open(path)))
    d = {}
    for r in rows:
        d.setdefault(int(r["dispatch"]), {})[r["counter"]] = float(r["value"])
    ids = sorted(d)
    assert len(ids) == len(L.FUNCS), (len(ids), len(L.FUNCS))
    per = {name: d[i]["SQ_INSTS_VALU"] / d[i]["SQ_WAVES"] for (_, name, _), i in zip(L.FUNCS, ids)}
    return {k: v - per["identity"] for k, v in per.items() if k != "identity"}


def measured(rnd, wl):
    p = REPO / "profiles" / f"{rnd}_pmc_valu_{wl}_f64.json"
    if not p.exists():
        return None
    ks = json.loads(p.read_text())["kernels"]
    return max(k["valu_instructions_per_point"] for k in ks.values())


def main():
    args = sys.argv[1:]
    rnd = args[args.index("--round") + 1] if "--round" in args else "r06"
    cen, cost = census(), costs(rnd)
    rows = []
    for name, tally in cen.items():
        trans = {k: v for k, v in tally.items() if k not in ARITH}
        unknown = [k for k in trans if k not in COST_OF]
        assert not unknown, unknown
        t_instr = sum(v * cost[COST_OF[k]] for k, v in trans.items())
        arith = sum(tally.get(k, 0) for k in ARITH)
        m = measured(rnd, WORKLOAD[name])
        rows.append((name, trans, t_instr, arith, t_instr + arith, m))
    if "--markdown" in args:
        print("| kernel (Float64) | elementary-function calls per point | their instructions | fma + mul + add + min/max + compare | floor | measured (SQ_INSTS_VALU / point) | above the floor |")
        print("|---|---|---|---|---|---|---|")
        for name, trans, t, a, f, m in rows:
            calls = ", ".join(f"{v} × {k} ({cost[COST_OF[k]]:.0f})" for k, v in sorted(trans.items(), key=lambda kv: -kv[1] * cost[COST_OF[kv[0]]]))
            print(f"| `{WORKLOAD[name]}` | {calls} | {t:.0f} | {a} | {f:.0f} | {m if m is not None else 'n/a'} | {('%.0f %%' % (100 * (m / f - 1))) if m else 'n/a'} |")
    else:
        print("dynamic cost of the elementary functions (VALU instructions):", {k: round(v, 1) for k, v in cost.items()})
        for name, trans, t, a, f, m in rows:
            print(f"{name}: calls {trans} -> {t:.0f} instr; arithmetic {a}; floor {f:.0f}; measured {m}; above the floor {('%.1f %%' % (100 * (m / f - 1))) if m else 'n/a'}")


if __name__ == "__main__":
    main()
