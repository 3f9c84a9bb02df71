#!/usr/bin/env python3
"""Issue-time estimate of a gfx950 kernel from its assembly listing and the measured per-instruction issue times of
tools/valu_probe.hip (round 3, units: cycles of a 2.4 GHz clock per wave64 instruction per SIMD):

    full-rate (v_mul/v_add/v_sub/v_mov v,v / v_fma with <= 2 distinct VGPR sources / literal forms) ~2.6
    v_fmac_f32 3.7;  v_fma_f32 with 3 distinct VGPRs 4.2;  any SGPR source operand 4.4;  v_max/v_min/v_med3/v_cmp/v_cndmask/v_ldexp 4.2-4.6
    transcendental (v_exp/v_log/v_rcp/v_rsq/v_sqrt f32) 8.3;  f64: add/mul/fma/max 4.3-5.2, rcp/rsq/sqrt 16.3
    packed f32 (two results; round 5 probe): v_pk_mul/add/fma with an SGPR operand 4.7, v_pk_add 4.4, v_pk_mul v,v 5.1, v_pk_fma of three registers 5.5

    tools/isa_cost.py file.s <kernel-name substring>... [--per K]
"""
import re
import sys
import collections


def cost(op, args):
    srcs = args[1:] if len(args) > 1 else []
    has_s = any(re.match(r"^-?\|?s(\d+|\[)", a) or a in ("vcc", "vcc_lo", "vcc_hi", "exec") for a in srcs)
    vs = {a.strip("-|") for a in srcs if re.match(r"^-?\|?v(\d+|\[)", a)}
    f64 = "_f64" in op or op.endswith("_b64") or "_i64" in op or "_u64" in op
    if op.startswith(("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")):
        return ("trans64", 16.3) if f64 else ("trans", 8.3)
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return ("lane", 4.2)
    if f64:
        if has_s:
            return ("f64+sgpr", 5.3)
        return ("f64", 4.7)
    if op.startswith("v_pk_"):      # round 5 probe (profiles/r05_probe_valu.txt): two results per instruction
        if has_s:
            return ("pk+sgpr", 4.7)
        if op.startswith("v_pk_fma") and len(vs) >= 3:
            return ("pk_fma3", 5.5)
        if op.startswith("v_pk_mul"):
            return ("pk_mul", 5.1)
        return ("pk", 4.4)
    if op.startswith("v_cmp"):
        return ("cmp", 4.5)
    if op.startswith("v_cndmask"):
        return ("cndmask", 4.6)
    if op.startswith(("v_max", "v_min", "v_med3", "v_ldexp")):
        return ("minmax", 4.3)
    if has_s:
        return ("sgpr-src", 4.4)
    if op.startswith("v_fmac"):
        return ("fmac", 3.7)
    if op.startswith("v_fma_") and len(vs) >= 3:
        return ("fma3", 4.2)
    return ("full", 2.7)


def main():
    argv = sys.argv[1:]
    per = 1
    if "--per" in argv:
        i = argv.index("--per"); per = int(argv[i + 1]); del argv[i:i + 2]
    path, subs = argv[0], argv[1:]
    cur = None
    tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for ln in open(path):
        m = re.match(r"^(_Z[\w.$]+):", ln)
        if m:
            cur = m.group(1); continue
        if cur is None:
            continue
        s = ln.split(";")[0].strip()
        if s.startswith(".Lfunc_end"):
            cur = None; continue
        if not s.startswith("v_"):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        k, c = cost(op, args)
        tot[cur][k][0] += 1; tot[cur][k][1] += c
    for name, d in tot.items():
        if subs and not all(x in name for x in subs):
            continue
        n = sum(v[0] for v in d.values()); c = sum(v[1] for v in d.values())
        print(name[:140])
        print(f"    VALU {n} ({n / per:.1f}/pt)  est. issue {c:.0f} cycles@2.4GHz ({c / per:.0f}/pt)   " +
              "  ".join(f"{k} {v[0]}→{v[1]:.0f}" for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])))


if __name__ == "__main__":
    main()
