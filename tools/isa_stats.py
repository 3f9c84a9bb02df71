#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    tools/isa_stats.py file.s [substring-of-kernel-name ...] [--top N] [--per K]

Counts VALU / transcendental / SALU / SMEM / VMEM / LDS / branch instructions of each kernel body (label to s_endpgm) and prints the
resource lines of its descriptor.  `--per K` divides by K (points a lane owns).  Static counts: a loop body counts once.
"""
import collections
import re
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "trans"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "lane"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "vmov"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith("v_cmp"):
            return "vcmp"
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep")):
        return "swait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    per = 1
    top = 0
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--per":
            per = int(argv[i + 1]); args.remove(argv[i + 1])
        if a == "--top":
            top = int(argv[i + 1]); args.remove(argv[i + 1])
    path, subs = args[0], args[1:]
    text = open(path).read().splitlines()
    kernels = {}
    cur = None
    for ln in text:
        m = re.match(r"^(_Z[\w.$]+):", ln)
        if m:
            cur = m.group(1); kernels[cur] = []; continue
        if cur is None:
            continue
        s = ln.strip()
        if s.startswith("s_endpgm"):
            kernels[cur].append("s_endpgm"); cur = None; continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        kernels[cur].append(s.split()[0])
    meta = collections.defaultdict(dict)
    name = None
    for ln in text:
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
        m = re.match(r"\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", ln)
        if m and name:
            meta[name][m.group(1)] = int(m.group(2))
    for k, ops in kernels.items():
        if subs and not all(s in k for s in subs):
            continue
        if not ops:
            continue
        c = collections.Counter(classify(o) for o in ops)
        vtot = sum(c[x] for x in ("valu", "trans", "lane", "vmov", "cndmask", "vcmp"))
        print(k[:150])
        print("   ", {kk: meta.get(k, {}).get(kk) for kk in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")})
        print(f"    total {len(ops)}  VALU-all {vtot} ({vtot / per:.1f}/pt)  " + "  ".join(f"{x} {c[x]}" for x in
              ("valu", "trans", "vmov", "cndmask", "vcmp", "lane", "salu", "smem", "vmem", "lds", "scratch", "branch", "swait", "other")))
        if top:
            cc = collections.Counter(ops)
            print("    " + "  ".join(f"{o} {n}" for o, n in cc.most_common(top)))


if __name__ == "__main__":
    main()
