#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: for every scratch_load / scratch_store of one kernel in a `hipcc -S` listing, the loop nest
(back-edge depth) it executes in.  Loops are recognised from backward branches to .LBB labels.

    tools/spill_map.py file.s <substring of the mangled kernel name>
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"[A-Za-z_]\S*:", l) and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    label_at = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            loops.append((label_at[m.group(1)], i))
    depth = lambda i: sum(1 for a, b in loops if a <= i <= b)  # noqa: E731
    hist = {}
    n_inst = {}
    for i, l in enumerate(body):
        t = l.strip().split()
        if not t or t[0].endswith(":") or t[0].startswith((".", ";")):
            continue
        d = depth(i)
        n_inst[d] = n_inst.get(d, 0) + 1
        if t[0].startswith(("scratch_", "v_readlane", "v_writelane")):
            hist.setdefault(d, {}).setdefault(t[0], 0)
            hist[d][t[0]] += 1
    print(f"{len(loops)} loops; instructions by loop depth: {dict(sorted(n_inst.items()))}")
    for d in sorted(hist):
        print(f"  depth {d}: {hist[d]}")
    # innermost loops holding scratch ops
    for a, b in sorted(loops, key=lambda ab: ab[1] - ab[0]):
        ops = [body[i].strip().split()[0] for i in range(a, b + 1) if body[i].strip().startswith(("scratch_", "v_readlane", "v_writelane"))]
        if ops:
            inner = [x for x in loops if a <= x[0] and x[1] <= b and x != (a, b)]
            print(f"  loop lines {a}-{b} ({b - a} lines, depth {depth(a)}, {len(inner)} nested loops): {len(ops)} scratch / SGPR-lane ops")


if __name__ == "__main__":
    main()
