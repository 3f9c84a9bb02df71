#!/usr/bin/env python3
"""VALU instructions of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only) by BRANCH NESTING depth: depth 1 is the kernel's main path (inside
the bounds check), deeper levels are regions behind a forward conditional branch (rescue blocks of the table-driven logarithm, compiler-sunk selects …).
Comparing the main-path count with the measured SQ_INSTS_VALU per point tells whether "rare" blocks run for ordinary data (HISTORY.md R6: they did in the
Float64 1-moment kernel).

    tools/isa_regions.py file.s <substring of the mangled kernel name>
"""
import re,sys,collections
path,kern=sys.argv[1],sys.argv[2]
inside=False; lines=[]
for line in open(path):
    if re.match(r'^_Z\S*:',line):
        if inside: break
        inside=kern in line; continue
    if inside:
        lines.append(line.rstrip())
        if 's_endpgm' in line: break
# conditional regions: from a forward branch to its label
open_until=[]  # labels
depthcount=collections.Counter(); regions=[]
cur=None
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\S+):',l)
    if m:
        lab=m.group(1)
        while lab in open_until:
            open_until.remove(lab)
        continue
    m=re.match(r'\s+([a-z_0-9]+)\s*(.*)',l)
    if not m: continue
    op,args=m.group(1),m.group(2)
    if op.startswith('v_'):
        depthcount[len(open_until)]+=1
    if op.startswith('s_cbranch'):
        lab=args.split()[-1]
        # forward only
        fwd=any(re.match(r'^'+re.escape(lab)+':',x) for x in lines[i+1:])
        if fwd and lab not in open_until: open_until.append(lab)
print(kern, dict(depthcount), 'total', sum(depthcount.values()))
