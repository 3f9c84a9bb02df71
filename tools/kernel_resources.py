#!/usr/bin/env python3
"""Registers, LDS and scratch of every gfx950 kernel in a built libcmx.so, from the code-object metadata (no GPU needed).

    tools/kernel_resources.py [lib.so] [--scratch]      all kernels, or only those with a private segment / spilled VGPRs
    tools/kernel_resources.py [lib.so] --loops          scratch instructions per kernel, and how many of them sit inside a loop (disassembly)

The library holds one clang offload bundle per translation unit (magic __CLANG_OFFLOAD_BUNDLE__); each bundle's gfx950 entry is an ELF
code object whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count / .sgpr_count / .vgpr_spill_count / .sgpr_spill_count /
.private_segment_fixed_size / .group_segment_fixed_size.  tests/test_kernel_resources.py asserts that the collision kernels keep spill traffic
out of their loops (`--loops` below) and that the streaming kernels spill nothing."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    data = Path(path).read_bytes()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            o, size, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24:off + 24 + tlen].decode()
            off += 24 + tlen
            if "gfx950" in triple and size:
                yield data[pos + o:pos + o + size]
        pos += len(MAGIC)


def kernels(path):
    out = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            d = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
            if "name" in d:
                out.append({"name": d["name"], "vgpr": int(d.get("vgpr_count", 0)), "sgpr": int(d.get("sgpr_count", 0)),
                            "vgpr_spill": int(d.get("vgpr_spill_count", 0)), "sgpr_spill": int(d.get("sgpr_spill_count", 0)),
                            "private": int(d.get("private_segment_fixed_size", 0)), "lds": int(d.get("group_segment_fixed_size", 0)),
                            "kernarg": int(d.get("kernarg_segment_size", 0))})
    return out


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def scratch_in_loops(path, substring=""):
    """{kernel name: (scratch instructions, those that lie inside a loop, those that lie inside an INNERMOST loop, loops found)} for the kernels whose mangled name contains `substring`,
    from the disassembly of the built code objects.  A loop is the address range between a backward branch and its target; a spilled value that is
    stored / reloaded once between two phases of a kernel is outside every loop and costs nothing measurable, spill traffic inside a loop does."""
    out = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            dis = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
        name, start, scr, loops = None, 0, [], []

        def close():
            if name is not None and substring in name:
                inner = [(lo, hi) for lo, hi in loops if not any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in loops)]
                out[name] = (len(scr), sum(1 for a in scr if any(lo <= a <= hi for lo, hi in loops)),
                             sum(1 for a in scr if any(lo <= a <= hi for lo, hi in inner)), len(loops))

        for line in dis.splitlines():
            m = re.match(r"^([0-9a-f]{8,16}) <(\S+)>:$", line)
            if m:
                close()
                start, name, scr, loops = int(m.group(1), 16), m.group(2), [], []
                continue
            m = re.search(r"//\s*([0-9A-Fa-f]{8,16}):", line)
            if not m or name is None:
                continue
            addr = int(m.group(1), 16) - start
            op = line.split()[0]
            if op.startswith("scratch_"):
                scr.append(addr)
            elif op.startswith(("s_cbranch", "s_branch")):
                t = re.search(r"<[^>]*\+0x([0-9a-f]+)>", line)
                tgt = int(t.group(1), 16) if t else (0 if re.search(r"<[^+>]+>\s*$", line) else None)
                if tgt is not None and tgt <= addr:
                    loops.append((tgt, addr))
        close()
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return r.stdout.splitlines() if r.returncode == 0 and r.stdout else names
    except OSError:
        return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else str(Path(__file__).resolve().parent.parent / "cloudmicrophysics.jl_amd" / "csrc" / "libcmx.so")
    if "--loops" in sys.argv:      # scratch instructions of every kernel that has some: total and inside loops
        d = scratch_in_loops(lib)
        for (name, (n, inside, innermost, _)), dem in zip(d.items(), demangle(list(d))):
            if n:
                print(f"scratch instructions {n:3d}, inside loops {inside:3d}, inside innermost loops {innermost:3d} | {dem[:170]}")
        return
    ks = kernels(lib)
    if "--scratch" in sys.argv:
        ks = [k for k in ks if k["private"] or k["vgpr_spill"]]
    # `scratch instr`: scratch_load / scratch_store instructions in the code.  A private segment WITHOUT any is the frame the compiler reserved for SGPR spills
    # that then went to VGPR lanes (v_writelane): nothing touches memory.
    scr = scratch_in_loops(lib)
    for k, name in zip(ks, demangle([k["name"] for k in ks])):
        print(f"vgpr {k['vgpr']:3d} spill {k['vgpr_spill']:3d} | sgpr {k['sgpr']:3d} spill {k['sgpr_spill']:3d} | private {k['private']:4d} B | scratch instr {scr.get(k['name'], (0,))[0]:2d} | "
              f"lds {k['lds']:6d} B | kernarg {k['kernarg']:4d} B | {name[:150]}")
    print(f"{len(ks)} kernels")


if __name__ == "__main__":
    main()
