#!/usr/bin/env python3
"""Registers, LDS and scratch of every gfx950 kernel in a built libcmx.so, from the code-object metadata (no GPU needed).

    tools/kernel_resources.py [lib.so] [--scratch]      all kernels, or only those with a private segment / spilled VGPRs

The library holds one clang offload bundle per translation unit (magic __CLANG_OFFLOAD_BUNDLE__); each bundle's gfx950 entry is an ELF
code object whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count / .sgpr_count / .vgpr_spill_count / .sgpr_spill_count /
.private_segment_fixed_size / .group_segment_fixed_size.  tests/test_build.py asserts that the two collision kernels stay at zero spilled
VGPRs."""
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


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return r.stdout.splitlines() if r.returncode == 0 and r.stdout else names
    except OSError:
        return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else str(Path(__file__).resolve().parent.parent / "cloudmicrophysics.jl_amd" / "csrc" / "libcmx.so")
    ks = kernels(lib)
    if "--scratch" in sys.argv:
        ks = [k for k in ks if k["private"] or k["vgpr_spill"]]
    for k, name in zip(ks, demangle([k["name"] for k in ks])):
        print(f"vgpr {k['vgpr']:3d} spill {k['vgpr_spill']:3d} | sgpr {k['sgpr']:3d} spill {k['sgpr_spill']:3d} | private {k['private']:4d} B | lds {k['lds']:6d} B | kernarg {k['kernarg']:4d} B | {name[:150]}")
    print(f"{len(ks)} kernels")


if __name__ == "__main__":
    main()
