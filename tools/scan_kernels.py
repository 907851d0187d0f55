#!/usr/bin/env python3
"""Kernel resource table of the built libisls_hip.so (no GPU, no recompilation): registers, scratch (private segment), LDS
and workgroup size of every gfx950 kernel, read from the code objects inside the library's HIP fat binary.

    python tools/scan_kernels.py [--lib path] [--filter substring] [--scratch-only]

Flags the two things that silently halve a kernel: scratch memory in use (register spills) and a register count just
past an occupancy step (512 / 256 / 168 / 128 registers per lane = 1 / 2 / 3 / 4 wavefronts per SIMD).
`kernel_table(path)` is what tests/test_capi_host.py uses to keep the hot-path kernels free of scratch."""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "ilqr-admm_amd", "csrc", "libisls_hip.so")
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(fat):
    """gfx950 ELF images of every offload bundle in a .hip_fatbin section."""
    i = 0
    while True:
        j = fat.find(MAGIC, i)
        if j < 0:
            return
        n, = struct.unpack_from("<Q", fat, j + len(MAGIC))
        p = j + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield fat[j + off:j + off + size]
        i = j + 1


def _metadata(elf):
    """AMDGPU metadata (msgpack) from the NT_AMDGPU_METADATA note of a 64-bit little-endian ELF."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        stype, = struct.unpack_from("<I", elf, sh + 4)
        if stype != 7:                                         # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name.startswith(b"AMDGPU") and ntype == 32:     # NT_AMDGPU_METADATA
                return msgpack.unpackb(desc, raw=False, strict_map_key=False)
    return None


def kernel_table(path=DEFAULT_LIB):
    """{demangled-ish kernel symbol: dict(vgpr, agpr, sgpr, scratch, lds, wg)} for every gfx950 kernel of the library."""
    with tempfile.TemporaryDirectory() as td:
        fb, dummy = os.path.join(td, "fat.bin"), os.path.join(td, "copy.so")
        subprocess.run([OBJCOPY, f"--dump-section=.hip_fatbin={fb}", path, dummy], check=True, capture_output=True)
        fat = open(fb, "rb").read()
    out = {}
    for elf in _code_objects(fat):
        md = _metadata(elf)
        for k in (md or {}).get("amdhsa.kernels", []):
            out[k[".name"]] = dict(vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count", 0),
                                   scratch=k.get(".private_segment_fixed_size", 0), lds=k.get(".group_segment_fixed_size", 0),
                                   wg=k.get(".max_flat_workgroup_size", 0), spill=k.get(".vgpr_spill_count", 0))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")[:len(names)]
    except Exception:
        return list(names)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--filter", default="")
    ap.add_argument("--scratch-only", action="store_true")
    a = ap.parse_args()
    tab = kernel_table(a.lib)
    names = sorted(tab)
    nice = dict(zip(names, demangle(names)))
    steps = (128, 168, 256)
    shown = 0
    for n in names:
        k = tab[n]
        d = nice[n].replace("void isls::", "").split("(")[0]
        if a.filter and a.filter not in d:
            continue
        if a.scratch_only and not k["scratch"]:
            continue
        regs = k["vgpr"]
        near = [s for s in steps if 0 < regs - s <= 12]
        flag = ("  SCRATCH" if k["scratch"] else "") + (f"  just past {near[0]} registers" if near else "")
        print(f"{d[:86]:86s} vgpr {regs:4d} sgpr {k['sgpr']:4d} scratch {k['scratch']:5d} lds {k['lds']:6d} wg {k['wg']:5d}{flag}")
        shown += 1
    print(f"{shown} of {len(names)} kernels; {sum(1 for n in names if tab[n]['scratch'])} use scratch memory", file=sys.stderr)


if __name__ == "__main__":
    main()
