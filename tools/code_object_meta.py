#!/usr/bin/env python3
"""Kernel metadata of the gfx950 code objects embedded in a host shared library (hipcc fat binary): name, VGPRs,
SGPRs, LDS and the private-segment (scratch) size of every kernel.  Pure Python: walks the clang offload bundles
(``__CLANG_OFFLOAD_BUNDLE__``), the ELF note of each device code object (NT_AMDGPU_METADATA = 32) and its msgpack.
usage: tools/code_object_meta.py [path/to/lib.so] [name-substring]"""
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_notes(elf: bytes):
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        off = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, off + 4)
        sh_offset, sh_size = struct.unpack_from("<QQ", elf, off + 0x18)
        if sh_type != 7:                                   # SHT_NOTE
            continue
        p, end = sh_offset, sh_offset + sh_size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name.rstrip(b"\0"), ntype, desc


def kernels(path: str):
    """[{name, vgpr, sgpr, lds, scratch, arch}] for every device kernel in ``path``."""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tsz].decode(errors="replace")
            q += 24 + tsz
            if "amdgcn" not in triple or size == 0:
                continue
            elf = data[pos + off: pos + off + size]
            for name, ntype, desc in _elf_notes(elf):
                if name == b"AMDGPU" and ntype == 32:
                    meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    for k in meta.get("amdhsa.kernels", []):
                        out.append({"name": k[".name"], "vgpr": k.get(".vgpr_count"), "agpr": k.get(".agpr_count"),
                                    "sgpr": k.get(".sgpr_count"), "lds": k.get(".group_segment_fixed_size"),
                                    "scratch": k.get(".private_segment_fixed_size"), "arch": triple})
        pos += len(MAGIC)
    return out


if __name__ == "__main__":
    import os
    import subprocess
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "graphnet_amd", "libgraphnet_amd.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(lib):
        nm = subprocess.run(["c++filt", k["name"]], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
        if flt in nm:
            print(f"{nm[:90]:90s} vgpr {k['vgpr']:>4} agpr {k['agpr'] or 0:>4} sgpr {k['sgpr']:>4} lds {k['lds']:>7} scratch {k['scratch']:>4}")
