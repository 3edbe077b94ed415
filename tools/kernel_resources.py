#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in librbnn_hip.so, read from the code objects inside the library (no GPU needed).

The library carries one clang offload bundle per translation unit (magic __CLANG_OFFLOAD_BUNDLE__: entry table of offset / size / target
triple); the gfx950 entry of each is an ELF code object whose NT_AMDGPU_METADATA note `llvm-readelf --notes` prints as YAML.
usage: python tools/kernel_resources.py [library] [name filter]      (tests/test_host_cpu.py asserts on kernel_resources())"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "robustbnns_amd", "csrc", "librbnn_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib=LIB, arch="gfx950"):
    """-> list of bytes: the `arch` code object of every bundle in the library."""
    blob = open(lib, "rb").read()
    out = []
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tl].decode()
            pos += 24 + tl
            if arch in triple and size:
                out.append(blob[base + off:base + off + size])
    return out


def demangle(names):
    try:
        r = subprocess.run(["/usr/bin/c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")[:len(names)]
    except Exception:
        return names


def kernel_resources(lib=LIB):
    """-> {demangled kernel name: {"vgpr", "agpr", "sgpr", "lds", "scratch", "spill_vgpr"}} for every gfx950 kernel of the library."""
    res = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        # one list item per kernel ("  - .agpr_count: ..." first: the keys are sorted), kernel-level keys at four spaces of indentation
        entries = []
        for block in re.split(r"(?m)^  - (?=\.agpr_count:)", txt)[1:]:
            cur = {}
            for line in block.split("\n"):
                m = re.match(r"(?:    )?\.(\w+):\s*(\S+)\s*$", line)
                if not m or (line.startswith("     ")):
                    continue
                k, v = m.group(1), m.group(2).strip("'")
                key = {"vgpr_count": "vgpr", "agpr_count": "agpr", "sgpr_count": "sgpr", "group_segment_fixed_size": "lds",
                       "private_segment_fixed_size": "scratch", "vgpr_spill_count": "spill_vgpr", "name": "name"}.get(k)
                if key:
                    cur[key] = v if key == "name" else int(v)
            if "name" in cur:
                entries.append(cur)
        names = demangle([e.get("name", "?") for e in entries])
        for e, nm in zip(entries, names):
            if "vgpr" in e:
                res[nm] = {k: e.get(k, 0) for k in ("vgpr", "agpr", "sgpr", "lds", "scratch", "spill_vgpr")}
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else LIB
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    res = kernel_resources(lib)
    print(f"{'kernel':110s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'LDS (static)':>13s} {'scratch':>8s}")
    for nm in sorted(res):
        if flt in nm:
            r = res[nm]
            print(f"{nm[:110]:110s} {r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['lds']:13d} {r['scratch']:8d}")
