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
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
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


def _regs(op):
    """'v12' / 'v[4:7]' / 'a[0:3]' (with an optional leading '-' or '|') -> ('v' | 'a', set of register numbers); anything else -> (None, empty)."""
    m = re.match(r"^[-|]?([va])(?:(\d+)|\[(\d+):(\d+)\])\|?$", op.strip())
    if not m:
        return None, set()
    lo = int(m.group(2) if m.group(2) is not None else m.group(3))
    hi = int(m.group(2) if m.group(2) is not None else m.group(4))
    return m.group(1), set(range(lo, hi + 1))


def mfma_operand_hazards(lib=LIB, need=2):
    """-> [(kernel, mfma line, writer line, wait states seen)]: every v_mfma that takes as an operand a VGPR some non-MFMA vector instruction wrote
    fewer than `need` wait states earlier (gfx90a+ rule 'VALU writes VGPR -> MFMA reads it: 2 wait states', LLVM's LegacyVALUWritesVGPRWaitStates).
    hipcc pads this hazard itself between instructions it can see; it cannot see INTO an inline-asm string, so a block that ends in a vector
    write (the v_fma_mixhi_f16 of the pair splits) must carry its own `s_nop 1` — this scan over the disassembly of the built library is the fence
    (tests/test_host_cpu.py).  Linear scan per kernel: an instruction is one wait state, `s_nop N` is N + 1."""
    bad = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kern, recent = "?", []                                   # recent: [(states since issue, vgprs written, text)], youngest first
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kern, recent = m.group(1), []
                continue
            ins = line.split("//")[0].strip()
            if not ins or ins.startswith("."):
                continue
            mn, _, rest = ins.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
                srcs = set()
                for o in ops[1:4]:
                    k, r = _regs(o.split(" ")[0])
                    if k == "v":
                        srcs |= r
                for age, regs, text in recent:
                    if age < need and regs & srcs:
                        bad.append((kern, ins, text, age))
            states = (int(rest.strip().split()[0], 0) + 1) if mn == "s_nop" else 1
            recent = [(a + states, r, t) for a, r, t in recent if a + states < need]
            if mn.startswith("v_") and not (mn.startswith("v_mfma") or mn.startswith("v_smfma") or mn.startswith("v_cmp") or mn.startswith("v_accvgpr_write")):
                k, r = _regs(ops[0].split(" ")[0])
                if k == "v":
                    recent.insert(0, (0, r, ins))
    names = demangle([b[0] for b in bad])
    return [(n,) + b[1:] for n, b in zip(names, bad)]


def _mfma_result_wait(mn):
    """instructions' worth of wait states (an instruction = 1, `s_nop N` = N + 1) that must lie BETWEEN an MFMA and a vector instruction that reads
    its result.  Calibrated on hipcc 7.2's own padding for gfx950 in this library (the closest pairs it emits: 8 behind v_mfma_f32_16x16x32_f16,
    10 behind v_mfma_f32_16x16x4_f32); anything else is held to the 16-pass figure."""
    if "16x16x32" in mn:
        return 8
    if "16x16x4_f32" in mn or "16x16x4f32" in mn:
        return 10
    return 18


def mfma_result_hazards(lib=LIB):
    """-> [(kernel, vector instruction, mfma, wait states seen, needed)]: every non-MFMA vector instruction that READS a VGPR an MFMA wrote fewer wait
    states earlier than the MFMA's passes need ('XDL write VGPR -> VALU read').  hipcc pads this between instructions it sees; the first read of a
    generator MFMA's result INSIDE an asm block (split3_pair<.., FIRST>) relies on the block's own leading `s_nop` plus the mask instructions in
    front of that read — this scan counts them in the built code."""
    bad = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kern, recent = "?", []                                   # recent: [(states since issue, vgprs written, needed, text)]
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kern, recent = m.group(1), []
                continue
            ins = line.split("//")[0].strip()
            if not ins or ins.startswith("."):
                continue
            mn, _, rest = ins.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            is_mfma = mn.startswith("v_mfma") or mn.startswith("v_smfma")
            if mn.startswith("v_") and not is_mfma:
                srcs = set()
                for o in ops[1:]:
                    k, r = _regs(o.split(" ")[0])
                    if k == "v":
                        srcs |= r
                for age, regs, need, text in recent:
                    if age < need and regs & srcs:
                        bad.append((kern, ins, text, age, need))
            states = (int(rest.strip().split()[0], 0) + 1) if mn == "s_nop" else 1
            recent = [(a + states, r, n, t) for a, r, n, t in recent if a + states < n]
            if is_mfma:
                k, r = _regs(ops[0].split(" ")[0])
                if k == "v":
                    recent.insert(0, (0, r, _mfma_result_wait(mn), ins))
    names = demangle([b[0] for b in bad])
    return [(n,) + b[1:] for n, b in zip(names, bad)]


if __name__ == "__main__":
    if "--hazards" in sys.argv:
        hz = mfma_operand_hazards()
        for h in hz:
            print(h)
        print(len(hz), "VALU-write -> MFMA-read pairs closer than 2 wait states")
        hr = mfma_result_hazards()
        for h in hr[:20]:
            print(h)
        print(len(hr), "MFMA-write -> VALU-read pairs closer than the MFMA's passes need")
        sys.exit(1 if (hz or hr) else 0)
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else LIB
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    res = kernel_resources(lib)
    print(f"{'kernel':110s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'LDS (static)':>13s} {'scratch':>8s}")
    for nm in sorted(res):
        if flt in nm:
            r = res[nm]
            print(f"{nm[:110]:110s} {r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['lds']:13d} {r['scratch']:8d}")
