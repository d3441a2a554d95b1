#!/usr/bin/env python3
"""dev: find vector instructions the compiler placed in a block that is only ever entered with EXEC = 0.

Round 4 found one in a variant of the fp64 energy kernel (forces mode with parked sums under the 168-register bound):
register allocation put the RELOAD of two spilled registers (the thread's group and lane numbers) into the exit block of
a divergent loop whose `s_or_b64 exec` had been merged into the enclosing region's (SILowerControlFlow removes an
end-of-control-flow restore that is directly followed by another one).  That block is reached only through
`s_cbranch_execz`, so the scratch loads execute with no lane enabled, the registers keep what the loop left in them and
the code after the join reads LDS rows by a garbage index: wrong, run-to-run different energies.

The scan, per kernel of a `hipcc -S --cuda-device-only` listing: a label whose preceding instruction does not fall
through (s_branch / s_endpgm / s_setpc) and that is targeted only by `s_cbranch_execz`; from the label to the first write of
EXEC, report every instruction that needs lanes (v_* other than v_readlane / v_writelane / v_readfirstlane, ds_*,
global_*, scratch_*, buffer_*, flat_*).

usage: check_exec0_reloads.py file.s [...]      exit code 1 if anything is reported
       check_exec0_reloads.py --lib libmythos_hip.so    the same scan over the disassembly of every gfx950 code object
                                                        inside the built library (what tests/test_api_cpu.py runs)
       check_exec0_reloads.py --scratch libmythos_hip.so   kernels of the built library that use scratch memory (both wrong
                                                        results of round 4 went with spill code at a register bound)
"""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)")
NO_FALL = re.compile(r"^\s+(s_branch|s_endpgm|s_setpc_b64)\b")
WRITES_EXEC = re.compile(r"^\s+s_\w+\s+exec(_lo|_hi)?\b|^\s+s_\w*saveexec\w*\s|^\s+v_cmpx_")
LANE_OP = re.compile(r"^\s+(v_|ds_|global_|scratch_|buffer_|flat_)")
LANE_FREE = re.compile(r"^\s+v_(readlane|writelane|readfirstlane)")
INSTR = re.compile(r"^\s+[a-z]")


def kernels(lines):
    name, start = None, 0
    for k, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            if name is not None:
                yield name, start, k
            name, start = m.group(1), k
    if name is not None:
        yield name, start, len(lines)


def scan(path):
    lines = open(path).read().split("\n")
    found = []
    for name, a, b in kernels(lines):
        body = lines[a:b]
        sources = {}
        for ln in body:
            m = BRANCH.match(ln)
            if m:
                sources.setdefault(m.group(2), set()).add(m.group(1))
        prev_instr = None
        aliases = []  # labels of the empty blocks directly in front of this one: they fall through into it
        for k, ln in enumerate(body):
            m = LABEL.match(ln)
            if m:
                label = m.group(1)
                aliases.append(label)
                into = set()
                for l in aliases:
                    into |= sources.get(l, set())
                if prev_instr is not None and NO_FALL.match(prev_instr) and into == {"s_cbranch_execz"}:
                    for j in range(k + 1, len(body)):
                        t = body[j]
                        if LABEL.match(t) or WRITES_EXEC.match(t) or BRANCH.match(t):
                            break
                        if LANE_OP.match(t) and not LANE_FREE.match(t):
                            found.append((name, a + j + 1, label, t.strip()))
            elif INSTR.match(ln) and not ln.lstrip().startswith((";", ".")):
                prev_instr = ln
                aliases = []
    return found


def code_objects(lib_path):
    """The device code objects of a hipcc-built shared library: its uncompressed clang offload bundles, one per unit."""
    blob = Path(lib_path).read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = blob.find(magic)
    while at >= 0:
        p = at + len(magic)
        (count,) = struct.unpack_from("<Q", blob, p)
        p += 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tlen].decode()
            p += tlen
            if "amdgcn" in triple and size:
                yield blob[at + off:at + off + size]
        at = blob.find(magic, at + 1)


DIS = re.compile(r"^\t(\S+)(.*?)//\s*([0-9A-F]+):")
FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:")
TARGET = re.compile(r"<([^>+]+)(?:\+0x([0-9a-f]+))?>\s*$")


def scan_disassembly(lines):
    """Same rule as scan(), on `llvm-objdump -d` text (addresses instead of labels)."""
    found = []
    funcs, cur = [], None
    for ln in lines:
        m = FUNC.match(ln)
        if m:
            cur = (m.group(2), int(m.group(1), 16), [])
            funcs.append(cur)
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        m = DIS.match(ln)
        if m:
            cur[2].append((int(m.group(3), 16), m.group(1), ln))
    for name, base, ins in funcs:
        into = {}
        for addr, op, ln in ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                t = TARGET.search(ln)
                if t and t.group(1) == name:
                    into.setdefault(base + int(t.group(2) or "0", 16), set()).add(op)
        for k in range(1, len(ins)):
            addr, op, ln = ins[k]
            if into.get(addr) != {"s_cbranch_execz"} or ins[k - 1][1] not in ("s_branch", "s_endpgm", "s_setpc_b64"):
                continue
            for j in range(k, len(ins)):
                a2, op2, l2 = ins[j]
                text = "\t" + op2 + l2.split("//")[0][len(op2) + 1:]
                if (j > k and a2 in into) or WRITES_EXEC.match(text) or op2.startswith(("s_cbranch", "s_branch")):
                    break
                if LANE_OP.match(text) and not LANE_FREE.match(text):
                    found.append((name, a2, text.strip()))
    return found


def scan_library(lib_path):
    found = []
    if not Path(OBJDUMP).exists():  # (a box without the ROCm LLVM tools: nothing to disassemble with - say so, do not fail the build)
        print(f"check_exec0_reloads: {OBJDUMP} not found, scan skipped", file=sys.stderr)
        return found
    with tempfile.TemporaryDirectory() as tmp:
        for k, obj in enumerate(code_objects(lib_path)):
            f = Path(tmp) / f"co{k}.elf"
            f.write_bytes(obj)
            out = subprocess.run([OBJDUMP, "-d", str(f)], check=True, capture_output=True, text=True).stdout
            found += scan_disassembly(out.split("\n"))
    return found


READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernels_with_scratch(lib_path):
    """{kernel symbol: private segment bytes} for every kernel of the library that has any (the code objects' metadata)."""
    out = {}
    if not Path(READELF).exists():
        print(f"check_exec0_reloads: {READELF} not found, scratch listing skipped", file=sys.stderr)
        return out
    with tempfile.TemporaryDirectory() as tmp:
        for k, obj in enumerate(code_objects(lib_path)):
            f = Path(tmp) / f"co{k}.elf"
            f.write_bytes(obj)
            notes = subprocess.run([READELF, "--notes", str(f)], check=True, capture_output=True, text=True).stdout
            name = None
            size = 0
            for ln in notes.split("\n"):
                t = ln.strip()
                if t.startswith("- .agpr_count:") or t.startswith("- .args:"):
                    name, size = None, 0
                if t.startswith(".private_segment_fixed_size:"):
                    size = int(t.split(":")[1])
                if t.startswith(".name:"):
                    name = t.split(":", 1)[1].strip()
                if t.startswith(".symbol:") and name is not None and size > 0:
                    out[name] = size
    return out


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--scratch":
        rows = kernels_with_scratch(sys.argv[2])
        for name, size in sorted(rows.items()):
            print(f"{size:5d} B  {name[:120]}")
        print(f"{len(rows)} kernel(s) with scratch")
        return 0
    if len(sys.argv) == 3 and sys.argv[1] == "--lib":
        hits = scan_library(sys.argv[2])
        for name, addr, text in hits:
            print(f"{addr:#x}: entered with EXEC = 0 only, yet holds: {text}\n    in {name[:100]}")
        print(f"{len(hits)} instruction(s) in EXEC = 0 blocks")
        return 1 if hits else 0
    bad = 0
    for path in sys.argv[1:]:
        hits = scan(path)
        for name, line, label, text in hits:
            print(f"{path}:{line}: {label} is entered with EXEC = 0 only, yet holds: {text}\n    in {name[:100]}")
        bad += len(hits)
    print(f"{bad} instruction(s) in EXEC = 0 blocks")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
