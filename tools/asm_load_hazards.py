#!/usr/bin/env python3
"""Static check of hipcc's output for hand-issued (inline asm) loads.

The front kernel (16-byte coherent exchange loads) and the f32 covariance update (operand ring) issue
loads from inline asm and wait for them with hand-written `s_waitcnt vmcnt(n)`.  The compiler does not
know that the destination registers of such a load are still in flight: a register copy, a spill or a
re-use it places between the load and the wait that covers it reads / clobbers the OLD register content
whenever memory is slower than the instruction stream (there is no hardware interlock on VGPRs), i.e.
a result that depends on timing.  This script walks the generated assembly in program order, keeps the
FIFO of outstanding vector-memory instructions (vmcnt counts all of them, in order) and reports every
instruction that touches a register of an inline-asm load that no wait has covered yet.

Straight-line approximation: labels do not reset the FIFO (conservative).  The sources are written so
that no divergent branch sits between an inline-asm load and its wait.

usage: asm_load_hazards.py file.s [file.s ...]   -> exit code 1 if a hazard is found
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic",
        "flat_load", "flat_store", "flat_atomic", "scratch_load", "scratch_store")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(path):
    hazards = []
    kernel = None
    in_asm = False
    fifo = []             # outstanding VMEM instructions, oldest first: (set of guarded vgprs, line, text)
    n_loads = 0
    for ln, raw in enumerate(open(path), 1):
        stripped = raw.lstrip()
        if stripped.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if stripped.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line:
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            fifo = []
            continue
        if line.endswith(":") or line.startswith("."):
            continue
        op = line.split()[0]
        if op == "s_endpgm":
            fifo = []
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m:
                keep = int(m.group(1))
                fifo = fifo[len(fifo) - keep:] if keep else []
            continue
        guarded = set()
        for g, _, _ in fifo:
            guarded |= g
        if op.startswith(VMEM):
            ops = line[len(op):].split(",")
            dest = regs(ops[0]) if (in_asm and "load" in op) else set()
            # operands other than the destination must not be in flight either
            touched = regs(",".join(ops[1:]) if dest else line) & guarded
            if touched:
                hazards.append((kernel, ln, line, sorted(touched)))
            if dest:
                n_loads += 1
            fifo.append((dest, ln, line))
            continue
        touched = regs(line) & guarded
        if touched:
            src = next(t for g, _, t in fifo if g & touched)
            hazards.append((kernel, ln, line + "      <- in flight: " + src, sorted(touched)))
    return n_loads, hazards


def main():
    bad = 0
    for path in sys.argv[1:]:
        n, hz = scan(path)
        print(f"{path}: {n} inline-asm loads, {len(hz)} hazards")
        for kernel, ln, line, touched in hz:
            print(f"  {kernel}\n    line {ln}: {line}\n    touches v{touched}")
        bad += len(hz)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
