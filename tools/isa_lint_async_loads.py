#!/usr/bin/env python3
"""Lint for kernels whose operand loads are inline-asm global_load_* with counted s_waitcnt vmcnt waits
(k_delta_direct): hipcc does not know that those loads are asynchronous, so nothing stops it from reading, copying
or re-using a destination register between the load and the wait that covers it -- which it did twice while the
kernel was written (copies of ring registers in front of a tied wait; accumulators moved into ring registers whose
surplus loads were still in flight).  This walks the kernel's assembly in text order with a model of the wave's load
queue (loads retire in order; `s_waitcnt vmcnt(N)` leaves the N youngest outstanding; the state at a branch is carried
to its target) and reports every instruction that touches a register with a load outstanding on it.

usage: isa_lint_async_loads.py file.s kernel_symbol_substring [--hazards-only]   (exit status 1 when something is found)"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_valu_to_mfma(lines, need=2):
    """A vector-ALU instruction of an inline-asm block that writes a VGPR, and an MFMA that reads that VGPR as an A / B
    operand fewer than `need` wait states later: hipcc puts ONE wait state behind an inline-asm output, the matrix
    instruction needs two (seen: k_delta_direct's second rest piece multiplied with the select's OLD register)."""
    problems = []
    recent = []  # (register set, wait states since)
    in_asm = False
    for no, raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        op = line.split()[0]
        ops = line[len(op):].split(",")
        if op.startswith("v_mfma"):
            src = regs(",".join(ops[1:3]))
            for written, ws in recent:
                if src & written and ws < need:
                    problems.append((no, raw.strip(), sorted(src & written)))
        states = int(line.split()[1]) + 1 if op == "s_nop" else 1
        recent = [(w, ws + states) for w, ws in recent if ws + states < 8]
        if in_asm and op.startswith("v_") and not op.startswith("v_mfma"):
            recent.append((regs(ops[0]), 0))
    return problems


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_valu_sgpr_to_vmem(lines, need=5):
    """A vector-ALU instruction that writes an SGPR (v_readlane / v_readfirstlane: hipcc's SGPR spills live in VGPR
    lanes; v_cmp with an SGPR destination) and an inline-asm memory instruction that takes that SGPR as its base fewer
    than `need` wait states later.  hipcc's hazard recognizer pads its OWN memory instructions; it does not look into
    an inline-asm block (seen: k_delta_direct's restart path with the top layer's delta in the launch -- SGPR pressure,
    bases restored by v_readlane directly in front of the loads, `Memory access fault ... address (nil)`)."""
    problems = []
    recent = []  # (SGPR set, wait states since)
    in_asm = False
    for no, raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        op = line.split()[0]
        ops = line[len(op):].split(",")
        if in_asm and (op.startswith("global_") or op.startswith("buffer_")):
            used = sregs(",".join(ops[1:]))
            for written, ws in recent:
                if used & written and ws < need:
                    problems.append((no, raw.strip(), ["s%d" % r for r in sorted(used & written)]))
        states = int(line.split()[1]) + 1 if op == "s_nop" else 1
        recent = [(w, ws + states) for w, ws in recent if ws + states < 8]
        if op.startswith("v_") and sregs(ops[0]):
            recent.append((sregs(ops[0]), 0))
    return problems


def lint(lines):
    queue = []          # outstanding loads, oldest first: sets of destination registers
    saved = {}          # label -> queue at the first branch that targets it
    problems = []
    reachable = True    # False behind an unconditional branch, until the next label
    for no, raw in lines:
        line = raw.split(";")[0].strip()
        m = re.match(r"^(\.LBB\w+):", raw.strip())
        if m:
            if not reachable and m.group(1) in saved:
                queue = [set(q) for q in saved[m.group(1)]]
            reachable = True
            continue
        if not line or line.startswith(".") or not reachable:
            continue
        op = line.split()[0]
        # waves run with every lane active in uniform control flow: `execnz` is always taken, `execz` never
        if op == "s_cbranch_execz":
            continue
        if op.startswith("s_cbranch") or op == "s_branch":
            saved.setdefault(line.split()[-1], [set(q) for q in queue])
            reachable = op not in ("s_branch", "s_cbranch_execnz")
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if 0 < n < len(queue) else ([] if n == 0 else queue)
            continue
        if op == "s_endpgm":
            break
        pending = set().union(*queue) if queue else set()
        if op.startswith("global_load") or op.startswith("buffer_load"):
            ops = line[len(op):].split(",")
            dst = regs(ops[0])
            src = regs(",".join(ops[1:]))
            # an address register that is also (part of) the destination is read at issue: fine
            if (src - dst) & pending:
                problems.append((no, raw.strip(), sorted((src - dst) & pending)))
            if dst & pending:
                problems.append((no, raw.strip(), sorted(dst & pending)))
            queue.append(dst)
            continue
        if op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("buffer_store"):
            queue.append(set())  # stores count in vmcnt too
        hit = regs(line) & pending
        if hit:
            problems.append((no, raw.strip(), sorted(hit)))
    return problems


def main():
    path, sym = sys.argv[1], sys.argv[2]
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith("_Z") and sym in l and l.split(";")[0].rstrip().endswith(":"))
    body = []
    for i in range(start + 1, len(text)):
        body.append((i + 1, text[i]))
        if "s_endpgm" in text[i]:
            break
    # --hazards-only (round 6, k_chain_persist): the two wait-state rules without the load-queue walk -- that kernel mixes
    # hipcc's own loads (which hipcc tracks) with inline-asm ones behind `s_waitcnt vmcnt(0)`, and its A fragments are LDS
    # reads with counted lgkmcnt waits across a loop's back edge, which the queue model (one in-order vmcnt queue, state
    # carried to a label from the first branch that targets it) does not describe
    hazards_only = len(sys.argv) > 3 and sys.argv[3] == "--hazards-only"
    problems = ([] if hazards_only else lint(body)) + lint_valu_to_mfma(body) + lint_valu_sgpr_to_vmem(body)
    loads = sum(1 for _, l in body if l.strip().startswith("global_load"))
    print("%s: %d instructions, %d global loads, %d problems" % (sym, len(body), loads, len(problems)))
    for no, line, hit in problems[:40]:
        print("  line %d: %s   <- %s" % (no, line, ("reads %s too early after a vector-ALU write" % hit) if hit and
                                       isinstance(hit[0], str) else ("touches v%s with a load outstanding" % hit)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
