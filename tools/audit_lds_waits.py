#!/usr/bin/env python3
"""Audit of the hand-counted LDS waits in the fused MLP kernels' generated code.

The fp16-pair kernel (csrc/mlp_kernel_h2.hip) issues its LDS reads from inline asm and wait for
them with hand-counted `s_waitcnt lgkmcnt(N)` statements, which pins the ORDER of the statements but not what hipcc
does with the destination registers in between (cdna_hip_programming.md, "What hipcc does not do", item 1). This script
walks the assembly of a kernel (hipcc -save-temps output) in program order, keeps the queue of LDS operations in
flight (LDS returns in order: `lgkmcnt(N)` retires all but the newest N) and reports every instruction that reads or
writes a register whose ds_read has not been retired yet. Loop bodies are straight-line code; at a label or branch the
queue is kept (a back edge re-enters with the same pattern).

    python tools/audit_lds_waits.py nerf-projects_amd/build/mlp_kernel_h2-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-substring]
"""
import re
import sys

NUM = r"(0x[0-9a-fA-F]+|\d+)"
REG = re.compile(r"\b([va])\[" + NUM + ":" + NUM + r"\]|\b([va])\[" + NUM + r"\]|\b([va])(\d+)\b")


def regs(text):
    """Registers named in an operand list: v12, a[4:7], and the forms hand-allocated code is printed in (v[0xb0:0xb3], a[0x83])."""
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2), 0), int(m.group(3), 0) + 1))
        elif m.group(4):
            out.add((m.group(4), int(m.group(5), 0)))
        else:
            out.add((m.group(6), int(m.group(7))))
    return out


def audit(path, want="kernelILi2ELb0E"):
    lines = open(path).read().split("\n")
    findings, n_reads, n_waits = [], 0, 0
    name, queue = None, []          # queue: list of (line_no, dest_regs) of LDS ops in flight, oldest first
    for no, raw in enumerate(lines, 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            name, queue = (line[:-1] if want in line else None), []
            continue
        if name is None or line.startswith("."):
            continue
        if line.startswith("s_endpgm"):
            name = None
            continue
        op, _, rest = line.partition(" ")
        m = re.match(r"s_waitcnt\b(.*)", line)
        if m:
            c = re.search(r"lgkmcnt\((\d+)\)", line)
            if c:
                n_waits += 1
                keep = int(c.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            continue
        touched = regs(rest)
        pending = set().union(*[d for _, d in queue]) if queue else set()
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            # the address operand of a new LDS op may not be a pending destination either
            dest = regs(rest.split(",")[0]) if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle") else set()
            srcs = touched - dest
            # a second LDS read into a register still in flight is harmless (in-order return: the later data lands
            # last; hipcc does this for destinations nobody consumes, e.g. the T1 fragments of a one-row layer)
            if srcs & pending:
                findings.append((no, raw.strip(), sorted(srcs & pending)[:4]))
            queue.append((no, dest))
            n_reads += 1
            continue
        if touched & pending:
            findings.append((no, raw.strip(), sorted(touched & pending)[:4]))
    return findings, n_reads, n_waits


SREG = re.compile(r"\bs\[" + NUM + ":" + NUM + r"\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1), 0), int(m.group(2), 0) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit_sgpr_hazards(path, want="kernelILi2ELb0E", need=5):
    """A vector-memory instruction (the inline-asm stores / atomics / LDS-DMA with a scalar base) must not read a scalar
    register that a VECTOR instruction (v_readlane, v_readfirstlane - e.g. the reload of a spilled SGPR - or a v_cmp into an
    SGPR pair) wrote fewer than `need` wait states earlier: hipcc's hazard recogniser inserts those s_nops for its own
    instructions but does not look inside inline asm. Returns the offending instructions."""
    lines = open(path).read().split("\n")
    findings, name = [], None
    age = {}                       # sgpr -> wait states since a VALU wrote it
    for no, raw in enumerate(lines, 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        if line.endswith(":") and not line.startswith("."):
            name, age = (line[:-1] if want in line else None), {}
            continue
        if name is None or line.startswith("."):
            continue
        if line.startswith("s_endpgm"):
            name = None
            continue
        op, _, rest = line.partition(" ")
        if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            used = sregs(rest)
            bad = sorted(r for r in used if age.get(r, 99) < need)
            if bad:
                findings.append((no, raw.strip(), [(r, age[r]) for r in bad]))
        states = 1
        if op == "s_nop":
            states = int(rest.strip(), 0) + 1
        for r in list(age):
            age[r] += states
        if op.startswith(("v_readlane", "v_readfirstlane")) or (op.startswith("v_cmp") and rest.lstrip().startswith("s[")) \
                or (op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64", "v_mad_i64"))):
            for r in sregs(rest.split(",")[0] if not op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad")) else
                           ",".join(rest.split(",")[:2])):
                age[r] = 0
        elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch")):
            for r in sregs(rest.split(",")[0]):
                age.pop(r, None)    # a scalar instruction rewrote it: no hazard towards vector memory
    return findings


if __name__ == "__main__":
    f, r, w = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "kernelILi2ELb0E")
    print(f"{sys.argv[1]}: {r} LDS operations, {w} lgkmcnt waits, {len(f)} accesses to registers still in flight")
    for no, text, which in f[:40]:
        print(f"  line {no}: {text}    <- {which}")
    hz = audit_sgpr_hazards(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "kernelILi2ELb0E")
    print(f"{len(hz)} vector-memory instructions reading a scalar register a vector instruction has just written")
    for no, text, which in hz[:40]:
        print(f"  line {no}: {text}    <- (sgpr, wait states) {which}")
    sys.exit(1 if (f or hz) else 0)
