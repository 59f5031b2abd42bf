#!/usr/bin/env python3
"""A model of the vector-memory counter over one kernel's gfx950 assembly (hipcc -S --cuda-device-only): does any instruction read (or
overwrite) a register that an earlier global / buffer load has not yet delivered, given the `s_waitcnt vmcnt(N)` the compiler placed?

vmcnt counts loads, stores and LDS-DMA together in issue order (MI355X_MICROARCH.md): `s_waitcnt vmcnt(N)` retires all but the N youngest.
The walk follows the text in order and takes every BACKWARD conditional branch once with the queue carried over (loads issued at the end
of a loop body and consumed at the top of the next iteration), then falls through.

    python profiles/tools/probe/vmcnt_model.py file.s kernel_label_substring      -> one line per hazard, or "no hazard"

Written to root-cause round 4's open finding (csrc/wgrad_split.h: zeroing the rows past a split's end by a 0 / 1 multiplication gave
wrong sums with two workgroups per CU; by selection it does not): see profiles/tools/probe/wgrad_opsel_repro.sh."""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
VM = re.compile(r"vmcnt\((\d+)\)")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def kernel_lines(path, needle):
    body, cap = [], False
    for ln in open(path):
        ln = ln.rstrip("\n")
        head = ln.split(";")[0].rstrip()
        if not cap and head.endswith(":") and needle in head and not head.startswith(".L") and not ln.lstrip().startswith((";", ".")):
            cap = True
            continue
        if cap and ln.strip().startswith(".Lfunc_end"):
            break
        if cap:
            body.append(ln)
    return body


def main(path, needle, counter="vm"):
    raw = kernel_lines(path, needle)
    ins, labels = [], {}
    for ln in raw:
        t = ln.split("//")[0].split(";")[0].strip() if not ln.strip().startswith(";;#") else ""
        if not t or t.startswith("."):
            if t.endswith(":"):
                labels[t[:-1]] = len(ins)
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(ins)
            continue
        ins.append(t)
    queue, hazards, taken = [], [], set()     # queue: (kind, dest regs, index)
    pc, steps = 0, 0
    while pc < len(ins) and steps < 20 * len(ins):
        steps += 1
        t = ins[pc]
        op = t.split()[0]
        rest = t[len(op):]
        if op.startswith("s_waitcnt"):
            m = (VM if counter == "vm" else LGKM).search(t)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n else []
            pc += 1
            continue
        busy = {}
        for k, dst, at in queue:
            if k == "load":
                for r in dst:
                    busy[r] = at
        if counter == "vm":
            is_vmem = op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "global_atomic", "buffer_atomic", "scratch_"))
        else:       # the LDS side of lgkmcnt: ds_* return in order; scalar loads (out of order) are assumed back only at lgkmcnt(0)
            is_vmem = op.startswith("ds_")
        touched = regs_of(rest)
        hit = touched & set(busy)
        if hit:
            hazards.append((pc, t, sorted(hit)[:4], ins[busy[sorted(hit)[0]]]))
        if is_vmem:
            if ("load" in op and "lds" not in rest) or op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")):
                queue.append(("load", frozenset(regs_of(rest.split(",")[0])), pc))
            else:
                queue.append(("other", frozenset(), pc))
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = rest.strip()
            if tgt in labels and labels[tgt] <= pc and (pc, tgt) not in taken:
                taken.add((pc, tgt))
                pc = labels[tgt]
                continue
        pc += 1
    n_loads = sum(1 for t in ins if t.startswith(("global_load", "buffer_load")))
    print(f"{path}: kernel *{needle}*: {len(ins)} instructions, {n_loads} vector loads; counter {counter}: {len(hazards)} reads / clobbers of a register with its load outstanding")
    for pc, t, regs, ld in hazards[:20]:
        print(f"  [{pc}] {t}    <- {regs} of `{ld}`")
    return 1 if hazards else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "vm"))
