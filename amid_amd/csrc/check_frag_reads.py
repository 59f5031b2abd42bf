#!/usr/bin/env python3
"""Build-time check of the hand-placed LDS fragment reads (strip_gemm.h lds_frag_issue / lds_frag_wait, used by the products on bf16
pieces in sasrec_strip.hip and sasrec_seqn.hip).

Those reads are inline asm: `ds_read_b128 vD, vA offset:N` issued several steps ahead of their use, and the wait is a second asm statement,
`s_waitcnt lgkmcnt(N)` with N counted by hand (LDS operations return in order).  Two things the compiler does not know and therefore
cannot keep true by itself:
  (1) the destination registers of such a read are NOT valid until its wait: a copy, a spill or any other use the compiler places in
      between reads (or clobbers) stale data;
  (2) scalar memory loads count in lgkmcnt too and return OUT of order, so the counter alone does not say WHICH operations are back.
      What `s_waitcnt lgkmcnt(N)` does guarantee: at most N operations of any kind are outstanding -- hence at most N LDS operations, and
      since those return in order, every LDS operation older than the N youngest LDS operations is back, whatever the scalar loads did (an
      outstanding scalar load only makes the wait stricter).  The hand counts N = "asm reads issued behind this step's own" rely on exactly
      that and on nothing else; the model below assumes no more.
This script compiles the given sources to gfx950 assembly (device only) and walks every function's instruction stream with that model: an
in-order queue of outstanding LDS operations (the asm reads and the compiler's own); `s_waitcnt lgkmcnt(N)` retires all but the N youngest
of them.  It fails when
  * any instruction touches a register an outstanding ASM ds_read will write (rule 1: a use, a copy, a spill or a clobber before the wait),
  * an ASM ds_read is still outstanding at a label or a branch (the model is per straight-line region; the products are fully unrolled).
Scalar loads the compiler schedules inside a window are counted and reported, not failed.
    python check_frag_reads.py [-I include_dir] file.hip ...     exit status 0 = every product's reads and waits are consistent"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
WAIT = re.compile(r"lgkmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def check_asm(path_s, name):
    errors, n_reads, n_fn = [], 0, 0
    n_smem = [0]
    fn, in_asm = None, False
    queue = []          # outstanding LGKM operations, oldest first: ("asm" | "lds" | "smem", frozenset of destination registers, line number)

    def pending_asm():
        return [q for q in queue if q[0] == "asm"]

    with open(path_s) as f:
        for no, raw in enumerate(f, 1):
            line = raw.split("//")[0].strip()
            if not line:
                continue
            if line.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if line.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if line.startswith(";") or line.startswith("."):
                continue
            if line.endswith(":"):                        # a label: function entry or basic-block boundary
                if pending_asm():
                    errors.append(f"{name}:{no}: {fn}: {len(pending_asm())} asm ds_read(s) outstanding at label {line}")
                if not line.startswith(".L"):
                    fn = line[:-1]
                    n_fn += 1
                queue = []
                continue
            op = line.split()[0]
            rest = line[len(op):]
            if op.startswith("s_waitcnt"):
                m = WAIT.search(line)
                if m is not None or op == "s_waitcnt_lgkmcnt":
                    n = int(m.group(1)) if m else 0
                    queue = queue[len(queue) - n:] if n < len(queue) else queue
                    if n == 0:
                        queue = []
                continue
            pend = pending_asm()
            if pend:
                busy = set().union(*(q[1] for q in pend))
                hit = regs_of(rest) & busy
                is_own_issue = in_asm and op == "ds_read_b128"
                if hit and not is_own_issue:
                    errors.append(f"{name}:{no}: {fn}: `{line}` touches {sorted(hit)[:4]} while the asm ds_read of line {pend[0][2]} is outstanding")
                elif hit:      # a new asm read into a register whose previous read has not been waited for
                    errors.append(f"{name}:{no}: {fn}: asm ds_read re-targets {sorted(hit)[:4]} before the wait of the read at line {pend[0][2]}")
            if op.startswith(("s_load", "s_buffer_load", "s_scratch_load")):
                if pend:
                    n_smem[0] += 1                        # (benign: see the header)
                continue
            if op.startswith("ds_") or op.startswith("flat_"):
                if in_asm and op == "ds_read_b128":
                    dst = regs_of(rest.split(",")[0])
                    queue.append(("asm", frozenset(dst), no))
                    n_reads += 1
                else:
                    queue.append(("lds", frozenset(), no))
                continue
            if op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
                if pend:
                    errors.append(f"{name}:{no}: {fn}: {len(pend)} asm ds_read(s) outstanding at `{line}`")
                queue = []
    return errors, n_reads, n_fn, n_smem[0]


def main(argv):
    inc, files = [], []
    it = iter(argv)
    for a in it:
        if a == "-I":
            inc.append(next(it))
        elif a.startswith("-I"):
            inc.append(a[2:])
        else:
            files.append(a)
    bad = 0
    for src in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-Wno-unused-command-line-argument", "-o", tmp.name, src]
            for d in inc:
                cmd += ["-I", d]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                return 2
            errors, n_reads, n_fn, n_smem = check_asm(tmp.name, os.path.basename(src))
        print(f"check_frag_reads: {os.path.basename(src)}: {n_reads} hand-placed fragment reads in {n_fn} functions, {len(errors)} violations "
              f"({n_smem} scalar loads scheduled inside a window: they only make a wait stricter)")
        for e in errors[:40]:
            print("  " + e)
        bad += len(errors)
        if n_reads == 0:
            print(f"  {os.path.basename(src)}: no asm ds_read_b128 found -- the check no longer sees the products it was written for")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
