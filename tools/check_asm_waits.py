"""Build-time audit of the hand-ordered LDS reads (cemlp_rl.hpp / cemlp_cl.hpp): an inline-asm `ds_read_*`
whose `s_waitcnt` sits in a LATER asm statement leaves its destination VGPRs in flight between the two statements; the
hardware has no scoreboard for LDS returns, so a compiler-inserted copy or spill of those registers in between would
read stale data silently. This script assembles a unit to text (hipcc -S) and walks every kernel: from each asm
`ds_read_*` it follows the instruction stream to the first `s_waitcnt` with lgkmcnt(0) and fails if any instruction
outside an asm block reads or writes one of the destination registers on the way. It also fails on scratch use in the
units that must not spill.

    python tools/check_asm_waits.py <file.s> [--no-scratch PATTERN ...]
"""
import re
import sys


def regs_of(tok):
    """v12 -> {12}; v[4:7] -> {4,5,6,7}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def audit(path, no_scratch=()):
    lines = open(path).read().split("\n")
    kernel, in_asm, errors, checked = None, False, [], 0
    pending = []   # (dest registers, line number) of asm ds_reads not yet waited for
    for n, raw in enumerate(lines, 1):
        line = raw.strip()
        m = re.match(r"(_Z\S+):", raw)
        if m:
            kernel, pending, in_asm = m.group(1), [], False
            continue
        if not kernel or not line or line.startswith(".") and not line.startswith(".LBB"):
            continue
        if ";#ASMSTART" in line.replace(" ", ""):
            in_asm = True
            continue
        if ";#ASMEND" in line.replace(" ", ""):
            in_asm = False
            continue
        if line.startswith(";") or line.endswith(":"):
            continue
        op = line.split()[0]
        if any(re.search(p, kernel) for p in no_scratch) and op.startswith("scratch_"):
            errors.append(f"{path}:{n}: scratch access in {kernel[:80]}: {line}")
        if op.startswith("s_waitcnt") and ("lgkmcnt(0)" in line or line.strip() == "s_waitcnt 0" or "vmcnt(0) lgkmcnt(0)" in line):
            pending = []
            continue
        if in_asm and op.startswith("ds_read"):
            dest = line.split(",")[0]
            pending.append((regs_of(dest), n))
            checked += 1
            continue
        if pending and not in_asm:
            used = regs_of(line.split(";")[0])
            for dest, at in pending:
                if used & dest:
                    errors.append(f"{path}:{n}: {kernel[:60]}: `{line}` touches v{sorted(used & dest)} of the asm ds_read at line {at} "
                                  f"before its s_waitcnt")
    return checked, errors


if __name__ == "__main__":
    args = sys.argv[1:]
    no_scratch = []
    while "--no-scratch" in args:
        i = args.index("--no-scratch")
        no_scratch.append(args[i + 1])
        del args[i:i + 2]
    total, errs = 0, []
    for f in args:
        c, e = audit(f, no_scratch)
        total += c
        errs += e
    print(f"check_asm_waits: {total} asm LDS reads followed to their wait, {len(errs)} finding(s)")
    for e in errs[:40]:
        print("  " + e)
    sys.exit(1 if errs else 0)
