"""Per-phase instruction mix of a kernel from its assembly text: counts VALU / MFMA / LDS / VMEM / scratch instructions
between the `; cb-phase N` markers (CB_MARK in cemlp_cmb.hpp). Usage: asm_phases.py file.s <substring of the kernel name>"""
import collections, re, sys

def main(path, key):
    lines = open(path).read().split("\n")
    cur, phase, seq = None, "pre", []
    out = collections.OrderedDict()
    for l in lines:
        m = re.match(r"^(\S+):\s*;\s*@", l) or re.match(r"^(_Z\S+):", l)
        if m:
            cur = m.group(1) if key in m.group(1) else None
            phase, n = "pre", 0
            continue
        if cur is None:
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
            continue
        t = l.strip()
        m = re.match(r";\s*cb-phase (\d+)", t)
        if m:
            phase = f"after {m.group(1)} #{len([k for k in out if k[0] == cur])}"
            continue
        if not t or t.startswith((".", ";")) or t.endswith(":"):
            continue
        c = out.setdefault((cur, phase), collections.Counter())
        op = t.split()[0]
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if "dpp" in t: c["dpp"] += 1
            if op.startswith("v_accvgpr"): c["acc_mov"] += 1
            if op.startswith("v_mov"): c["v_mov"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    last = None
    for (k, ph), c in out.items():
        if k != last:
            print("==", k[-60:])
            last = k
        print(f"  {ph:16s} " + " ".join(f"{n}={c[n]}" for n in ("valu", "dpp", "v_mov", "acc_mov", "mfma", "lds", "vmem", "scratch", "waitcnt", "salu") if c[n]))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "cemlp_cmb")
