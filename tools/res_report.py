"""Resource report of a HIP unit: kernel, VGPRs, AGPRs, SGPRs, scratch, occupancy (from -Rpass-analysis=kernel-resource-usage
on stdin)."""
import re, sys
name = None
vals = {}
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for l in sys.stdin:
    if "error" in l:
        print(l.rstrip())
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        name, vals = m.group(1), {}
    for key, rx in (("V", r" VGPRs: (\d+)"), ("A", r"AGPRs: (\d+)"), ("S", r" SGPRs: (\d+)"), ("scr", r"ScratchSize \[bytes/lane\]: (\d+)"),
                    ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(rx, l)
        if m:
            vals[key] = m.group(1)
    if "LDS Size" in l and name and pat in name:
        short = re.sub(r"_ZN5csmpn\d+|INS_3AlgILi|EvNS_8DevCemlpENS_5RowIOE", "", name)
        print(short, vals)
