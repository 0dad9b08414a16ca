"""Condenses the rocprofv3 output of tools/profile_r0N.sh into the two files kept under
profiles/: the kernel-trace stats table (csv, as rocprofv3 wrote it, cemlp/csr kernels first)
and a per-kernel JSON of the PMC counters averaged per launch."""
import collections, csv, glob, json, os, shutil, sys

out, tag = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r01"
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join("gpurun_out", f"{rnd}_{tag}_kernel_stats.csv"))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "csmpn" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
summary = {k: {c: int(round(v / cnt[k][c])) for c, v in sorted(d.items())} for k, d in acc.items()}
with open(os.path.join("gpurun_out", f"{rnd}_{tag}_pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps({k[-70:]: v for k, v in summary.items()}, indent=1)[:6000])
