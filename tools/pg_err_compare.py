"""Per-tensor errors of one test_lane_kernel_shapes case (relative to the float64 oracle) - run once per kernel family
(CSMPN_NO_PG=1 for the wide parity-lane kernels) to compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib, torch
pkg = importlib.import_module("clifford-group-equivariant-simplicial-message-passing-networks_amd")
import test_hip_parity as T
import contextlib
metric = [1.0, 1.0, 1.0, 1.0, -1.0]
det = os.environ.get("DET", "1") == "1"
ctx = T.deterministic_aggregation() if det else contextlib.nullcontext()
orig = T.check
rows = []
def check(name, hip, truth, ref32=None, tol=T.TOL, slack=4.0):
    rows.append((name, T.relmax(hip, truth), T.relmax(ref32, truth) if ref32 is not None else 0.0))
    return rows[-1][1]
T.check = check
with ctx:
    T._oracle_egcl_case(metric, 130, 1027, 32, 32, "mean", seed=130 + 1027, residual=True, neg_scale=0.02, attr_grad=True, slack=6.0)
for n, e, y in sorted(rows, key=lambda r: -r[1] / max(r[2], 1e-9))[:int(os.environ.get("TOP", "8"))]:
    print(f"{n:48s} err {e:.2e}  ref32 {y:.2e}  ratio {e / max(y, 1e-12):.1f}")
