"""CPU, build container only (skipped where /root/reference is absent): with this package's
directory in front of the reference on sys.path, the reference's OWN task model builds on the
HIP-backed layers (namespace-package shadowing)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "clifford-group-equivariant-simplicial-message-passing-networks_amd")
REF = "/root/reference"

SCRIPT = r"""
import sys
sys.dont_write_bytecode = True
sys.path.insert(0, {golden!r}); import pyg_standin; pyg_standin.install()
sys.path.insert(0, {ref!r}); sys.path.insert(0, {pkg!r})
from csmpn.models.hulls_cssmpnn import HullsCliffordSharedSimplicialMPNN      # reference file
import csmpn.models.cegnn_utils as ours, csmpn.algebra.cliffordalgebra as alg
assert ours.__file__.startswith({pkg!r}), ours.__file__
assert alg.__file__.startswith({pkg!r}), alg.__file__
import csmpn.models.hulls_cssmpnn as hm
assert hm.__file__.startswith({ref!r}), hm.__file__
m = HullsCliffordSharedSimplicialMPNN()
assert all(type(l) is ours.EGCL for l in m.layers)
assert type(m.algebra) is alg.CliffordAlgebra and m.algebra.hip_supported
print(sum(p.numel() for p in m.parameters()))
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference not present (GPU box)")
def test_reference_model_builds_on_hip_layers():
    code = SCRIPT.format(golden=os.path.join(ROOT, "tests", "golden"), ref=REF, pkg=PKG_DIR)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "254329"     # hulls model parameter count
