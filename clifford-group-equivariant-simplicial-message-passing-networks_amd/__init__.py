"""MI355X-native drop-in for the Clifford / CEMLP / EGCL hot path of
congliuUvA/Clifford-Group-Equivariant-Simplicial-Message-Passing-Networks.

Importing this package puts its directory on sys.path so that
  csmpn.algebra.cliffordalgebra.CliffordAlgebra
  csmpn.models.cegnn_utils.{MVLinear, MVSiLU, NormalizationLayer,
      SteerableGeometricProductLayer, MVLayerNorm, CEMLP, EGCL}
resolve to the HIP-backed modules (same import paths, constructor signatures
and state_dict keys as the reference).
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import csmpn_hip  # noqa: E402
from csmpn_hip import native, ops  # noqa: E402,F401
from csmpn.algebra.cliffordalgebra import CliffordAlgebra  # noqa: E402,F401
from csmpn.models.cegnn_utils import (  # noqa: E402,F401
    CEMLP,
    EGCL,
    MVLayerNorm,
    MVLinear,
    MVSiLU,
    NormalizationLayer,
    SteerableGeometricProductLayer,
)

SimplicialMessagePassing = EGCL  # BASELINE.json's name for the shared simplicial layer
PACKAGE_DIR = _HERE
