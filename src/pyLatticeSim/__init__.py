"""Import-compatible front of the reference's ``pyLatticeSim`` package (src/pyLatticeSim/): the hot path is served by
``pylatticedso_amd`` (HIP library on MI355X); everything else of the reference package is out of scope."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
