"""``from pyLatticeDesign.timing import timing, Timing`` (reference: src/pyLatticeDesign/timing.py:16-288)."""
from pylatticedso_amd.timing import Timing, timing  # noqa: F401
