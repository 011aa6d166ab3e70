"""Design-only lattice: the base class ``pyLatticeDesign.lattice.Lattice`` (lattice.py:36-109) on the array-backed model.

``LatticeSim`` / ``LatticeOpti`` of this package derive from the same array model; ``Lattice`` is that model with the
simulation layer switched off - geometry, gradients, cells / beams / nodes views, bounding box, relative density - what
``Tests/Lattice_test.py`` of the reference and its design-side examples use.  No GPU involved."""
from __future__ import annotations

import numpy as np

from .lattice_sim import LatticeSim


class Lattice(LatticeSim):
    def __init__(self, name_file, mesh_trimmer=None, verbose: int = 0):
        self.name_lattice = name_file if isinstance(name_file, str) else "lattice"
        super().__init__(name_file, mesh_trimmer=mesh_trimmer, verbose=verbose)

    def define_simulation_parameters(self, name_file):
        """The base class reads no simulation block: no material penalisation, no boundary conditions."""
        super().define_simulation_parameters(name_file)
        self.enable_simulation_properties = False
        self.boundary_conditions = {}

    # -- lattice.py:163-199 ------------------------------------------------------------------------------------------
    def __repr__(self) -> str:
        return (f"Lattice name_lattice: {self.name_lattice}\n"
                f"Dimensions: {self.size_x} x {self.size_y} x {self.size_z}\n"
                f"Number of cells: {self.num_cells_x} x {self.num_cells_y} x {self.num_cells_z}\n"
                f"Cell size: {self.cell_size_x} x {self.cell_size_y} x {self.cell_size_z}\n"
                f"radii: {self.radii}\n")

    def __eq__(self, other):
        if not isinstance(other, LatticeSim):
            return NotImplemented
        a, b = self._cell_parameter_radii(), other._cell_parameter_radii()
        return a.shape == b.shape and bool(np.all(np.abs(a - b) <= 1e-9))

    __hash__ = object.__hash__
