"""Result export for Paraview (cf. the reference's src/pyLatticeSim/export_simulation_results.py:35-200, which writes
dolfinx functions on the gmsh sub-mesh to VTU/PVD).  Here the FE model is the condensed one: the file holds every lattice
node AND every penalisation point (``node_mod``, displacements recovered by pl_node_mod), the segments between them as
line cells with their radius / penalised flag / geometry type, nodal displacement, rotation, applied and reaction forces -
legacy-VTK polydata, which Paraview's Tube filter turns into the strut picture.  Post-processing is outside the
accelerated path; same call names as the reference so that its simulation examples run end to end."""
import os

import numpy as np


class exportSimulationResults:
    def __init__(self, simulation_model, name_file="lattice", out_dir="data/outputs/simulation_results"):
        self.simulation_model = self.model = simulation_model
        self.name = os.path.basename(str(name_file))
        self.out_dir = out_dir
        self._fields = {}
        self.result_to_export = []

    # ---- what goes into the file ------------------------------------------------------------------------------
    def _all_nodes(self, lattice_rows):
        """(n_all, k) array: the given per-lattice-node rows followed by the penalisation points' rows."""
        sim = self.model.lattice
        from pylatticedso_amd.views import _tables
        t = _tables(sim)
        extra = t.n_nodes - t.n_design_nodes
        lattice_rows = np.asarray(lattice_rows, dtype=float)
        if extra == 0:
            return lattice_rows
        tail = np.zeros((extra, lattice_rows.shape[1]))
        return np.concatenate([lattice_rows, tail])

    def export_displacement_rotation(self, case: int = 0):
        sim = self.model.lattice
        u = np.asarray(self.model.u, dtype=float)
        full = self._all_nodes(u)
        if len(full) > len(u):
            full[len(u):] = sim._node_mod_rows("displacement_vector")      # back-substituted on first access
        self._fields["displacement"] = full[:, :3]
        self._fields["rotation"] = full[:, 3:]

    def export_reaction_force(self, lattice_data=None):
        sim = self.model.lattice
        self._fields["reaction_force"] = self._all_nodes(sim.reaction_force_vector[:, :3])
        self._fields["reaction_moment"] = self._all_nodes(sim.reaction_force_vector[:, 3:])
        self._fields["applied_force"] = self._all_nodes(sim.applied_force[:, :3])

    def full_export(self, case: int = 0):
        self.export_displacement_rotation(case)
        self.export_reaction_force()
        return self.export_finalize()

    def export_data_homogenization(self, homogenization_surface: bool = True):
        """One file per macro-strain case with the total displacement field (reference :197-230; the polar stiffness
        surface goes through gmsh there and is not written here)."""
        paths = []
        for case, u_tot in enumerate(self.model.saveDataToExport):
            self._fields = {"Displacement": self._all_nodes(np.asarray(u_tot)[:, :3]),
                            "Rotation": self._all_nodes(np.asarray(u_tot)[:, 3:])}
            paths.append(self.export_finalize(suffix=f"_case{case + 1}"))
        return paths

    # ---- writer -------------------------------------------------------------------------------------------------
    def export_finalize(self, time: float = 0.0, suffix=""):
        sim = self.model.lattice
        from pylatticedso_amd.views import _tables
        t = _tables(sim)
        os.makedirs(self.out_dir, exist_ok=True)
        path = os.path.join(self.out_dir, f"{self.name}{suffix}.vtk")
        nb, nn = t.n_beams, t.n_nodes
        with open(path, "w") as fh:
            fh.write("# vtk DataFile Version 3.0\npylattice-mi355x result\nASCII\nDATASET POLYDATA\n")
            fh.write(f"POINTS {nn} double\n")
            np.savetxt(fh, t.node_xyz, fmt="%.12g")
            fh.write(f"LINES {nb} {3 * nb}\n")
            np.savetxt(fh, np.c_[np.full(nb, 2), t.beam_conn], fmt="%d")
            fh.write(f"CELL_DATA {nb}\n")
            for name, arr, kind in (("radius", t.beam_radius, "double"), ("beam_mod", t.beam_mod.astype(int), "int"),
                                    ("type_beam", sim.lattice.beam_type[t.beam_parent], "int")):
                fh.write(f"SCALARS {name} {kind} 1\nLOOKUP_TABLE default\n")
                np.savetxt(fh, arr, fmt="%.12g" if kind == "double" else "%d")
            fh.write(f"POINT_DATA {nn}\n")
            fh.write("SCALARS node_mod int 1\nLOOKUP_TABLE default\n")
            np.savetxt(fh, (np.arange(nn) >= t.n_design_nodes).astype(int), fmt="%d")
            for name, arr in self._fields.items():
                fh.write(f"VECTORS {name} double\n")
                np.savetxt(fh, arr, fmt="%.12g")
        self.result_to_export.append(path)
        return path

    def write_function(self, time: float = 0.0):
        return self.export_finalize(time)

    def close_file(self):
        return None
