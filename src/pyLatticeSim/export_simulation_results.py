"""Minimal stand-in for the reference's Paraview exporter (src/pyLatticeSim/export_simulation_results.py:35): writes
the lattice struts with nodal displacement / rotation as a legacy-VTK polydata file.  Post-processing is outside the
accelerated path; this exists so that the reference's simulation example runs end to end."""
import os

import numpy as np


class exportSimulationResults:
    def __init__(self, simulation_model, name_file="lattice", out_dir="data/outputs/simulation_results"):
        self.model = simulation_model
        self.name = os.path.basename(str(name_file))
        self.out_dir = out_dir
        self._fields = {}

    def export_displacement_rotation(self):
        self._fields["displacement"] = np.asarray(self.model.u)[:, :3]
        self._fields["rotation"] = np.asarray(self.model.u)[:, 3:]

    def export_data_homogenization(self, homogenization_surface: bool = True):
        """One file per macro-strain case with the total displacement field (reference :197-230; the polar stiffness
        surface goes through gmsh there and is not written here)."""
        paths = []
        for case, u_tot in enumerate(self.model.saveDataToExport):
            self._fields = {"Displacement": np.asarray(u_tot)[:, :3], "Rotation": np.asarray(u_tot)[:, 3:]}
            paths.append(self.export_finalize(suffix=f"_case{case + 1}"))
        return paths

    def export_finalize(self, suffix=""):
        lat = self.model.lattice.lattice
        os.makedirs(self.out_dir, exist_ok=True)
        path = os.path.join(self.out_dir, f"{self.name}{suffix}.vtk")
        with open(path, "w") as fh:
            fh.write("# vtk DataFile Version 3.0\npylattice-mi355x result\nASCII\nDATASET POLYDATA\n")
            fh.write(f"POINTS {lat.n_nodes} double\n")
            np.savetxt(fh, lat.node_xyz, fmt="%.12g")
            fh.write(f"LINES {lat.n_beams} {3 * lat.n_beams}\n")
            np.savetxt(fh, np.c_[np.full(lat.n_beams, 2), lat.beam_conn], fmt="%d")
            if self._fields:
                fh.write(f"POINT_DATA {lat.n_nodes}\n")
                for name, arr in self._fields.items():
                    fh.write(f"VECTORS {name} double\n")
                    np.savetxt(fh, arr, fmt="%.12g")
        return path
