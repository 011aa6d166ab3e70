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
        self._cell_fields = {}          # per-segment vectors (the reference's DG0 functions: Forces, Moment, a1, a2, t)
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

    # ---- per-segment (DG0) fields of the reference: local frame, section forces and moments ---------------------------
    def _segment_geometry(self):
        """Per exported segment: mid-point, unit tangent and the local frame (t, a1, a2) of
        BeamModel.calculate_local_coordinate_system (beam_model.py:197-216)."""
        from pylatticedso_amd.views import _tables
        t = _tables(self.model.lattice)
        x1, x2 = t.node_xyz[t.beam_conn[:, 0]], t.node_xyz[t.beam_conn[:, 1]]
        tan = x2 - x1
        tan /= np.linalg.norm(tan, axis=1)[:, None]
        ex, ey, ez = np.eye(3)
        e1 = np.where((np.abs(tan[:, 1]) < np.abs(tan[:, 0]))[:, None], ey, ex)
        te1 = np.einsum("ij,ij->i", tan, e1)
        e2 = np.where((np.abs(tan[:, 2]) < np.abs(te1))[:, None], ez, e1)
        a1 = np.cross(tan, e2)
        a1 /= np.linalg.norm(a1, axis=1)[:, None]
        a2 = np.cross(tan, a1)
        a2 /= np.linalg.norm(a2, axis=1)[:, None]
        return t, 0.5 * (x1 + x2), tan, a1, a2

    def export_local_coordinates_system(self):
        """Cell fields "t", "a1", "a2" (reference :120-144)."""
        _, _, tan, a1, a2 = self._segment_geometry()
        self._cell_fields.update({"a1": a1, "a2": a2, "t": tan})

    def _section_forces(self, u=None):
        """Section force and moment vectors at the mid-point of every segment, as exerted by the material on the +t side
        on the material on the -t side (the sign of sig = C eps, simulation_base.py:116-125): along an unloaded strut the
        force F is constant and the moment about a point q is M_B + (x_B - q) x F, with (F, M_B) the tip force of the
        condensed record."""
        sim = self.model.lattice
        lat = sim.lattice
        u = np.asarray(self.model.u if u is None else u, dtype=float).reshape(-1, 6)
        rec = self.model.device.records()
        a, c, e1, e2, e3, d = rec[:, 0:1], rec[:, 1:2], rec[:, 2:3], rec[:, 3:4], rec[:, 4:5], rec[:, 5:8]
        A, B = lat.beam_conn[:, 0], lat.beam_conn[:, 1]
        du = u[B, :3] - u[A, :3] + np.cross(d, u[A, 3:])
        dth = u[B, 3:] - u[A, 3:]
        F = a * du + e1 * np.einsum("ij,ij->i", du, d)[:, None] * d + e2 * np.cross(d, dth)
        M = c * dth + e3 * np.einsum("ij,ij->i", dth, d)[:, None] * d - e2 * np.cross(d, du)
        t, mid, tan, a1, a2 = self._segment_geometry()
        par = t.beam_parent
        sign = np.sign(np.einsum("ij,ij->i", tan, d[par]))[:, None]
        Fi = sign * F[par]
        Mi = sign * (M[par] + np.cross(lat.node_xyz[B[par]] - mid, F[par]))
        loc = lambda v: np.stack([np.einsum("ij,ij->i", v, tan), np.einsum("ij,ij->i", v, a1),      # noqa: E731
                                  np.einsum("ij,ij->i", v, a2)], axis=1)
        return loc(Fi), loc(Mi)

    def export_internal_force(self, u=None, case: int = 0):
        """Cell field "Forces" = (N, V_a1, V_a2) in the local frame (reference :146-160, calculate_forces :160-165)."""
        self._cell_fields[f"Forces_{case}" if case else "Forces"] = self._section_forces(u)[0]

    def export_moments(self, FE_result=None, case: int = 0):
        """Cell field "Moment" = (M_t, M_a1, M_a2) at the segment mid-points (reference :94-107, calculate_moments :169-174)."""
        self._cell_fields[f"Moment_{case}" if case else "Moment"] = self._section_forces(FE_result)[1]

    def full_export(self, case: int = 0):
        self.export_displacement_rotation(case)
        self.export_reaction_force()
        self.export_moments(case=case)
        self.export_internal_force(case=case)
        self.export_local_coordinates_system()
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
        self._write_vtu_pvd(t, sim, suffix, time)
        return path

    def _write_vtu_pvd(self, t, sim, suffix, time):
        """The same data as one VTU piece + the PVD collection the reference's dolfinx.io.VTKFile produces
        (``<name>.pvd`` next to ``<name>_p0_000000.vtu``, reference :49-63,163-176), plus the per-segment fields."""
        stem = f"{self.name}{suffix}"
        vtu = f"{stem}_p0_000000.vtu"

        def arr(a, name, comps, kind="Float64"):
            a = np.asarray(a)
            fmt = "%.12g" if kind == "Float64" else "%d"
            body = "\n".join(" ".join(fmt % v for v in np.atleast_1d(row)) for row in a)
            nm = f' Name="{name}"' if name else ""
            return f'<DataArray type="{kind}"{nm} NumberOfComponents="{comps}" format="ascii">\n{body}\n</DataArray>\n'
        nb, nn = t.n_beams, t.n_nodes
        with open(os.path.join(self.out_dir, vtu), "w") as fh:
            fh.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">\n'
                     f'<UnstructuredGrid>\n<Piece NumberOfPoints="{nn}" NumberOfCells="{nb}">\n')
            fh.write("<Points>\n" + arr(t.node_xyz, "", 3) + "</Points>\n")
            fh.write("<Cells>\n" + arr(t.beam_conn, "connectivity", 1, "Int64")
                     + arr(2 * np.arange(1, nb + 1), "offsets", 1, "Int64")
                     + arr(np.full(nb, 3), "types", 1, "UInt8") + "</Cells>\n")
            fh.write("<PointData>\n" + arr((np.arange(nn) >= t.n_design_nodes).astype(int), "node_mod", 1, "Int64"))
            for name, a in self._fields.items():
                fh.write(arr(a, name, 3))
            fh.write("</PointData>\n<CellData>\n" + arr(t.beam_radius, "radius", 1)
                     + arr(t.beam_mod.astype(int), "beam_mod", 1, "Int64")
                     + arr(sim.lattice.beam_type[t.beam_parent], "type_beam", 1, "Int64"))
            for name, a in self._cell_fields.items():
                fh.write(arr(a, name, 3))
            fh.write("</CellData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")
        self.pvd_path = os.path.join(self.out_dir, f"{stem}.pvd")
        with open(self.pvd_path, "w") as fh:
            fh.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n<Collection>\n'
                     f'<DataSet timestep="{float(time)}" part="0" file="{vtu}" />\n</Collection>\n</VTKFile>\n')
        self.result_to_export.append(self.pvd_path)

    def write_function(self, time: float = 0.0):
        return self.export_finalize(time)

    def close_file(self):
        return None
