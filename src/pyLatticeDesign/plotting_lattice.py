"""Minimal stand-in for the reference's ``LatticePlotting`` (src/pyLatticeDesign/plotting_lattice.py:58): a matplotlib
3-D line plot of the struts (optionally in deformed configuration), saved to a file instead of opening a window when
no display is available.  Visualisation is outside the accelerated path."""
import os

import numpy as np


class LatticePlotting:
    def __init__(self, out_dir="data/outputs/plots"):
        self.out_dir = out_dir

    def visualize_lattice(self, lattice_object, beam_color_type="radii", deformed_form=False,
                          enable_boundary_conditions=False, file_save_path=None, **_ignored):
        import matplotlib
        if not os.environ.get("DISPLAY"):
            matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        from mpl_toolkits.mplot3d.art3d import Line3DCollection
        lat = lattice_object.lattice
        xyz = lat.node_xyz.copy()
        if deformed_form:
            xyz = xyz + 5.0 * lattice_object.displacement_vector[:, :3]    # Point.magnification_factor (point.py:75)
        seg = xyz[lat.beam_conn]
        fig = plt.figure(figsize=(8, 6))
        ax = fig.add_subplot(111, projection="3d")
        col = Line3DCollection(seg, linewidths=0.6, cmap="viridis")
        col.set_array(lat.beam_radius if beam_color_type == "radii" else lat.beam_type.astype(float))
        ax.add_collection3d(col)
        if enable_boundary_conditions:
            fx = lattice_object.fixed_DOF.any(axis=1)
            ld = np.any(lattice_object.applied_force != 0, axis=1)
            ax.scatter(*xyz[fx].T, c="k", s=6)
            ax.scatter(*xyz[ld].T, c="r", s=6)
        ax.set_xlim(xyz[:, 0].min(), xyz[:, 0].max())
        ax.set_ylim(xyz[:, 1].min(), xyz[:, 1].max())
        ax.set_zlim(xyz[:, 2].min(), xyz[:, 2].max())
        if matplotlib.get_backend().lower() == "agg" or file_save_path:
            os.makedirs(self.out_dir, exist_ok=True)
            path = file_save_path or os.path.join(self.out_dir, "lattice.png")
            fig.savefig(path, dpi=120)
            plt.close(fig)
            return path
        plt.show()

    def subplot_lattice_hybrid_geometries(self, lattice, explode_voxel: float = 0.0, file_save_path=None):
        """One panel per geometry of a hybrid lattice, struts coloured by radius (plotting_lattice.py:637-700)."""
        import matplotlib
        if not os.environ.get("DISPLAY"):
            matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        from mpl_toolkits.mplot3d.art3d import Line3DCollection
        lat = lattice.lattice
        geoms = list(lattice.geom_types)
        if len(geoms) <= 1:
            print("Lattice is not hybrid; only one geometry type found.")
        fig = plt.figure(figsize=(5 * len(geoms), 5))
        for g, name in enumerate(geoms):
            ax = fig.add_subplot(1, len(geoms), g + 1, projection="3d")
            sel = lat.beam_type == g
            col = Line3DCollection(lat.node_xyz[lat.beam_conn[sel]], linewidths=0.6, cmap="viridis")
            col.set_array(lat.beam_radius[sel])
            col.set_clim(0.025, 0.1)
            ax.add_collection3d(col)
            ax.set_xlim(lattice.x_min, lattice.x_max)
            ax.set_ylim(lattice.y_min, lattice.y_max)
            ax.set_zlim(lattice.z_min, lattice.z_max)
            ax.set_title(str(name))
        if matplotlib.get_backend().lower() == "agg" or file_save_path:
            os.makedirs(self.out_dir, exist_ok=True)
            path = file_save_path or os.path.join(self.out_dir, "lattice_hybrid_geometries.png")
            fig.savefig(path, dpi=120)
            plt.close(fig)
            return path
        plt.show()
