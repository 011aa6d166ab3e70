"""``pyLatticeSim.utils`` of the reference (utils.py:19-148): directory clean-up, directional stiffness modulus and the
homogenisation figure.  The modulus is evaluated for the whole (theta, phi) grid at once (the reference calls a
four-fold Python loop 80 601 times); matplotlib is only imported when a figure is asked for, with the Agg backend when
there is no display (the reference forces TkAgg at import, utils.py:16)."""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

from .homogenization_cell import directional_modulus        # noqa: F401  (utils.py:35-73, re-exported)


def clear_directory(directoryPath):
    """utils.py:19-33: remove the files (not the sub-directories) of a directory."""
    for name in os.listdir(directoryPath):
        p = os.path.join(directoryPath, name)
        if os.path.isfile(p):
            os.remove(p)


def directional_modulus_grid(matS, thetavalues, phivalues):
    """(n_theta, n_phi, 3) vectors E(theta, phi) u of ``directional_modulus`` for a grid of angles in degrees."""
    th, ph = np.deg2rad(np.asarray(thetavalues))[:, None], np.deg2rad(np.asarray(phivalues))[None, :]
    u = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th) * np.ones_like(ph)], axis=-1)
    idx = np.array([[0, 3, 4], [3, 1, 5], [4, 5, 2]])
    coef = np.where(np.eye(3, dtype=bool), 1.0, 2.0)
    S4 = np.asarray(matS)[idx[:, :, None, None], idx[None, None, :, :]] / (coef[:, :, None, None] * coef[None, None, :, :])
    inv_e = np.einsum("ijkl,tpi,tpj,tpk,tpl->tp", S4, u, u, u, u)
    return u / inv_e[..., None]


def create_homogenization_figure(mat_S_orthotropic, plot: bool = True, save: bool = False,
                                 name_file: str = "homogenization_figure"):
    """utils.py:75-148: 3-D surface of the directional stiffness modulus.  Returns the path of the saved image (or None)."""
    n = 200
    thetavalues = np.linspace(0, 180, n + 1)
    phivalues = np.linspace(0, 360, 2 * n + 1)
    data = directional_modulus_grid(mat_S_orthotropic, thetavalues, phivalues)
    X, Y, Z = data[..., 0], data[..., 1], data[..., 2]
    if not (plot or save):
        return None
    import matplotlib
    if not os.environ.get("DISPLAY") or not plot:
        matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    norm = plt.Normalize(vmin=0, vmax=float(np.sqrt(X ** 2 + Y ** 2 + Z ** 2).max()))
    facecolors = plt.cm.jet(norm(np.sqrt(X ** 2 + Y ** 2 + Z ** 2)))
    fig = plt.figure()
    ax = fig.add_subplot(111, projection="3d")
    ax.plot_surface(X, Y, Z, facecolors=facecolors, rstride=4, cstride=4)
    ax.set_proj_type("ortho")
    ax.set_aspect("equal")
    ax.set_xticks([])
    ax.set_yticks([])
    ax.set_zticks([])
    ax.grid(False)
    m = plt.cm.ScalarMappable(cmap=plt.cm.jet, norm=norm)
    m.set_array([])
    fig.colorbar(m, ax=ax, shrink=1, aspect=20, orientation="vertical").set_label("Directional Stiffness [GPa]")
    path = None
    if save:
        root = Path(__file__).resolve().parent.parent
        path = root / "data" / "outputs" / "simulation_results" / "figure_homogenization" / name_file
        path.parent.mkdir(parents=True, exist_ok=True)
        if path.suffix != ".png":
            path = path.with_suffix(".png")
        plt.savefig(path, dpi=150, bbox_inches="tight")
        print(f"Saved image to {path}")
    if plot and matplotlib.get_backend().lower() != "agg":
        plt.show()
    plt.close(fig)
    return None if path is None else str(path)
