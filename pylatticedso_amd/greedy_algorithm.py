"""Reduced basis of a family of cell Schur complements by greedy pursuit (host side, numpy).

Mirrors ``src/pyLatticeSim/greedy_algorithm.py:35-233`` of the reference: the snapshots (matrices flattened
column-major, normalised) are approximated by an orthonormal basis grown one vector at a time - always the snapshot
whose residual has the largest infinity norm, orthogonalised against the basis so far - until every residual is below
``tol * max |snapshot|``; the coefficients ``alpha_ortho`` are the least-squares coordinates of the UN-normalised
snapshots in that basis.  The result (``basis_reduced_ortho``, ``alpha_ortho``, ``list_elements``) is the ``.npz`` the
surrogate DDM modes read (``schur_surrogate.py``).  Together with ``construct_schur_complement_dataset`` on the GPU
this closes the surrogate tool chain without dolfinx.
"""
import os

import numpy as np

from .schur_surrogate import reduced_basis_file_name


def find_name_file_reduced_basis(lattice_object_sim, tol_greedy):
    """greedy_algorithm.py:214-233 (without the extension, as the reference returns it)."""
    return reduced_basis_file_name(lattice_object_sim.geom_types, tol_greedy)[:-len(".npz")]


def save_reduced_basis(file_name, basis_reduced_ortho, alpha_ortho, list_elements, root=None):
    """greedy_algorithm.py:156-183: ``<root>/data/outputs/schur_complement/reduced_basis/<file_name>.npz``."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "data", "outputs", "schur_complement", "reduced_basis", file_name)
    if not path.endswith(".npz"):
        path += ".npz"
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, basis_reduced_ortho=basis_reduced_ortho, alpha_ortho=alpha_ortho,
                        list_elements=list_elements)
    print(f"Reduced basis saved to {path}")
    return path


def reduce_basis_greedy(schur_complement_dict_to_reduce, tol_greedy, file_name=None, verbose=1, root=None):
    """Returns the reference's 7-tuple ``(mainelem, reducedcoef, projfieldpp, basis_reduced_ortho, alpha_ortho,
    matP_sorted, norm_mainelem_sorted)`` (greedy_algorithm.py:35-154)."""
    if not isinstance(schur_complement_dict_to_reduce, dict):
        raise ValueError("schur_complement_dict_to_reduce should be a dict of Schur complements.")
    keys = sorted(schur_complement_dict_to_reduce.keys())
    list_elements = np.array(keys)
    mats = np.array([schur_complement_dict_to_reduce[k] for k in keys])
    n_data = len(keys)
    fields = np.stack([np.ravel(m, order="F") for m in mats], axis=1)          # (n_S^2, n_data)
    norms = np.linalg.norm(fields, axis=0)
    resid = fields / norms
    atol = tol_greedy * np.abs(resid).max()
    basis, coefs, main = [], [], []
    while len(main) < n_data:
        s = int(np.argmax(np.abs(resid).max(axis=0)))
        vec = resid[:, s] / np.linalg.norm(resid[:, s])
        coef = resid.T @ vec
        resid -= np.outer(vec, coef)
        basis.append(vec)
        coefs.append(coef)
        main.append(s)
        if np.abs(resid).max() < atol:
            break
    main = np.array(main)
    reducedcoef = np.stack(coefs)                                               # (m, n_data)
    matP = np.triu(reducedcoef[:, main])
    reducedcoef = np.linalg.solve(matP, reducedcoef) if len(main) else reducedcoef
    reducedcoef = reducedcoef * np.outer(1.0 / norms[main], norms)
    basis_reduced_ortho = np.column_stack(basis) if basis else np.empty((fields.shape[0], 0))
    alpha_ortho = np.linalg.lstsq(basis_reduced_ortho, fields, rcond=None)[0]   # (m, n_data)
    vsort = np.argsort(main)
    if file_name is not None:
        save_reduced_basis(file_name, basis_reduced_ortho, alpha_ortho, list_elements, root=root)
    if verbose >= 1:
        print("Number of elements in the reduced basis:", len(main))
        print("Selected elements:", list_elements[main[vsort]])
    return (main[vsort], reducedcoef[vsort, :], [mats[i] for i in main[vsort]], basis_reduced_ortho, alpha_ortho,
            matP[np.ix_(vsort, vsort)], norms[main[vsort]])
