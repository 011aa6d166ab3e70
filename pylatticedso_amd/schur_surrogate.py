"""Surrogate Schur complements of a unit cell over a reduced basis:  S(r) ~ reshape(B @ alpha(r)).

Mirrors what the reference does when ``schur_complement_computation.type`` is ``nearest_neighbor``, ``linear`` or
``RBF`` (``src/pyLatticeSim/lattice_sim.py:126-135,755-813,919-977,1056-1082``, ``utils_rbf.py``,
``greedy_algorithm.py:186-233``):

* the reduced basis ``B`` (``basis_reduced_ortho``, (n_S^2, m)), the coefficients of the training cells
  (``alpha_ortho``, (m, N)) and their radii (``list_elements``, (N, d)) come from the ``.npz`` the reference's greedy
  algorithm wrote under ``data/outputs/schur_complement/reduced_basis/``;
* ``alpha(r)`` is interpolated from the training set - nearest training point, piecewise-linear, or a thin-plate
  spline with a linear polynomial tail;
* ``S = reshape(B @ alpha, (n, n), order="F")`` and, for the spline, ``dS/dr_j = reshape(B @ dalpha/dr_j)``.

Host-side and tiny (a few training points, m <= ~10^2): it feeds the device palette of cell matrices
(``LatticeSim.set_schur_complements`` -> ``pl_create_ddm``); nothing here is on the per-iteration path.
"""
import os
import re

import numpy as np


def _single_threaded_blas():
    """Context in which numpy's BLAS runs on the calling thread only.  The worker threads of a threaded BLAS keep spinning for a
    while after a GEMM; a device solve that follows within milliseconds then shares the job's CPUs with them and its launch
    loop crawls (measured: 4.3 -> 32 ms per solve_DDM at 16^3 cells on a 16-CPU share).  These GEMMs are small: one thread does."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1, user_api="blas")
    except Exception:                     # (no threadpoolctl: as before)
        import contextlib
        return contextlib.nullcontext()

KINDS = ("nearest_neighbor", "linear", "RBF")


def reduced_basis_file_name(geom_types, tol):
    """``reduced_basis_<geoms>_tol_<1e-6>`` (greedy_algorithm.py:214-233)."""
    suffix = "_".join(re.sub(r"\W+", "-", str(g)) for g in geom_types)
    tol_str = re.sub(r"e([+-])0+(\d+)$", r"e\1\2", f"{tol:.0e}")
    return f"reduced_basis_{suffix}_tol_{tol_str}.npz"


def find_reduced_basis(geom_types, tol, search_dirs=None):
    """Look for the reduced-basis file where the reference keeps it (``<root>/data/outputs/schur_complement/
    reduced_basis/``), under every root in ``search_dirs``, ``$PYLATTICE_DATA_ROOT`` and this repository."""
    name = reduced_basis_file_name(geom_types, tol)
    roots = list(search_dirs or [])
    if os.environ.get("PYLATTICE_DATA_ROOT"):
        roots.append(os.environ["PYLATTICE_DATA_ROOT"])
    roots.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    tried = []
    for root in roots:
        for sub in (os.path.join("data", "outputs", "schur_complement", "reduced_basis"), ""):
            path = os.path.join(root, sub, name)
            tried.append(path)
            if os.path.isfile(path):
                return path
    raise FileNotFoundError("Reduced basis file not found: " + name + " (looked in " + ", ".join(tried) + ")")


class ThinPlateSpline:
    """f(x) = sum_i w_i phi(|x - x_i|) + c0 + c.x with phi(r) = r^2 log r, vector valued; weights from the usual
    saddle system [[Phi, P], [P^T, 0]] [w; c] = [y; 0] (utils_rbf.py:20-62)."""

    def __init__(self, x_train, y_train):
        X = np.asarray(x_train, dtype=float)
        Y = np.asarray(y_train, dtype=float)
        if X.ndim == 1:
            X = X[:, None]
        if Y.ndim == 1:
            Y = Y[:, None]
        n, d = X.shape
        P = np.hstack([np.ones((n, 1)), X])
        A = np.zeros((n + d + 1, n + d + 1))
        A[:n, :n] = self._phi(self._dist(X, X))
        A[:n, n:] = P
        A[n:, :n] = P.T
        sol = np.linalg.solve(A, np.vstack([Y, np.zeros((d + 1, Y.shape[1]))]))
        self.x, self.w, self.c = X, sol[:n], sol[n:]

    @staticmethod
    def _dist(a, b):
        return np.linalg.norm(a[:, None, :] - b[None, :, :], axis=2)

    @staticmethod
    def _phi(r):
        out = np.zeros_like(r)
        m = r > 0
        out[m] = r[m] ** 2 * np.log(r[m])
        return out

    def evaluate(self, xq):
        """(M, d) -> (M, m)."""
        Xq = np.atleast_2d(np.asarray(xq, dtype=float))
        with _single_threaded_blas():
            return self._phi(self._dist(Xq, self.x)) @ self.w + np.hstack([np.ones((len(Xq), 1)), Xq]) @ self.c

    def gradient(self, xq):
        """(M, d) -> (M, d, m):  sum_i w_i (2 log r_i + 1)(x - x_i) + c."""
        Xq = np.atleast_2d(np.asarray(xq, dtype=float))
        D = Xq[:, None, :] - self.x[None, :, :]
        r = np.linalg.norm(D, axis=2)
        fac = np.zeros_like(r)
        m = r > 0
        fac[m] = 2.0 * np.log(r[m]) + 1.0
        return np.einsum("qnd,nk->qdk", fac[:, :, None] * D, self.w) + self.c[None, 1:, :]


class SchurSurrogate:
    def __init__(self, basis, alpha_ortho, list_elements, kind):
        if kind not in KINDS:
            raise NotImplementedError("Not implemented schur complement computation method.")
        self.kind = kind
        self.basis = np.asarray(basis, dtype=float)                      # (n_S^2, m)
        self.points = np.asarray(list_elements, dtype=float)             # (N, d)
        if self.points.ndim == 1:
            self.points = self.points[:, None]
        self.alpha_train = np.asarray(alpha_ortho, dtype=float).T        # (N, m)  (lattice_sim.py:132)
        if self.alpha_train.shape[0] != self.points.shape[0]:
            raise ValueError(f"Incompatible shapes: points={self.points.shape}, values={self.alpha_train.shape}.")
        self.n = int(round(np.sqrt(self.basis.shape[0])))
        self._tps = None
        self._lin = None

    @classmethod
    def load(cls, geom_types, tol, kind, search_dirs=None):
        d = np.load(find_reduced_basis(geom_types, tol, search_dirs))
        return cls(d["basis_reduced_ortho"], d["alpha_ortho"], d["list_elements"], kind)

    # ---- alpha(r) --------------------------------------------------------------------------------------------
    def alphas(self, radii_batch):
        """(n_q, d) radii -> (n_q, m) coefficients."""
        Xq = np.atleast_2d(np.asarray(radii_batch, dtype=float))
        if Xq.shape[1] != self.points.shape[1]:
            raise ValueError(f"expected {self.points.shape[1]} radii per cell, got {Xq.shape[1]}")
        if self.kind == "nearest_neighbor":
            i0 = np.argmin(np.linalg.norm(Xq[:, None, :] - self.points[None, :, :], axis=2), axis=1)
            return self.alpha_train[i0]
        if self.kind == "RBF":
            if self._tps is None:
                self._tps = ThinPlateSpline(self.points, self.alpha_train)
            return self._tps.evaluate(Xq)
        # linear (lattice_sim.py:755-807): 1-D -> np.interp clamped at the ends; N-D -> Delaunay-linear inside the
        # hull of the training points, nearest training point outside
        if self.points.shape[1] == 1:
            order = np.argsort(self.points[:, 0])
            xs, A = self.points[order, 0], self.alpha_train[order]
            return np.stack([np.interp(Xq[:, 0], xs, A[:, j]) for j in range(A.shape[1])], axis=1)
        if self._lin is None:
            from scipy.interpolate import LinearNDInterpolator, NearestNDInterpolator
            self._lin = (LinearNDInterpolator(self.points, self.alpha_train),
                         NearestNDInterpolator(self.points, self.alpha_train))
        y = np.asarray(self._lin[0](Xq))
        bad = np.isnan(y).any(axis=1)
        if bad.any():
            y[bad] = np.asarray(self._lin[1](Xq[bad]))
        return y

    # ---- S(r), dS/dr -----------------------------------------------------------------------------------------
    def _matrices(self, coeffs):
        # the basis vectors are column-major images of (n, n) matrices (lattice_sim.py:919-977 reshapes with order "F"): with
        # the rows of the basis permuted to row-major ONCE, one GEMM writes the (n_q, n, n) stack as it is wanted (the
        # transposing copy of the first version was 40 ms per 4 096 matrices, twice per design iteration)
        bt = self.__dict__.get("_basis_rowmajor_T")
        if bt is None:
            n, m = self.n, self.basis.shape[1]
            bt = self._basis_rowmajor_T = np.ascontiguousarray(self.basis.reshape(n, n, m).transpose(1, 0, 2).reshape(n * n, m).T)
        with _single_threaded_blas():
            return (np.atleast_2d(np.asarray(coeffs, dtype=float)) @ bt).reshape(-1, self.n, self.n)

    def schur_batch(self, radii_batch):
        """(n_q, d) -> (n_q, n, n)  (lattice_sim.py:919-977)."""
        return self._matrices(self.alphas(radii_batch))

    def schur_gradients_batch(self, radii_batch):
        """(n_q, d) -> (n_q, d, n, n): dS/dr_j of every radius set in ONE spline evaluation and ONE GEMM (RBF surrogate; the
        per-set calls of schur_gradients were 0.2 ms each - 4 096 distinct cells: 0.8 s per design iteration).  None for the
        surrogates whose gradient is a finite difference."""
        if self.kind != "RBF":
            return None
        Xq = np.atleast_2d(np.asarray(radii_batch, dtype=float))
        if self._tps is None:
            self._tps = ThinPlateSpline(self.points, self.alpha_train)
        g = self._tps.gradient(Xq)                                           # (n_q, d, m)
        nq, d, m = g.shape
        return self._matrices(g.reshape(nq * d, m)).reshape(nq, d, self.n, self.n)

    def schur_gradients(self, radii, eps_rel=1e-6):
        """[dS/dr_j for every radius j].  RBF: analytic through the spline (lattice_sim.py:1056-1082); the other
        surrogates: central differences of S(r) with the reference's step rule (lattice_sim.py:1020-1054)."""
        radii = [float(r) for r in radii]
        if self.kind == "RBF":
            if self._tps is None:
                self._tps = ThinPlateSpline(self.points, self.alpha_train)
            g = self._tps.gradient(np.asarray(radii)[None, :])[0]          # (d, m)
            return list(self._matrices(g))
        grads = []
        for j, rj in enumerate(radii):
            h = max(1e-8, eps_rel * max(1.0, abs(rj)))
            rp, rm = list(radii), list(radii)
            rp[j], rm[j] = rj + h, max(1e-12, rj - h)
            Sp, Sm = self.schur_batch([rp, rm])
            grads.append((Sp - Sm) / (rp[j] - rm[j]))
        return grads
