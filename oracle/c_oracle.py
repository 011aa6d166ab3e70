"""ctypes access to oracle/_build/liboracle_pcg.so (plain-C CPU restatement; TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_pcg.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def condense(radius, seg_len, seg_nsub, E, nu, kappa=0.9, pen=1.5):
    lib = _lib()
    lib.oracle_condense.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.c_void_p]
    radius = np.asarray(radius, float)
    seg_len = np.ascontiguousarray(seg_len, dtype=np.float64).reshape(-1, 3)
    seg_nsub = np.ascontiguousarray(seg_nsub, dtype=np.int32).reshape(-1, 3)
    out = np.empty((len(radius), 5))
    for b in range(len(radius)):
        lib.oracle_condense(float(radius[b]), _p(seg_len[b]), _p(seg_nsub[b]), E, nu, kappa, pen, _p(out[b]))
    return out


def condense_all(radius, seg_len, seg_nsub, E, nu, kappa=0.9, pen=1.5):
    """Every strut's condensation in C on all cores (oracle_condense_all): the assembly half of the all-cores CPU leg."""
    lib = _lib()
    lib.oracle_condense_all.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_void_p]
    radius = np.ascontiguousarray(radius, dtype=np.float64)
    seg_len = np.ascontiguousarray(seg_len, dtype=np.float64).reshape(-1, 3)
    seg_nsub = np.ascontiguousarray(seg_nsub, dtype=np.int32).reshape(-1, 3)
    out = np.empty((len(radius), 5))
    lib.oracle_condense_all(len(radius), _p(radius), _p(seg_len), _p(seg_nsub), E, nu, kappa, pen, _p(out))
    return out


def condense_unique(radius, seg_len, seg_nsub, E, nu, kappa=0.9, pen=1.5):
    """Same as condense() but evaluates each distinct (r, lengths, n) once (big lattices)."""
    key = np.c_[np.asarray(radius, float), np.asarray(seg_len, float).reshape(-1, 3),
                np.asarray(seg_nsub).reshape(-1, 3)]
    uq, inv = np.unique(key, axis=0, return_inverse=True)
    sc = condense(uq[:, 0], uq[:, 1:4], uq[:, 4:7].astype(np.int32), E, nu, kappa, pen)
    return sc[inv.ravel()]


def spmv(xyz, conn, scalars, x):
    lib = _lib()
    lib.oracle_spmv_add.argtypes = [C.c_int64] + [C.c_void_p] * 5
    xyz = np.ascontiguousarray(xyz, np.float64)
    conn = np.ascontiguousarray(conn, np.int32)
    sc = np.ascontiguousarray(scalars, np.float64)
    x = np.ascontiguousarray(np.asarray(x, np.float64).reshape(-1))
    y = np.zeros_like(x)
    lib.oracle_spmv_add(len(conn), _p(xyz), _p(conn), _p(sc), _p(x), _p(y))
    return y.reshape(-1, 6)


def available_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a one-GPU
    job 16 of its 256 hardware threads - OpenMP's default of one thread per hardware thread oversubscribes 8 x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def set_threads(n):
    lib = _lib()
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_set_threads(int(n))


def num_threads():
    lib = _lib()
    lib.oracle_num_threads.restype = C.c_int
    return int(lib.oracle_num_threads())


def pcg(xyz, conn, scalars, fixed, ubar, f, rtol=1e-10, maxit=100000, all_cores=False):
    """Jacobi-PCG on the condensed struts; all_cores=True runs the OpenMP variant (same algorithm, per-node gather)."""
    lib = _lib()
    if all_cores:
        lib.oracle_pcg = lib.oracle_pcg_mt
    lib.oracle_pcg.argtypes = [C.c_int64, C.c_int64] + [C.c_void_p] * 6 + [C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_pcg.restype = C.c_int
    xyz = np.ascontiguousarray(xyz, np.float64)
    conn = np.ascontiguousarray(conn, np.int32)
    sc = np.ascontiguousarray(scalars, np.float64)
    fx = np.ascontiguousarray(np.asarray(fixed).reshape(-1) != 0, np.uint8)
    ub = np.ascontiguousarray(np.asarray(ubar, np.float64).reshape(-1))
    ff = np.ascontiguousarray(np.asarray(f, np.float64).reshape(-1))
    u = np.empty_like(ub)
    rel = C.c_double()
    it = lib.oracle_pcg(len(xyz), len(conn), _p(xyz), _p(conn), _p(sc), _p(fx), _p(ub), _p(ff), rtol, maxit, _p(u),
                        C.byref(rel))
    return u.reshape(-1, 6), it, rel.value
