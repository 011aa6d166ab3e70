"""MI355X-native stiffness assembly + solve for pyLatticeDSO's ``pyLatticeSim`` hot path.

``_capi``            ctypes binding of ``libpylattice_hip.so`` (include/pylattice_hip.h)
``lattice_arrays``   array-backed lattice generation, joint penalisation, gmsh subdivision counts
``lattice_sim``      ``LatticeSim`` mirror (boundary conditions, ``solve_DDM``, Schur complements of the cells)
``utils_simulation`` ``solve_FEM_FenicsX`` mirror;  ``utils_schur`` ``get_schur_complement`` and the dataset helpers
``schur_surrogate``, ``greedy_algorithm``   reduced-basis surrogates of the cell Schur complements
``lattice_opti``     ``LatticeOpti`` mirror (objective, adjoint gradient, SLSQP driver)
``partition``        slab partition for one-process-per-GPU runs

There is no CPU fallback: without the shared library or without a GPU the device calls raise.
"""
