"""The timing collector (pylatticedso_amd/timing.py) against the behaviour the reference's own Tests/Timing_test.py
asks of ``pyLatticeDesign.timing.Timing`` (timing.py:16-288), plus the device rows and the hot-path decorators."""
import pickle
import time

from pylatticedso_amd.timing import Timing, timing


def test_counts_hierarchy_and_durations():
    t = Timing()
    assert t.call_stack == [] and t.timings is not None and hasattr(t, "local")

    @t.timeit
    def level3():
        time.sleep(0.002)

    @t.timeit
    def level2():
        level3()

    @t.timeit
    def level1():
        level2()
        level2()

    for _ in range(3):
        level1()
    n1, n2, n3 = ([n for n in t.timings if k in n][0] for k in ("level1", "level2", "level3"))
    assert t.call_counts[n1] == 3 and t.call_counts[n2] == 6 and t.call_counts[n3] == 6
    assert n2 in t.call_graph[n1] and n3 in t.call_graph[n2]
    assert all(d >= 0.002 for d in t.timings[n3]) and sum(t.timings[n1]) >= sum(t.timings[n2]) >= sum(t.timings[n3])
    assert t.call_stack == []
    t.reset()
    assert not t.timings and not t.call_graph and t._first_start is None


def test_names_categories_and_summary(capsys):
    t = Timing()

    class Solver:
        @t.category("simulation")
        @t.timeit
        def solve(self, n):
            t.device("PCG", 1.5)               # 1.5 ms measured on the GPU inside this call
            return n

        @t.timeit
        def idle(self):
            return None

    s = Solver()
    assert s.solve(3) == 3 and s.idle() is None
    assert "Solver.solve" in t.timings and t.func_category["Solver.solve"] == "simulation"
    assert abs(t.timings["device:PCG"][0] - 1.5e-3) < 1e-12 and "device:PCG" in t.call_graph["Solver.solve"]
    t.summary(name_width=60)
    out = capsys.readouterr().out
    assert "Solver.solve" in out and "└─ device:PCG" in out and "Total runtime" in out and "Calls" in out
    t.summary(classes=["Solver"], name_pattern="idle", group_by_category=True, top_n=5, min_total=0.0, max_depth=3)
    out = capsys.readouterr().out
    assert "Solver.idle" in out and "Solver.solve" not in out and "[uncategorized]" in out
    # exceptions still close the frame
    @t.timeit
    def boom():
        raise ValueError("x")
    try:
        boom()
    except ValueError:
        pass
    assert t.call_stack == [] and any("boom" in n for n in t.timings)
    t2 = pickle.loads(pickle.dumps(t))
    assert dict(t2.call_counts) == dict(t.call_counts)


def test_singleton_times_the_host_path():
    from pylatticedso_amd.lattice_sim import LatticeSim
    timing.reset()
    LatticeSim({"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 2, "y": 2, "z": 2},
                             "radii": [0.05], "geom_types": ["BCC"]},
                "simulation_parameters": {"enable": True, "material": "VeroClear"}})
    assert timing.call_counts["LatticeSim.set_penalized_beams"] == 1
    assert "LatticeSim.set_penalized_beams" in timing.call_graph["LatticeSim._generate_and_prepare"]
    assert timing.func_category["LatticeSim.define_angles_between_beams"] == "design"
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "src"))
    from pyLatticeDesign.timing import timing as shim     # the reference's import path (src/ layout)
    assert shim is timing
