"""Wall-clock / call-graph collector with the decorator API of the reference's ``pyLatticeDesign.timing``
(timing.py:16-288: ``Timing.timeit``, ``Timing.category``, ``reset``, ``summary`` and the module singleton ``timing``),
so that user scripts which decorate their own functions or print ``timing.summary()`` keep working.

Host time is ``time.perf_counter`` around the decorated call, as in the reference.  The C-ABI wrappers
(``_capi.HipLattice``) are decorated with it too, and they additionally report what the DEVICE spent inside a call -
HIP-event times the library measures on its own stream (``pl_stats_t.ms_assembly`` / ``ms_solve``) - through
``Timing.device``: those show up as ``device:<kernel group>`` rows, children of the host call that produced them.
"""
from __future__ import annotations

import re
import threading
import time
from collections import defaultdict
from functools import wraps


def _float_dict():
    return defaultdict(float)      # module-level so that a Timing object pickles


class Timing:
    def __init__(self):
        self.timings = defaultdict(list)            # qualified name -> durations [s]
        self.call_counts = defaultdict(int)
        self.call_graph = defaultdict(_float_dict)  # parent -> {child: seconds spent in child below parent}
        self.call_stack = []
        self.func_category = {}
        self.local = threading.local()
        self._first_start = None
        self._last_end = None

    # pickling (the reference pickles lattices that hold a reference to the collector)
    def __getstate__(self):
        d = dict(self.__dict__)
        d["local"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.local = threading.local()

    # ---- naming --------------------------------------------------------------------------------------------
    @staticmethod
    def _qualified_name(func, args):
        """``Class.method`` when the first positional argument is an instance or a class, else ``module:qualname``
        (timing.py:51-73 - any first argument counts as 'self', which is what gives free functions called with
        arguments names like ``int.f``; kept, since summaries are matched by these names)."""
        fname = getattr(func, "__name__", None)
        if args and fname is not None:
            return f"{type(args[0]).__name__}.{fname}"
        mod = getattr(func, "__module__", None) or "<unknown>"
        return f"{mod}:{getattr(func, '__qualname__', None) or fname or '<unnamed>'}"

    # ---- decorators ----------------------------------------------------------------------------------------
    def category(self, label):
        def tag(func):
            func._timing_category = label
            return func
        return tag

    def timeit(self, func):
        @wraps(func)
        def timed(*args, **kwargs):
            name = self._qualified_name(func, args)
            # @category above @timeit (the order the reference writes) tags the wrapper, below it the function
            label = getattr(timed, "_timing_category", None) or getattr(func, "_timing_category", None)
            if label is not None:
                self.func_category[name] = label
            parent = self.call_stack[-1] if self.call_stack else None
            self.call_stack.append(name)
            t0 = time.perf_counter()
            if self._first_start is None:
                self._first_start = t0
            try:
                return func(*args, **kwargs)
            finally:
                t1 = time.perf_counter()
                self._last_end = t1
                self._record(name, t1 - t0, parent)
                self.call_stack.pop()
        return timed

    def _record(self, name, seconds, parent):
        self.timings[name].append(seconds)
        self.call_counts[name] += 1
        if parent:
            self.call_graph[parent][name] += seconds

    def device(self, name, milliseconds, category="device"):
        """A duration measured ON THE GPU (HIP events inside libpylattice_hip), booked as ``device:<name>`` below the
        host call that is running."""
        qn = f"device:{name}"
        self.func_category[qn] = category
        self._record(qn, float(milliseconds) * 1e-3, self.call_stack[-1] if self.call_stack else None)

    def reset(self):
        self.timings = defaultdict(list)
        self.call_counts = defaultdict(int)
        self.call_graph = defaultdict(_float_dict)
        self.call_stack = []
        self._first_start = self._last_end = None

    # ---- report --------------------------------------------------------------------------------------------
    def summary(self, classes=None, name_pattern=None, max_depth=None, min_total=0.0, top_n=None,
                print_children=True, name_width=40, group_by_category=False):
        """Aligned table: one row per function (calls, total, average, maximum), children indented below their parent;
        filters as in the reference (timing.py:128-166)."""
        keep_cls = set(classes) if classes else None
        rx = re.compile(name_pattern) if name_pattern else None

        def wanted(name):
            if keep_cls is not None and not any(name.startswith(c + ".") for c in keep_cls):
                return False
            return rx is None or rx.search(name) is not None

        parents_of = defaultdict(list)
        for par, kids in self.call_graph.items():
            for kid in kids:
                parents_of[kid].append(par)
        depth = {}

        def depth_of(name, seen=()):
            if name not in depth:
                ps = [p for p in parents_of.get(name, ()) if p not in seen]
                depth[name] = 0 if not ps else 1 + min(depth_of(p, seen + (name,)) for p in ps)
            return depth[name]

        rows = [(n, sum(t)) for n, t in self.timings.items() if wanted(n)]
        if max_depth is not None:
            rows = [r for r in rows if depth_of(r[0]) <= max_depth]
        rows = sorted((r for r in rows if r[1] >= float(min_total)), key=lambda r: -r[1])
        if top_n is not None:
            rows = rows[:top_n]

        def clip(text, width):
            return text if len(text) <= width else text[:max(0, width - 1)] + "…"

        print(f"{'Function':<{name_width}} {'Calls':>10} {'Total (s)':>12} {'Avg (s)':>12} {'Max (s)':>12}")
        print("-" * (name_width + 50))

        def emit(name, total):
            t = self.timings[name]
            print(f"{clip(name, name_width):<{name_width}} {len(t):>10} {total:>12.6f} {total / len(t):>12.6f} "
                  f"{max(t):>12.6f}")
            if not print_children:
                return
            for kid, spent in sorted(self.call_graph.get(name, {}).items(), key=lambda kv: -kv[1]):
                if wanted(kid) and (max_depth is None or depth_of(kid) <= max_depth):
                    label = "└─ " + clip(kid, max(0, name_width - 3))
                    print(f"{label:<{name_width}} {self.call_counts[kid]:>10} {spent:>12.6f}")

        if group_by_category:
            groups = defaultdict(list)
            for name, total in rows:
                groups[self.func_category.get(name, "uncategorized")].append((name, total))
            order = sorted((g for g in groups if g != "uncategorized"), key=lambda g: -sum(t for _, t in groups[g]))
            if "uncategorized" in groups:
                order.append("uncategorized")
            for g in order:
                print(f"\n[{g}]")
                for name, total in groups[g]:
                    emit(name, total)
        else:
            for name, total in rows:
                emit(name, total)
        if self._first_start is not None and self._last_end is not None:
            print(f"\nTotal runtime: {self._last_end - self._first_start:.4f} s")
        else:
            print("\nTotal runtime: n/a")


timing = Timing()
