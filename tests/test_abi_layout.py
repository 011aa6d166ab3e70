"""The three descriptions of the C ABI's structs must agree: include/pylattice_hip.h (truth), the product's ctypes
binding (pylatticedso_amd/_capi.py) and the binding INTEGRATION.md shows a reference maintainer.  Round 2's snippet had
drifted 7 fields behind the header, which overruns the caller's struct in pl_default_opts; since then the structs carry
their own size (ABI handshake) and this test pins field names, order, types, offsets and sizeof.  CPU only."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HEADER = os.path.join(ROOT, "include", "pylattice_hip.h")

CTYPE = {"uint32_t": C.c_uint32, "int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double,
         "const double *": C.c_void_p, "const int32_t *": C.c_void_p}


def header_struct(name):
    """[(field, ctypes type)] of `typedef struct { ... } name;` parsed from the header."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct \{([^}]*)\}\s*" + name + r"\s*;", src).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(const \w+ \*|\w+)\s*(.*)", decl)
        ctype, names = m.group(1), m.group(2)
        for nm in names.split(","):
            nm = nm.strip()
            arr = re.match(r"(\w+)\[(\d+)\]", nm)
            out.append((arr.group(1), CTYPE[ctype] * int(arr.group(2))) if arr else (nm.lstrip("*").strip(), CTYPE[ctype]))
    return out


def same_fields(a, b):
    def norm(t):
        return (getattr(t, "_type_", t), getattr(t, "_length_", 0), C.sizeof(t))
    return [(n, norm(t)) for n, t in a] == [(n, norm(t)) for n, t in b]


def test_capi_structs_follow_the_header():
    from pylatticedso_amd import _capi
    for cname, cls in (("pl_opts_t", _capi.PlOpts), ("pl_stats_t", _capi.PlStats), ("pl_mesh_t", _capi.PlMesh),
                       ("pl_lattice_info_t", _capi.PlLatticeInfo)):
        hdr = header_struct(cname)
        assert same_fields(hdr, cls._fields_), (cname, [n for n, _ in hdr], [n for n, _ in cls._fields_])


def test_layout_matches_the_c_compiler(tmp_path):
    """gcc's sizeof / offsetof of every field against the ctypes classes."""
    from pylatticedso_amd import _capi
    structs = {"pl_opts_t": _capi.PlOpts, "pl_stats_t": _capi.PlStats, "pl_mesh_t": _capi.PlMesh,
               "pl_lattice_info_t": _capi.PlLatticeInfo}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_integration_md_binding_follows_the_header():
    """The ctypes classes in INTEGRATION.md's sketch are executed and compared with the header."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md[md.index("class _Mesh(C.Structure)"):md.index("# ABI handshake")]
    ns = {"C": C}
    exec(block, ns)
    for cname, key in (("pl_mesh_t", "_Mesh"), ("pl_opts_t", "_Opts"), ("pl_stats_t", "_Stats")):
        hdr = header_struct(cname)
        assert same_fields(hdr, ns[key]._fields_), (cname, [n for n, _ in hdr], [n for n, _ in ns[key]._fields_])
    assert "pl_default_opts(C.byref(opts), C.sizeof(_Opts))" in md
    assert "st.struct_size = C.sizeof(_Stats)" in md


def test_library_refuses_a_stale_binding():
    """An old caller (smaller pl_opts_t, no stamp) gets PL_ERR_ARG and its memory is left alone."""
    from pylatticedso_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        pytest.skip("libpylattice_hip.so not built")
    lib = _capi.load_library()
    assert lib.pl_opts_size() == C.sizeof(_capi.PlOpts) and lib.pl_stats_size() == C.sizeof(_capi.PlStats)
    m = re.search(r"#define PL_ABI_VERSION (\d+)u", open(HEADER).read())
    assert lib.pl_abi_version() == int(m.group(1))

    class OldOpts(C.Structure):        # round-2 INTEGRATION.md layout: ends at grid_nodes, no size stamp
        _fields_ = _capi.PlOpts._fields_[2:21]
    buf = (C.c_uint8 * (C.sizeof(_capi.PlOpts) + 64))(*([0xAB] * (C.sizeof(_capi.PlOpts) + 64)))
    old = OldOpts.from_buffer(buf)
    assert lib.pl_default_opts(C.byref(old), C.sizeof(OldOpts)) == _capi.PL_ERR_ARG
    assert all(b == 0xAB for b in buf)                       # nothing written, nothing overrun
    assert b"pl_opts_t" in lib.pl_last_error()
    # a struct of the right size that never went through pl_default_opts is refused by pl_create as well
    raw = _capi.PlOpts()
    mesh = _capi.PlMesh()
    h = C.c_void_p()
    assert lib.pl_create(C.byref(mesh), C.byref(raw), C.byref(h)) == _capi.PL_ERR_ARG
    assert b"pl_default_opts" in lib.pl_last_error()
    ok = _capi.default_opts(lib)
    assert ok.struct_size == C.sizeof(_capi.PlOpts) and ok.abi_version == lib.pl_abi_version()
    assert ok.young == 1013.0 and ok.kappa == 0.9 and ok.pen_coef == 1.5 and ok.reorder == 1
