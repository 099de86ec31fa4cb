"""CPU-side checks of the boundary: the shared library loads and exports every symbol the
header declares; no compute calls are made (there is no GPU in this container)."""
import ctypes as C
import os
import re

from sift_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sift_texture.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.EXPORTS)


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof as a C compiler sees include/sift_texture.h against the ctypes mirrors."""
    import subprocess
    structs = {"stx_params": _lib.StxParams, "stx_metrics": _lib.StxMetrics, "stx_timing": _lib.StxTiming,
               "stx_extended": _lib.StxExtended}
    lines = []
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sift_texture.h"\nint main(void) {\n'
                   + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_version_and_defaults():
    lib = _lib.load()
    assert lib.stx_version().startswith(b"sift-texture")
    p = _lib.StxParams()
    lib.stx_default_params(C.byref(p))
    # the literals of /root/reference/app/passive_checker.py:103, :116
    assert (p.lbp_points, p.lbp_radius, p.canny_low, p.canny_high, p.gray_shift, p.fft_h, p.fft_w) == (
        8, 1.0, 50, 150, 15, 0, 0)


def test_score_summary_matches_numpy_mean():
    import numpy as np
    lib = _lib.load()
    rng = np.random.default_rng(1)
    for n in (0, 1, 3, 7, 8, 9, 10, 17, 129, 300):
        vals = rng.random(n)
        arr = (_lib.StxMetrics * max(n, 1))()
        for i, v in enumerate(vals):
            arr[i].texture_score = float(v)
        avg = C.c_double()
        passed = C.c_int()
        assert lib.stx_score_summary(arr, n, 0.6, C.byref(avg), C.byref(passed)) == 0
        want = float(np.mean(vals)) if n else 0.0
        assert avg.value == want
        assert passed.value == int(want >= 0.6)
