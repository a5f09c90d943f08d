"""CPU-only checks of the product's C ABI library: it loads, exports every symbol include/fdcm.h
declares, and its host-side exact-float code agrees with this machine's libm.  No GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import ROOT


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as g
    from openfdcm_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        g.build()
    return _capi


def test_library_exports_every_declared_symbol(capi):
    header = open(os.path.join(ROOT, "include", "fdcm.h")).read()
    declared = set(re.findall(r"\b(fdcm_[a-z_0-9]+)\s*\(", header))
    bound = {s[0] for s in capi.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    lib = C.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in capi.lib().fdcm_version()


def test_match_record_layout(capi):
    assert capi.MATCH_DTYPE.itemsize == 32
    assert capi.MATCH_DTYPE.fields["score"][1] == 4 and capi.MATCH_DTYPE.fields["transform"][1] == 8


def test_atanf_restatement_matches_libm_sampled(capi):
    # every 257th float bit pattern (16.7M values); the exhaustive sweep is test_atanf_exhaustive
    assert capi.lib().fdcm_selftest_atanf(0, 257, (1 << 32) // 257) == 0


@pytest.mark.slow
def test_atanf_exhaustive(capi):
    assert capi.lib().fdcm_selftest_atanf(0, 1, 1 << 32) == 0


def test_argument_errors_are_reported_not_thrown(capi):
    lib = capi.lib()
    out = C.c_void_p()
    rc = lib.fdcm_featuremap_build(None, 3, 30, 5.0, 1.0, 0, C.byref(out))
    assert rc == -1 and b"scene_lines" in lib.fdcm_last_error()
    scene = np.zeros((1, 4), dtype=np.float32)
    rc = lib.fdcm_featuremap_build(capi.fptr(scene), 1, 30, 5.0, 1.0, 7, C.byref(out))
    assert rc == -1 and b"distance" in lib.fdcm_last_error()
    rc = lib.fdcm_search_capacity(None, 1, 1, 1, None)
    assert rc == -1
    # round 3's entry points: the feature-map seam and the sharded engine's frames in flight
    one = np.zeros(2, dtype=np.float32)
    off = np.array([0, 1], dtype=np.int64)
    i64p = C.POINTER(C.c_int64)
    assert lib.fdcm_featuremap_minmax_translation(None, capi.fptr(scene), 1, capi.fptr(one), capi.fptr(one)) == -1
    assert b"featuremap" in lib.fdcm_last_error()
    assert lib.fdcm_featuremap_minmax_translation(None, capi.fptr(scene), -1, capi.fptr(one), capi.fptr(one)) == -1
    assert lib.fdcm_featuremap_evaluate(None, capi.fptr(scene), off.ctypes.data_as(i64p), 1, capi.fptr(one),
                                        off.ctypes.data_as(i64p), capi.fptr(one)) == -1
    t = C.c_int64()
    assert lib.fdcm_sharded_submit(None, capi.fptr(scene), 1, 4, 4, 1, 10, C.byref(t)) == -1
    assert lib.fdcm_sharded_wait(None, 0, C.byref(out), C.byref(t)) == -1
    assert lib.fdcm_sharded_set_frames_in_flight(None, 2) == -1
    d = C.c_int(-7)
    assert lib.fdcm_get_device(C.byref(d)) == 0 and d.value == 0  # the thread's default
    assert lib.fdcm_get_device(None) == -1


def test_lineio_roundtrip_and_assets(tmp_path):
    from openfdcm_amd import lineio
    from helpers import create_lines
    lines = create_lines(100, 10)
    p = str(tmp_path / "a.lines")
    lineio.write(p, lines)
    back = lineio.read(p)
    assert back.shape == (4, 100) and np.array_equal(back, lines)
    with pytest.raises(RuntimeError):
        lineio.read(str(tmp_path / "missing.lines"))


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "fdcm_example")
    libdir = os.path.join(ROOT, "openfdcm_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "fdcm_example.c"), "-o", exe, "-L", libdir, "-lfdcm_hip",
                           f"-Wl,-rpath,{libdir}", "-lm"])
    return exe


def test_header_is_plain_c_and_example_runs(capi, tmp_path):
    """include/fdcm.h compiles as pedantic C99 and a C program links against the library: the boundary
    is a C ABI, not a C++ one.  Without a GPU the example stops at the first device call."""
    import subprocess
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gfx950" in out.stdout and "lines with centre in [10, 18) of (20, 15): 2" in out.stdout
    assert "best penalised score 0.024516 (expected 0.024516)" in out.stdout


@pytest.mark.gpu
def test_c_example_on_device(capi, tmp_path):
    import subprocess
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "raw matches" in out.stdout and "#0 score" in out.stdout and "seam: multipliers of (1, 0) in [" in out.stdout, out.stdout
