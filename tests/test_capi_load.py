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


def test_sweep_mincols_switch_takes(capi):
    """FDCM_SWEEP_MINCOLS reaches the kernel's range count (csrc/fdcm_sweep.h: sweep_ranges is the one function the
    kernel and this self test call).  Default: 16 columns per range at least; the fuzz variants' 2 and 1 give a slice of 16
    seeded columns all 8 ranges."""
    import subprocess
    import sys
    assert capi.lib().fdcm_selftest_sweep_ranges(16) == 1 and capi.lib().fdcm_selftest_sweep_ranges(127) == 7
    assert capi.lib().fdcm_selftest_sweep_ranges(829) == 8 and capi.lib().fdcm_selftest_sweep_ranges(0) == 1
    code = ("import sys; sys.path.insert(0, %r); from openfdcm_amd import _capi; "
            "print(_capi.lib().fdcm_selftest_sweep_ranges(16), _capi.lib().fdcm_selftest_sweep_ranges(5))" % ROOT)
    for val, want in (("2", "8 2"), ("1", "8 5"), ("64", "1 1")):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "FDCM_SWEEP_MINCOLS": val})
        assert out.returncode == 0 and out.stdout.split("\n")[0].strip() == want, (val, out.stdout, out.stderr[-500:])


def test_sweep_descriptor_prefetch_is_untouched_until_its_wait():
    """local_run (csrc/fdcm_sweep.hip) issues the next column's s_load_dwordx4 from inline assembly without a
    wait; the compiler does not know the SGPRs are in flight.  tools/check_sweep_prefetch.py reads the built library's
    gfx950 code: nothing names those registers before an s_waitcnt that covers lgkmcnt(0)."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_sweep_prefetch.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "sweep prefetch check ok" in out.stdout and "k_sweep_balanced" in out.stdout, out.stdout + out.stderr[-1000:]


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


def test_lines_read_write_through_the_c_abi(capi, tmp_path):
    """fdcm_lines_read / _write (serialization.h:99-132) against the reference's shipped files and the Python reader:
    the same lines from the same bytes, files written through the ABI read by the Python reader and the other way
    round, and the reference's error messages."""
    import ctypes as C
    from openfdcm_amd import lineio
    lib = capi.lib()

    def c_read(path):
        p, n = C.POINTER(C.c_float)(), C.c_int64()
        rc = lib.fdcm_lines_read(path.encode(), C.byref(p), C.byref(n))
        if rc != 0:
            return rc, lib.fdcm_last_error().decode()
        a = np.ctypeslib.as_array(p, shape=(n.value, 4)).copy() if n.value else np.zeros((0, 4), np.float32)
        lib.fdcm_lines_free(p)
        return 0, np.ascontiguousarray(a.T)

    gold = os.path.join(ROOT, "tests", "golden", "obj_04")
    for name in ("scene_0.scene", "template_0.tmpl", "template_57.tmpl", "template_121.tmpl"):
        rc, got = c_read(os.path.join(gold, name))
        want = lineio.read(os.path.join(gold, name))
        assert rc == 0 and got.shape == want.shape and got.tobytes() == want.tobytes(), name
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 1000):
        lines = rng.uniform(-500, 500, size=(4, n)).astype(np.float32)
        p1, p2 = str(tmp_path / f"c_{n}.lines"), str(tmp_path / f"py_{n}.lines")
        rec = np.ascontiguousarray(lines.T)
        assert lib.fdcm_lines_write(p1.encode(), capi.fptr(rec), n) == 0, lib.fdcm_last_error()
        assert lineio.read(p1).tobytes() == lines.tobytes()           # ABI writer -> Python reader
        lineio.write(p2, lines)
        rc, back = c_read(p2)                                          # Python writer -> ABI reader
        assert rc == 0 and back.tobytes() == lines.tobytes()
        assert lib.fdcm_lines_write(p1.encode(), capi.fptr(rec), n) == 0  # an existing file is replaced
    rc, msg = c_read(str(tmp_path / "missing.lines"))
    assert rc == -1 and "does not exist" in msg
    junk = tmp_path / "junk.lines"
    junk.write_bytes(b"not a line file at all, but long enough to hold a header of 39 bytes")
    rc, msg = c_read(str(junk))
    assert rc == -1 and "not an OPENFDCM line file" in msg
    assert lib.fdcm_lines_read(None, None, None) == -1
    # a well-formed file with another record length is "unsupported", not "truncated" (the reference does not check
    # lineDataRecordLen, serialization.h:114-131, but only reads 16-byte lines correctly)
    import zlib
    blob = bytearray(open(str(tmp_path / "c_7.lines"), "rb").read())  # outer header: flag @22, sizes @23 / @31, payload @39
    body = bytearray(zlib.decompress(bytes(blob[39:])) if blob[22] else blob[39:])
    body[35:37] = (24).to_bytes(2, "little")                          # lineDataRecordLen
    comp = zlib.compress(bytes(body)) if blob[22] else bytes(body)
    head = bytearray(blob[:39])
    head[23:31] = len(body).to_bytes(8, "little")
    head[31:39] = len(comp).to_bytes(8, "little")
    odd = tmp_path / "odd.lines"
    odd.write_bytes(bytes(head) + comp)
    rc, msg = c_read(str(odd))
    assert rc == -1 and "unsupported line record length <24>" in msg, msg


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


@pytest.mark.gpu
def test_c_example_runs_config_1_from_the_shipped_assets(capi, tmp_path):
    """`fdcm_example tests/golden/obj_04`: the reference's scene and its 122 templates read with fdcm_lines_read, config 1 of
    BASELINE.json from plain C; the match count and the three best matches are the Python path's (which the parity suite
    holds to the oracle)."""
    import glob
    import subprocess
    from openfdcm_amd import lineio, _capi
    from openfdcm_amd.engine import DeviceFeatureMap, DeviceTemplates, search_raw, topk
    gold = os.path.join(ROOT, "tests", "golden", "obj_04")
    exe = _build_c_example(tmp_path)
    out = subprocess.run([exe, gold], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout + out.stderr
    scene = lineio.read(os.path.join(gold, "scene_0.scene"))
    tmpls = [lineio.read(os.path.join(gold, f"template_{i}.tmpl")) for i in range(len(glob.glob(os.path.join(gold, "*.tmpl"))))]
    fm = DeviceFeatureMap.build(scene, depth=30, coeff=5.0, padding=1.0, distance=0)
    tset = DeviceTemplates(tmpls)
    raw = search_raw(fm, tset, scene, 4, 4, _capi.BATCH_OPTIMIZE, 10)
    best = topk(fm, tset, 3, _capi.EXPONENTIAL_PENALTY, 1.5)
    assert f"assets: {scene.shape[1]} scene lines, {len(tmpls)} templates" in out.stdout, out.stdout
    assert f"x 30, {len(raw)} raw matches" in out.stdout, out.stdout
    for i in range(3):
        assert f"best {i}: template {int(best[i]['tmpl_idx'])} score {float(best[i]['score']):.9g}" in out.stdout, out.stdout
