"""CPU-side checks of the boundary: liblrpx.so loads, exports every symbol include/lrpx.h declares, and the
ctypes table (lrp_amd._lib.SIGNATURES) covers exactly those symbols.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

import lrp_amd  # noqa: F401
from lrp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "lrpx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lrpx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("liblrpx.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/lrpx.h but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == header_symbols()


def test_version_and_pure_host_entry_points():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("liblrpx.so not built")
    lib = _lib.load()
    assert lib.lrpx_version() >= 100
    assert lib.lrpx_conv_kc(224, 9, 8) == 8 and lib.lrpx_conv_kc(224, 9, 64) == 8 and lib.lrpx_conv_kc(112, 9, 128) == 8
    assert lib.lrpx_conv_kc(56, 9, 256) == 16 and lib.lrpx_conv_kc(0, 1, 512) == 32
    assert lib.lrpx_packed_floats(64, 64, 9, 16) == 64 * 64 * 9
    assert lib.lrpx_packed_floats(6, 64, 9, 16) == 32 * 64 * 9          # output channels pad to 32
    assert lib.lrpx_vgg16_trace_bytes(2) == 2 * lib.lrpx_vgg16_trace_bytes(1)
    # argument validation happens on the host, before any launch
    assert lib.lrpx_pack_weights(None, 1, 1, 9, 0, 16, None, None) == _lib.EINVAL
    assert b"null" in lib.lrpx_last_error_string()
    with pytest.raises(ValueError):
        _lib.check(_lib.EINVAL)
    with pytest.raises(AssertionError):
        _lib.check(_lib.EZERO)


def test_release_library_carries_no_experiment_switches():
    """wrong-result timing experiments must not be able to pose as the product."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("liblrpx.so not built")
    assert _lib.load().lrpx_build_flags() == b"", "this liblrpx.so is a timing-experiment / profiling build"


def _compile_flags_unit(tmp_path, name, flags):
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler on this machine")
    src = os.path.join(ROOT, "lrp-imagecaptioning-pytorch_amd", "csrc", "lrpx_build_flags.hip")
    out = str(tmp_path / name)
    r = subprocess.run(["g++", "-x", "c++", "-shared", "-fPIC", "-o", out, src] + flags, capture_output=True, text=True)
    return r, out


def test_experiment_switches_need_the_experiments_flag_and_show_up(tmp_path):
    """build_guard.h: -DLRPXH_EXP=.. (and the other wrong-result / profiling switches) alone does not compile; together with
    -DLRPX_EXPERIMENTS the library names them in lrpx_build_flags(), which the test above (and smoke()) refuse."""
    for sw in ("-DLRPXH_EXP=1", "-DLRPX_EPI_EXP=2", "-DLRPXB_EXP=1", "-DLRPXD_EXP=8", "-DLRPXH_END_SLEEP=4",
               "-DLRPXH_START_SKEW=2", "-DLRPX_STAMP"):
        r, _ = _compile_flags_unit(tmp_path, "refused.so", [sw])
        assert r.returncode != 0 and "LRPX_EXPERIMENTS" in r.stderr, sw
    r, so = _compile_flags_unit(tmp_path, "exp.so", ["-DLRPX_EXPERIMENTS", "-DLRPXH_EXP=33", "-DLRPX_EPI_EXP=2"])
    assert r.returncode == 0, r.stderr
    f = ctypes.CDLL(so).lrpx_build_flags
    f.restype = ctypes.c_char_p
    assert f() == b"LRPX_EXPERIMENTS LRPXH_EXP=33 LRPX_EPI_EXP=2"
    r, so = _compile_flags_unit(tmp_path, "rel.so", [])
    assert r.returncode == 0, r.stderr
    f = ctypes.CDLL(so).lrpx_build_flags
    f.restype = ctypes.c_char_p
    assert f() == b""


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblrpx.so")
    with pytest.raises(_lib.LrpxError):
        _lib.load()


def test_per_call_context_is_not_affected_by_other_threads_setters():
    """SURVEY §8(b) threading: the conv mode / forward switch travel per call (lrpx_vgg16_opts); the setters only move the
    process defaults.  Two threads with opts of their own resolve THEIR values 20 000 times each while a third thread keeps
    flipping the defaults; a thread without opts sees one of the defaults, never a torn value.  (Host logic: no device.)"""
    import ctypes as C
    import threading
    from lrp_amd import _lib
    lib = _lib.load()
    prev_mode, prev_fwd = lib.lrpx_set_conv_mode(-1), lib.lrpx_set_forward_f16(-1)
    stop = threading.Event()
    bad = []

    def flipper():
        k = 0
        while not stop.is_set():
            lib.lrpx_set_conv_mode(k % 4)
            lib.lrpx_set_forward_f16(k % 2)
            k += 1

    def pinned(mode, fwd):
        o = _lib.VggOpts(mode, fwd, None)
        m, f = C.c_int(-9), C.c_int(-9)
        for _ in range(20000):
            lib.lrpx_vgg16_resolve_opts(C.byref(o), C.byref(m), C.byref(f))
            if (m.value, f.value) != (mode, fwd):
                bad.append((mode, fwd, m.value, f.value))
                return

    def defaults():
        m, f = C.c_int(-9), C.c_int(-9)
        for _ in range(20000):
            lib.lrpx_vgg16_resolve_opts(None, C.byref(m), C.byref(f))
            if m.value not in (0, 1, 2, 3) or f.value not in (0, 1):
                bad.append(("default", m.value, f.value))
                return
    ts = [threading.Thread(target=pinned, args=(1, 0)), threading.Thread(target=pinned, args=(2, 1)),
          threading.Thread(target=defaults)]
    fl = threading.Thread(target=flipper)
    fl.start()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    fl.join()
    lib.lrpx_set_conv_mode(prev_mode)
    lib.lrpx_set_forward_f16(prev_fwd)
    assert not bad, bad[:3]
    # half-specified opts: the unspecified half comes from the defaults
    o = _lib.VggOpts(-1, 0, None)
    m, f = C.c_int(), C.c_int()
    lib.lrpx_vgg16_resolve_opts(C.byref(o), C.byref(m), C.byref(f))
    assert (m.value, f.value) == (prev_mode, 0)


def test_process_defaults_and_their_environment_overrides():
    """Round 6: the library's process default is conv mode 1 (exact bf16 splits: arithmetic no narrower than the reference's fp32,
    LRPtools/lrp_modules.py:124-150) with the exact forward trace; LRPX_CONV_MODE / LRPX_FORWARD_F16 move the INITIAL defaults (read once
    at load), out-of-range values are clamped.  Host logic only: one child process per environment."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import lrp_amd; from lrp_amd import _lib; lib = _lib.load(); "
            "print('DEFAULTS', lib.lrpx_set_conv_mode(-1), lib.lrpx_set_forward_f16(-1))" % root)
    for env, want in (({}, (1, 0)), ({"LRPX_CONV_MODE": "3"}, (3, 0)), ({"LRPX_CONV_MODE": "2", "LRPX_FORWARD_F16": "1"}, (2, 1)),
                      ({"LRPX_CONV_MODE": "9"}, (3, 0)), ({"LRPX_CONV_MODE": "0"}, (0, 0)), ({"LRPX_CONV_MODE": ""}, (1, 0))):
        e = {k: v for k, v in os.environ.items() if k not in ("LRPX_CONV_MODE", "LRPX_FORWARD_F16")}
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("DEFAULTS")][0].split()
        assert (int(line[1]), int(line[2])) == want, (env, line)
