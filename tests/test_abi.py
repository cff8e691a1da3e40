"""CPU tier: the C-ABI library builds for gfx950, loads, exports every symbol include/sqd_hip.h
declares, and the ctypes signatures agree with the header (argument counts and kinds).
No compute call is made (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_protos():
    txt = open(os.path.join(ROOT, "include", "sqd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(sqd_\w+)\s*\(([^)]*)\)\s*;", txt):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        protos[m.group(1)] = args
    return protos


def _kind(carg):
    if "*" in carg:
        return "ptr"
    if carg.startswith("float"):
        return "float"
    if carg.startswith("long long"):
        return "longlong"
    return "int"


def test_library_builds_and_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    from squeezedet_pytorch_amd import _native as nat
    lib = nat.lib()
    protos = _header_protos()
    assert len(protos) >= 14
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/sqd_hip.h but not exported"


def test_ctypes_signatures_match_header():
    from squeezedet_pytorch_amd import _native as nat
    protos = _header_protos()
    sigs = dict(nat._SIGNATURES)
    sigs.update({k: v for k, v in nat._OPTIONAL.items() if k in protos})
    for name, cargs in protos.items():
        assert name in sigs, f"{name} has no ctypes signature"
        at = sigs[name]
        assert len(at) == len(cargs), f"{name}: header has {len(cargs)} args, ctypes {len(at)}"
        for c, t in zip(cargs, at):
            k = _kind(c)
            if k == "ptr":
                assert t is ctypes.c_void_p or issubclass(t, ctypes._Pointer), (name, c, t)
            elif k == "float":
                assert t is ctypes.c_float, (name, c, t)
            elif k == "longlong":
                assert t is ctypes.c_longlong, (name, c, t)
            else:
                assert t is ctypes.c_int, (name, c, t)


def test_status_codes_without_gpu():
    """Argument validation happens on the host before any launch, so it can be exercised here."""
    from squeezedet_pytorch_amd import _native as nat
    lib = nat.lib()
    n = lib.sqd_conv_num_cfgs()
    assert n >= 10
    t, k, px, bn = (ctypes.c_int() for _ in range(4))
    assert lib.sqd_conv_cfg_info(0, ctypes.byref(t), ctypes.byref(k), ctypes.byref(px), ctypes.byref(bn)) == 0
    assert t.value in (1, 9) and k.value in (16, 32, 64) and px.value in (64, 128, 256) and bn.value % 16 == 0
    assert lib.sqd_conv_cfg_is_dma(0) == 0 and lib.sqd_conv_cfg_is_dma(n) == -1
    assert lib.sqd_conv_cfg_info(n, None, None, None, None) == 1           # bad cfg id
    null = ctypes.c_void_p(0)
    assert lib.sqd_conv_fwd(null, null, null, null, null, null, null, 1, 1, 1, 4, 4, 0, 4, 16, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, null) == 1
    assert lib.sqd_detect_fwd(null, null, null, null, null, null, null, null, null, 1, 1, 3, 1, 1, 64, 0.4, 0.3, null) == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from squeezedet_pytorch_amd import _native as nat
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(nat.NativeLibraryError):
        nat.lib()
