"""Randomised parity sweep (tools/fuzz_conv.py, fixed seed, short budget): every tile configuration x odd shapes,
channel windows, partial K chunks, epilogue options, fused expand and weight gradients against torch CPU fp32."""
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_conv_family_randomised_sweep(capsys):
    argv = sys.argv
    sys.argv = ["fuzz_conv.py", "12", "20261003"]
    try:
        cwd = os.getcwd()
        os.chdir(ROOT)
        try:
            runpy.run_path(os.path.join(ROOT, "tools", "fuzz_conv.py"), run_name="__main__")
        except SystemExit as e:                                   # the script exits 1 on the first mismatch
            assert not e.code, capsys.readouterr().out
        finally:
            os.chdir(cwd)
    finally:
        sys.argv = argv
    assert "fuzz ok" in capsys.readouterr().out


@pytest.mark.gpu
def test_stem_family_randomised_sweep(capsys, monkeypatch):
    """tools/fuzz_stem.py, fixed seed, short budget: the wave stem kernels (values bit for bit against the workgroup kernel, arg-max
    codes), the gather stem weight gradient (against the dense kernel on the same codes and, flip-aware, against autograd), stem +
    first squeeze, squeeze + expand1x1, on random image sizes / channel windows."""
    for k in ('SQD_STEM_WAVE', 'SQD_STEM_WGRAD_GATHER'):          # the script switches kernels through the environment: restore it afterwards
        monkeypatch.setenv(k, os.environ.get(k, '2' if k == 'SQD_STEM_WAVE' else '1'))
    argv = sys.argv
    sys.argv = ["fuzz_stem.py", "12", "20261004"]
    try:
        cwd = os.getcwd()
        os.chdir(ROOT)
        try:
            runpy.run_path(os.path.join(ROOT, "tools", "fuzz_stem.py"), run_name="__main__")
        except SystemExit as e:
            assert not e.code, capsys.readouterr().out
        finally:
            os.chdir(cwd)
    finally:
        sys.argv = argv
    assert "cases ok" in capsys.readouterr().out


@pytest.mark.gpu
def test_storing_bridges_randomised_sweep(capsys):
    """tools/fuzz_bridge.py, fixed seed, short budget: the storing forms of the fused launches of the training forward (Fire -> Fire
    bridge, Fire -> pool -> Fire bridge with pooled tensor + arg-max / ReLU codes, stem + squeeze) bit for bit against the launches
    they replace, and the in-place refresh of the bridges' operands against a fresh packing, on random shapes / windows / segments."""
    argv = sys.argv
    sys.argv = ["fuzz_bridge.py", "12", "20261005"]
    try:
        cwd = os.getcwd()
        os.chdir(ROOT)
        try:
            runpy.run_path(os.path.join(ROOT, "tools", "fuzz_bridge.py"), run_name="__main__")
        except SystemExit as e:
            assert not e.code, capsys.readouterr().out
        finally:
            os.chdir(cwd)
    finally:
        sys.argv = argv
    assert "OK {" in capsys.readouterr().out
