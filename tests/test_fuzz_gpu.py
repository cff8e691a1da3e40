"""Randomised parity sweep (scratch/fuzz_conv.py, fixed seed, short budget): every tile configuration x odd shapes,
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
            runpy.run_path(os.path.join(ROOT, "scratch", "fuzz_conv.py"), run_name="__main__")
        except SystemExit as e:                                   # the script exits 1 on the first mismatch
            assert not e.code, capsys.readouterr().out
        finally:
            os.chdir(cwd)
    finally:
        sys.argv = argv
    assert "fuzz ok" in capsys.readouterr().out
