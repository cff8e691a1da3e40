"""Import alias: ``import squeezedet_pytorch_amd`` loads the package that lives in the
directory ``squeezedet-pytorch_amd/`` (a hyphen cannot appear in a Python module name)."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "squeezedet-pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    "squeezedet_pytorch_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["squeezedet_pytorch_amd"] = _mod
_spec.loader.exec_module(_mod)
