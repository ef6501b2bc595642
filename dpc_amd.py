"""Import alias: ``import dpc_amd`` == the package in
``differentiable-point-clouds_amd/`` (whose directory name, fixed by the
project layout, is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("differentiable-point-clouds_amd")
sys.modules[__name__] = _pkg
