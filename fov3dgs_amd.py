"""Import shim: the package directory is named ``fov-3dgs_amd`` (not a valid Python
identifier), so ``import fov3dgs_amd`` loads it from that directory under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fov-3dgs_amd")
_spec = importlib.util.spec_from_file_location(
    "fov3dgs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fov3dgs_amd"] = _mod
_spec.loader.exec_module(_mod)
