"""Import shim: the package lives in the directory ``graphem-rapids_amd/`` (not a valid
Python identifier); ``import graphem_rapids_amd`` from the repository root loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graphem-rapids_amd")
_spec = importlib.util.spec_from_file_location(
    "graphem_rapids_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["graphem_rapids_amd"] = _mod
_spec.loader.exec_module(_mod)
