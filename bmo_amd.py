"""Import shim: the package directory is `beamletoptics.jl_amd/` (a dot is not importable as-is)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "beamletoptics.jl_amd")
_spec = importlib.util.spec_from_file_location("bmo_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bmo_amd"] = _mod
_spec.loader.exec_module(_mod)
