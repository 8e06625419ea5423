"""Loader for the package directory ``toyfhe.jl_amd/`` (a dot is not importable as a module name).

``import toyfhe_jl_amd`` makes the package available under that name (with working relative imports)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "toyfhe.jl_amd")
_spec = importlib.util.spec_from_file_location("toyfhe_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["toyfhe_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
