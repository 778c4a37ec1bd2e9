"""Import shim: makes ``import ratilqr.jl_amd`` resolve to the package directory ``ratilqr.jl_amd/``
at the repo root (a directory name with a dot cannot be imported by the normal machinery)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ratilqr.jl_amd")
if "ratilqr.jl_amd" not in _sys.modules:
    _spec = _ilu.spec_from_file_location(
        "ratilqr.jl_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
    )
    jl_amd = _ilu.module_from_spec(_spec)
    _sys.modules["ratilqr.jl_amd"] = jl_amd
    _spec.loader.exec_module(jl_amd)
else:
    jl_amd = _sys.modules["ratilqr.jl_amd"]
