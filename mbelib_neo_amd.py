"""Import shim: the package directory is named ``mbelib-neo_amd`` (hyphen, as the project
layout prescribes), which is not a valid Python identifier.  ``import mbelib_neo_amd`` loads
that directory as the package ``mbelib_neo_amd``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mbelib-neo_amd")
_spec = _ilu.spec_from_file_location(
    "mbelib_neo_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["mbelib_neo_amd"] = _mod
_spec.loader.exec_module(_mod)
