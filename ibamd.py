"""Import alias for the package directory ``immersedboundary.jl_amd/``.

The directory name contains a dot, so it cannot be named in an ``import``
statement.  ``import ibamd`` loads this file, which registers the directory as
the package ``ibamd`` (sub-modules resolve through its ``__path__``).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "immersedboundary.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "ibamd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ibamd"] = _mod
_spec.loader.exec_module(_mod)
