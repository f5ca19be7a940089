"""Import shim: ``import dgsct_amd`` loads the package that lives in ./dg-sct_amd (a directory name
that is not a Python identifier) under the importable name ``dgsct_amd``."""
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
_dir = os.path.join(_root, "dg-sct_amd")
_spec = importlib.util.spec_from_file_location("dgsct_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["dgsct_amd"] = _pkg
_spec.loader.exec_module(_pkg)
