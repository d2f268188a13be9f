"""Import shim: the package directory is named ``sliceslice-rs_amd`` (hyphen), which Python cannot
import by name.  ``import sliceslice_rs_amd`` loads that directory as a package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sliceslice-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "sliceslice_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sliceslice_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
