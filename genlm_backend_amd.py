"""Import shim: the package directory is named `genlm-backend_amd` (not a legal Python identifier),
so `import genlm_backend_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "genlm-backend_amd")
_spec = importlib.util.spec_from_file_location(
    "genlm_backend_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["genlm_backend_amd"] = _mod
_spec.loader.exec_module(_mod)
