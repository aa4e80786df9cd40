"""Import shim: the package directory is named `brotli-rs_amd/` (not a valid Python identifier), so this
module loads it under the importable name `brotli_rs_amd` and replaces itself with it in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "brotli-rs_amd")
_spec = importlib.util.spec_from_file_location("brotli_rs_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["brotli_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
