"""Import shim: the product package lives in the directory `rust-raytracing_amd/` (a name Python's
import statement cannot spell); `import rust_raytracing_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rust-raytracing_amd")
_spec = importlib.util.spec_from_file_location(
    "rust_raytracing_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rust_raytracing_amd"] = _mod
_spec.loader.exec_module(_mod)
