"""Import shim: the package directory is `plonky2-lib_amd/` (not a valid Python identifier), so
`import plonky2_lib_amd` loads that directory as a package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plonky2-lib_amd")
_spec = importlib.util.spec_from_file_location(
    "plonky2_lib_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["plonky2_lib_amd"] = _mod
_spec.loader.exec_module(_mod)
