"""Import shim: the package directory is `gs-sdf_amd/` (hyphenated, as the build contract names
it), which Python cannot import by name.  `import gs_sdf_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gs-sdf_amd")
_spec = importlib.util.spec_from_file_location(
    "gs_sdf_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gs_sdf_amd"] = _mod
_spec.loader.exec_module(_mod)
