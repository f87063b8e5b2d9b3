"""Import shim: `import lrp_amd` loads the package that lives in the directory
`lrp-imagecaptioning-pytorch_amd/` (the directory name is not a valid Python identifier,
so it is mounted under the importable name `lrp_amd`)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lrp-imagecaptioning-pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    "lrp_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lrp_amd"] = _mod
_spec.loader.exec_module(_mod)
