"""Import alias: `import agcn_amd` loads the package that lives in
`adaptive-graph-convolutional-network_amd/` (a directory name Python cannot import directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adaptive-graph-convolutional-network_amd")
_spec = importlib.util.spec_from_file_location(
    "agcn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["agcn_amd"] = _mod
_spec.loader.exec_module(_mod)
