"""Import shim: loads the directory ``neural-flow-style_amd/`` (not a valid Python
identifier) as the package ``neural_flow_style_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neural-flow-style_amd")
_spec = importlib.util.spec_from_file_location(
    "neural_flow_style_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["neural_flow_style_amd"] = _mod
_spec.loader.exec_module(_mod)
