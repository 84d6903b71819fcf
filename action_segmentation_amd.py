"""Import shim: the package directory is named ``action-segmentation_amd`` (not a valid Python
identifier), so ``import action_segmentation_amd`` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "action-segmentation_amd")
_spec = importlib.util.spec_from_file_location(
    "action_segmentation_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["action_segmentation_amd"] = _mod
_spec.loader.exec_module(_mod)
