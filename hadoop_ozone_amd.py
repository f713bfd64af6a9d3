"""Import shim: the package lives in the directory `hadoop-ozone_amd/` (not a valid Python
identifier), so `import hadoop_ozone_amd` lands here and is redirected to it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hadoop-ozone_amd")
_spec = importlib.util.spec_from_file_location(
    "hadoop_ozone_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hadoop_ozone_amd"] = _mod
_spec.loader.exec_module(_mod)
