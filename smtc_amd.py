"""Import alias: the package directory is named `socialmedia-textimage-classification-auxlosses_amd`
(not a Python identifier); `import smtc_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "socialmedia-textimage-classification-auxlosses_amd")
_spec = importlib.util.spec_from_file_location("smtc_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["smtc_amd"] = _mod
_spec.loader.exec_module(_mod)
