"""Import shim: the package directory is `open-universe_amd/` (not a valid Python identifier);
`import open_universe_amd` loads it under this importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "open-universe_amd")
_spec = importlib.util.spec_from_file_location(
    "open_universe_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["open_universe_amd"] = _mod
_spec.loader.exec_module(_mod)
