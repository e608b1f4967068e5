"""Import shim: the package directory is named `wgpu-renderer_amd`, which is not a valid
Python identifier; `import wgpu_renderer_amd` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wgpu-renderer_amd")
_spec = importlib.util.spec_from_file_location(
    "wgpu_renderer_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["wgpu_renderer_amd"] = _mod
_spec.loader.exec_module(_mod)
