"""Import shim: the package directory is named `weekend-raytracer-wgpu_amd` (as the project layout
prescribes), which is not a valid Python identifier.  `import weekend_raytracer_wgpu_amd` loads that
directory as a regular package under this importable name."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "weekend-raytracer-wgpu_amd"
_spec = importlib.util.spec_from_file_location(
    __name__, _pkg_dir / "__init__.py", submodule_search_locations=[str(_pkg_dir)])
_module = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _module
_spec.loader.exec_module(_module)
