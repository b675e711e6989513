"""Import shim: the package directory is `spasm.jl_amd/` (a dot is not importable), so this module
loads it under the name `spasm_jl_amd`.  `import spasm_jl_amd` then yields the package itself."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spasm.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "spasm_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["spasm_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
