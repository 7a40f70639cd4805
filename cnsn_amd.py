"""Import alias: `import cnsn_amd` loads the package that lives in `crossnorm-selfnorm_amd/`
(a directory name Python cannot spell in an import statement)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "crossnorm-selfnorm_amd")
_spec = importlib.util.spec_from_file_location(
    "cnsn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["cnsn_amd"] = _pkg
_spec.loader.exec_module(_pkg)
