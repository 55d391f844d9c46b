"""Import alias: `import lsnf_amd` loads the package that lives in the directory
`latent-space-normalizing-flow_amd/` (whose name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "latent-space-normalizing-flow_amd")
_spec = importlib.util.spec_from_file_location("lsnf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lsnf_amd"] = _mod
_spec.loader.exec_module(_mod)
