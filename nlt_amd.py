"""Import alias: `import nlt_amd` loads the package that lives in `neural-light-transport_amd/`
(a directory name Python cannot import directly because of the hyphens)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'neural-light-transport_amd')
_spec = importlib.util.spec_from_file_location(
    'nlt_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['nlt_amd'] = _mod
_spec.loader.exec_module(_mod)
