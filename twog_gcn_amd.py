"""Import shim: makes the package directory ``2g-gcn_amd/`` importable as ``twog_gcn_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), '2g-gcn_amd')
_spec = importlib.util.spec_from_file_location('twog_gcn_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['twog_gcn_amd'] = _mod
_spec.loader.exec_module(_mod)
