"""Import shim: ``import nerf_projects_amd`` resolves to the ``nerf-projects_amd/`` package.

The package directory carries the repository's hyphenated name, which Python
cannot import by identifier; this module replaces itself in ``sys.modules``
with a package object whose search path is that directory.
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nerf-projects_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
