"""Importable alias of the package directory `semantic-superpoint_amd/` (a hyphen cannot be imported):
`import semantic_superpoint_amd as ssp` executes semantic-superpoint_amd/__init__.py with this module's
__path__ pointing at that directory, so sub-modules resolve there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "semantic-superpoint_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
