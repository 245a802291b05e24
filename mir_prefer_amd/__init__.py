"""Import shim: the package directory is `mir-prefer_amd/` (not a valid Python identifier), so this
module makes it importable as `mir_prefer_amd` by pointing its search path there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mir-prefer_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
