"""Import alias: the product package lives in the directory `self-forcing_amd/` (the
name the project layout prescribes), which is not a valid Python identifier.  This
module makes it importable as `self_forcing_amd` by pointing `__path__` at that
directory and executing its `__init__.py` in this namespace."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "self-forcing_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
