"""Import shim: the package sources live in ``gsm-vi_amd/`` (not a valid Python identifier), so
``import gsmvi_amd`` redirects there.  Nothing else lives in this directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gsm-vi_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
