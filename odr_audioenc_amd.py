"""Import shim: the package directory is `odr-audioenc_amd/` (the name the project layout asks for),
which is not a valid Python identifier; this module makes it importable as `odr_audioenc_amd`."""
import os as _os

__package__ = __name__
__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "odr-audioenc_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f, _os
