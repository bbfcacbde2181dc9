"""Import shim: the package directory is `anofox-forecast_amd/` (not a valid identifier), so this
module makes it importable as `anofox_forecast_amd` by pointing __path__ at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "anofox-forecast_amd")]
__package__ = "anofox_forecast_amd"
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
