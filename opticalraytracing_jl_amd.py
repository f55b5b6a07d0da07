"""Import shim: the package directory is `opticalraytracing.jl_amd/` (a dot is not legal in a
Python package name), so this module adopts that directory as its package path.
`import opticalraytracing_jl_amd as ort` and `from opticalraytracing_jl_amd import api` work."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "opticalraytracing.jl_amd")]
__package__ = __name__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f, _os
