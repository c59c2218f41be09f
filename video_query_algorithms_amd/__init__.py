"""Importable alias of the package directory ``video-query-algorithms_amd/`` (a hyphen is not a
valid module name).  ``import video_query_algorithms_amd`` executes that directory's __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "video-query-algorithms_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
