"""Importable alias for the product package.

The product lives in the directory
``hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd/``
(the name the build contract asks for).  Hyphens cannot be imported, so this
stub re-points its ``__path__`` at that directory: ``import hupr_amd.models``
resolves to ``<that dir>/models/__init__.py`` and so on.
"""
import os as _os

PACKAGE_DIR = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
    "hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd",
)
__path__ = [PACKAGE_DIR]

# Execute the real package __init__ in this namespace (version string, etc.).
with open(_os.path.join(PACKAGE_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(PACKAGE_DIR, "__init__.py"), "exec"))
