"""Importable alias for the product package.

The product package directory is named after the reference repository
(``super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd``),
which is not a valid Python identifier.  This shim makes it importable as
``surs_amd`` by pointing ``__path__`` at that directory; it holds no code of
its own.
"""
import os as _os

_PKG_DIR = _os.path.abspath(_os.path.join(
    _os.path.dirname(__file__), "..",
    "super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd"))
if not _os.path.isdir(_PKG_DIR):  # pragma: no cover
    raise ImportError("product package directory missing: " + _PKG_DIR)
__path__ = [_PKG_DIR]
PKG_DIR = _PKG_DIR

with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
