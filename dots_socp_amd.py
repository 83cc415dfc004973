"""Import alias: the package directory is ``dots-socp_amd/`` (not a valid Python
identifier), so ``import dots_socp_amd`` resolves here and this module turns itself
into that package by pointing ``__path__`` at the directory and running its
``__init__.py``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "dots-socp_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r", encoding="utf-8") as _fh:
    exec(compile(_fh.read(), __file__, "exec"))
del _fh
