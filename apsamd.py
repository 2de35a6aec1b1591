"""Import shim: the package directory name required by the build contract contains hyphens
(``automaticpanoramicimagestitching-autopanostitch-matlab_amd``), which the ``import`` statement
cannot spell.  ``import apsamd`` gives the same module object."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
PACKAGE_NAME = "automaticpanoramicimagestitching-autopanostitch-matlab_amd"
_pkg = importlib.import_module(PACKAGE_NAME)
sys.modules[__name__] = _pkg
