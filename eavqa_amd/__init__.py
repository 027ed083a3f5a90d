"""Import name for the product package.

The package directory is named ``explicit-alignment-for-vqa-tasks_amd`` (hyphens are not valid in a
Python identifier), so this thin module points ``eavqa_amd`` at it: ``import eavqa_amd.models.clipcap``
loads ``explicit-alignment-for-vqa-tasks_amd/models/clipcap.py`` under the one canonical name
``eavqa_amd.models.clipcap``.
"""
import os as _os

_PKG = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "explicit-alignment-for-vqa-tasks_amd")
__path__.insert(0, _PKG)

with open(_os.path.join(_PKG, "_version.py")) as _f:
    exec(_f.read())
