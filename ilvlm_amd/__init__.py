"""Importable alias of the package directory `iterated-learning-for-vlm_amd/` (a hyphenated name
cannot be imported directly).  All code lives there; this file only redirects the package path."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                 "iterated-learning-for-vlm_amd"))
from ._pkg import *  # noqa: F401,F403,E402
