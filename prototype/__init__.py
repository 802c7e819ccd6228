"""`import prototype...` keeps working for code written against the reference layout: every `prototype.X` import is
served by the MI355X implementation module `ilvlm_amd.prototype.X` (same module object under both names)."""
import importlib
import importlib.abc
import importlib.util
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == "prototype" or name.startswith("prototype."):
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("ilvlm_amd." + spec.name)

    def exec_module(self, module):
        pass


_sys.meta_path.insert(0, _Alias())
_impl = importlib.import_module("ilvlm_amd.prototype")
_sys.modules[__name__] = _impl
