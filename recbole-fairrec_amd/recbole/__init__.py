"""`recbole` import alias: code and configs written against RecBole-FairRec's package path (`from recbole.quick_start
import run_recbole`, `from recbole.model.abstract_recommender import FairRecommender`, `recbole.trainer.Trainer`,
`recbole.utils.InputType`, ...) resolve to the MI355X-native implementation in `fairrec`, module for module
(the plugin surface is Python classes resolved by name).  Only what `fairrec` provides exists here;
anything else raises ImportError as usual.  Do not install next to the real RecBole.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import fairrec as _impl

__version__ = _impl.__version__
_PREFIX = __name__ + "."


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = "fairrec." + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except ModuleNotFoundError:
            return None
        return importlib.util.spec_from_loader(fullname, self)

    def create_module(self, spec):
        return importlib.import_module("fairrec." + spec.name[len(_PREFIX):])   # the SAME module object, second name

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
