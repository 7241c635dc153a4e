"""The reference's module-level names and submodule paths, as aliases of what this package attaches to the layouts.

`torchrua/__init__.py:1-8` star-imports every helper, so user code written for the reference may say
`from torchrua import to_pack, cat_roll, pack_head, cat_pack_to_left, new_cat, tensor_getitem` or
`from torchrua.select.roll import cat_roll`.  Here the operators live as methods on C / L / P / R (one kernel-backed
function often serves several layouts); this module hands them out under the reference's free-function names and
builds the reference's submodule tree (`core.cast`, `core.get`, `core.set`, `core.view`, `layout.{cat,left,pack,right}`,
`select.{head,last,rev,roll,trunc}`) as namespace modules over the same objects.  Nothing is re-implemented here.
`tests/golden/names.json` (the reference's own `dir()` of every module, a committed fixture) is what
`tests/test_namespace.py` holds this table to.
"""
import sys
import types
from numbers import Number
from typing import Any, List, NamedTuple, Tuple, Union

import torch
from torch import Tensor

from importlib import import_module as _im

# (import_module, not `from torchrua_amd import x`: the package attribute `compose` is the FUNCTION, as in the reference)
_compose_mod, _core, _detach, _layout, _mask_mod, _reduce, _segment, _select, _utils = (
    _im(f'torchrua_amd.{m}') for m in ('compose', 'core', 'detach', 'layout', 'mask', 'reduce', 'segment', 'select', 'utils'))
from torchrua_amd.layout import C, L, P, R, T, Z, CattedSequence, LeftAlignedSequence, RightAlignedSequence, PackedSequence

_PKG = 'torchrua_amd'
Key = _core.Key
Value = Union[Tensor, Z]


def _fn(cls, name):
    """The plain function behind a method attached to a layout class."""
    f = cls.__dict__[name]
    return f.__func__ if isinstance(f, staticmethod) else f


_TYPES = dict(C=C, L=L, P=P, R=R, T=T, Z=Z, Tensor=Tensor, Tuple=Tuple, Union=Union, torch=torch)

# reference module -> {name: object}; typing names ride along so that `dir()` matches the reference's
CAST = dict(to_self=_core._to_self, to_cat=_core._to_cat, to_pack=_core._to_pack,
            cat_pack_to_left=_fn(C, 'left'), cat_pack_to_right=_fn(C, 'right'),
            right_to_left=_fn(R, 'left'), left_to_right=_fn(L, 'right'), Number=Number)
GET = dict(cat_getitem=_fn(C, '__getitem__'), left_getitem=_fn(L, '__getitem__'), pack_getitem=_fn(P, '__getitem__'),
           right_getitem=_fn(R, '__getitem__'), tensor_getitem=_core.tensor_getitem, Key=Key, Value=Value)
SET = dict(cat_setitem=_fn(C, '__setitem__'), left_setitem=_fn(L, '__setitem__'), pack_setitem=_fn(P, '__setitem__'),
           right_setitem=_fn(R, '__setitem__'), tensor_setitem=_core.tensor_setitem, Key=Key)
VIEW = dict(get_mask=_core.get_mask, cat_view=_core._cat_view, left_view=_fn(C, 'left_view'),
            pack_view=_core._pack_view, right_view=_fn(C, 'right_view'), to_self=_core._to_self,
            invert_permutation=_utils.invert_permutation, Number=Number)
NEW = dict(new_cat=_fn(C, 'new'), new_left=_fn(L, 'new'), new_pack=_fn(P, 'new'), new_right=_fn(R, 'new'), Any=Any,
           List=List)
HEAD = dict(cat_head=_fn(C, 'head'), left_head=_fn(L, 'head'), pack_head=_fn(P, 'head'), right_head=_fn(R, 'head'))
LAST = dict(last=_fn(C, 'last'))
REV = dict(cat_rev=_fn(C, 'rev'), left_rev=_fn(L, 'rev'), pack_rev=_fn(P, 'rev'), right_rev=_fn(R, 'rev'))
ROLL = dict(cat_roll=_fn(C, 'roll'), left_roll=_fn(L, 'roll'), pack_roll=_fn(P, 'roll'), right_roll=_fn(R, 'roll'))
TRUNC = dict(cat_trunc=_fn(C, 'trunc'), left_trunc=_fn(L, 'trunc'), pack_trunc=_fn(P, 'trunc'),
             right_trunc=_fn(R, 'trunc'), major_sizes_to_ptr=_utils.major_sizes_to_ptr)
SEG = dict(cat_seg=_fn(C, 'seg'), left_seg=_fn(L, 'seg'), pack_seg=_fn(P, 'seg'), right_seg=_fn(R, 'seg'))
MASK = dict(mask=_fn(C, 'mask'), bmask=_fn(C, 'bmask'), fmask=_fn(C, 'fmask'))
DETACH = dict(cat_pack_split=_fn(C, 'split'), left_split=_fn(L, 'split'), right_split=_fn(R, 'split'),
              tolist=_fn(C, 'tolist'), List=List, Number=Number)
# layout/pack.py defines size / ptr / idx / offsets / raw as free functions of a PackedSequence and
# layout/__init__.py star-exports them (layout/pack.py:12-55)
PACK = dict(size=_fn(P, 'size'), ptr=_fn(P, 'ptr'), idx=_fn(P, 'idx'), offsets=_fn(P, 'offsets'), raw=_fn(P, 'raw'),
            PackedSequence=PackedSequence, get_offsets=_utils.get_offsets, major_sizes_to_ptr=_utils.major_sizes_to_ptr)
LAY_CAT = dict(CattedSequence=CattedSequence, NamedTuple=NamedTuple, get_offsets=_utils.get_offsets,
               major_sizes_to_ptr=_utils.major_sizes_to_ptr)
LAY_LEFT = dict(LeftAlignedSequence=LeftAlignedSequence, NamedTuple=NamedTuple,
                major_sizes_to_ptr=_utils.major_sizes_to_ptr)
LAY_RIGHT = dict(RightAlignedSequence=RightAlignedSequence, NamedTuple=NamedTuple,
                 major_sizes_to_ptr=_utils.major_sizes_to_ptr)
UTILS = dict(to_self=_core._to_self, Any=Any, List=List)


def _module(path: str, *tables) -> types.ModuleType:
    """A namespace module `torchrua_amd.<path>` over the given name tables (registered so that `import` finds it)."""
    name = f'{_PKG}.{path}'
    mod = sys.modules.get(name)
    if mod is None:
        mod = types.ModuleType(name, f'Namespace twin of the reference\'s torchrua.{path} (see torchrua_amd/_namespace.py).')
        sys.modules[name] = mod
    _fill(mod, _TYPES, *tables)
    return mod


def _fill(mod, *tables) -> None:
    for table in tables:
        for k, v in table.items():
            if table is _TYPES and k in mod.__dict__:       # typing names never displace what a module already holds
                continue
            setattr(mod, k, v)


def build(pkg) -> None:
    """Called once at the end of torchrua_amd/__init__.py."""
    # core: a real module here (core.py) that gains the reference's four submodules and their names
    cast, get, set_, view = (_module('core.cast', CAST), _module('core.get', GET), _module('core.set', SET),
                             _module('core.view', VIEW))
    _fill(_core, CAST, GET, SET, VIEW, NEW, dict(cast=cast, get=get, set=set_, view=view, Value=Value))
    # layout
    lay = dict(cat=_module('layout.cat', LAY_CAT), left=_module('layout.left', LAY_LEFT),
               pack=_module('layout.pack', PACK), right=_module('layout.right', LAY_RIGHT))
    _fill(_layout, PACK, LAY_CAT, lay)
    # select
    sel = dict(head=_module('select.head', HEAD), rev=_module('select.rev', REV), roll=_module('select.roll', ROLL),
               trunc=_module('select.trunc', TRUNC))
    _module('select.last', LAST)              # (`select.last` the ATTRIBUTE is the function: select/__init__.py star-imports it)
    _fill(_select, HEAD, LAST, REV, ROLL, TRUNC, sel)
    _fill(_segment, SEG)
    _fill(_mask_mod, MASK)
    _fill(_detach, DETACH)
    _fill(_utils, UTILS)
    _fill(_compose_mod, dict(invert_permutation=_utils.invert_permutation, List=List))
    for mod in (_compose_mod, _core, _detach, _layout, _mask_mod, _reduce, _segment, _select, _utils):
        _fill(mod, _TYPES)
    # the package itself: every public name of every submodule (torchrua/__init__.py:1-8), then the submodules
    for table in (CAST, GET, SET, VIEW, NEW, HEAD, LAST, REV, ROLL, TRUNC, SEG, MASK, DETACH, PACK, LAY_CAT,
                  dict(NamedTuple=NamedTuple, Number=Number, Any=Any, List=List, Tensor=Tensor, Tuple=Tuple, Union=Union,
                       torch=torch, to_self=_core._to_self),
                  dict(cast=cast, get=get, set=set_, view=view), lay, sel):
        for k, v in table.items():
            setattr(pkg, k, v)


class _AliasFinder:
    """Meta-path finder that answers `import <alias>.x.y` with the module `torchrua_amd.x.y` ITSELF.  Without it a
    submodule that was not yet imported when alias_as() ran (`import torchrua.parallel`) would be found through the
    package's shared __path__ and executed a second time under the alias: two module objects, two sets of classes and
    autograd Functions, two copies of every piece of module state."""

    def __init__(self, alias: str):
        self.alias = alias

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.alias and not fullname.startswith(self.alias + '.'):
            return None
        import importlib
        import importlib.util
        real = _PKG + fullname[len(self.alias):]
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name == real:          # no such module under torchrua_amd either: let the import fail as usual
                return None
            raise
        own_spec = mod.__spec__

        class _Loader:
            def create_module(self, spec):
                return mod

            def exec_module(self, module):      # already executed under its own name; keep its own spec
                module.__spec__ = own_spec

        spec = importlib.util.spec_from_loader(fullname, _Loader(), origin=getattr(mod, '__file__', None),
                                               is_package=hasattr(mod, '__path__'))
        return spec


def alias_as(name: str) -> None:
    """Every `torchrua_amd[.x.y]` module also answers to `<name>[.x.y]` in sys.modules (install_as_torchrua): the ones
    already imported are aliased at once, the rest (`torchrua.parallel`, private modules) when somebody imports them."""
    for key, mod in list(sys.modules.items()):
        if key == _PKG or key.startswith(_PKG + '.'):
            tail = key[len(_PKG):]
            if tail.startswith('._'):          # private modules: aliased lazily, by the finder
                continue
            sys.modules.setdefault(name + tail, mod)
    if not any(isinstance(f, _AliasFinder) and f.alias == name for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder(name))
