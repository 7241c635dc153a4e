"""torchrua_amd — MI355X-native ragged-sequence operators behind the torchrua API.

Same names, fields and call signatures as speedcell4/torchrua 0.5.1 for its hot path (layout
conversion cat/pack/left/right, select head/last/roll/rev/trunc, segmented and scatter reduce); every
index computation and payload move is a hand-written gfx950 HIP kernel reached through the C ABI in
include/rua.h.  There is no CPU or eager fallback: tensors must live on a HIP device and
librua_hip.so must have been built (`make -C torchrua_amd/csrc`).
"""
from torchrua_amd.layout import *  # noqa: F401,F403
from torchrua_amd.utils import *  # noqa: F401,F403
from torchrua_amd.core import *  # noqa: F401,F403
from torchrua_amd.core import patch_tensor_indexing, unpatch_tensor_indexing, with_host_sizes  # noqa: F401
from torchrua_amd.select import *  # noqa: F401,F403
from torchrua_amd.reduce import *  # noqa: F401,F403
from torchrua_amd.segment import *  # noqa: F401,F403
from torchrua_amd.mask import *  # noqa: F401,F403
from torchrua_amd.compose import *  # noqa: F401,F403
from torchrua_amd.detach import *  # noqa: F401,F403
from torchrua_amd._lib import RuaError, load as load_library  # noqa: F401

# BASELINE.json's names for the constructors (README.md:13 of the reference speaks of them too)
cat_sequence = C.new
pack_sequence = P.new
pad_sequence = L.new
PaddedSequence = LeftAlignedSequence

__version__ = '0.1.0'


def install_as_torchrua() -> None:
    """Make `import torchrua` resolve to this package (drop-in under code written for the reference).

    Like importing the reference, this also teaches `tensor[Z]` / `tensor[Z] = value` to take a container of row
    indices (the reference patches Tensor.__getitem__/__setitem__ at import time: core/get.py:11-18,
    core/set.py:10-18).  A plain `import torchrua_amd` leaves torch.Tensor untouched; call
    `patch_tensor_indexing()` yourself to opt in without the alias."""
    _namespace.alias_as('torchrua')           # torchrua, torchrua.core.cast, torchrua.select.roll, ...
    patch_tensor_indexing()


from torchrua_amd import _namespace  # noqa: E402

_namespace.build(__import__('sys').modules[__name__])
