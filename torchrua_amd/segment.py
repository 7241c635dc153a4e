"""X.seg(duration, fn): reduce runs of tokens inside every sequence — mirror of torchrua.segment
(reference segment.py:6-50).  `fn` is any (tensor, segment_sizes) -> tensor callable, normally one of
torchrua_amd.reduce.segment_*; the composition around it follows the reference so custom `fn`s
keep working, with the conversions done by the row mover."""
import torch

from torchrua_amd.layout import C, L, P, R, Z


__all__ = []  # methods are attached to the layout classes


def _cat_seg(self: C, duration: Z, fn) -> C:
    """segment.py:6-10."""
    duration = duration.cat()
    return duration._replace(data=fn(self.data, duration.data))


def _left_seg(self: L, duration: Z, fn) -> L:
    """segment.py:16-25: the padding run of every row is one extra trailing segment, dropped afterwards."""
    duration = duration.left(0)
    b, t, *sizes = self.size()
    token_sizes = torch.cat([duration.data, t - self.token_sizes[:, None]], dim=-1).view(-1)
    data = fn(self.data.flatten(start_dim=0, end_dim=1), token_sizes).view((b, -1, *sizes))
    return L(data=data[:, :-1], token_sizes=duration.token_sizes)


def _pack_seg(self: P, duration: Z, fn) -> P:
    """segment.py:31-32."""
    return self.cat().seg(duration, fn).pack()


def _right_seg(self: R, duration: Z, fn) -> R:
    """segment.py:38-47: the padding run is one extra leading segment."""
    duration = duration.right(0)
    b, t, *sizes = self.size()
    token_sizes = torch.cat([t - self.token_sizes[:, None], duration.data], dim=-1).view(-1)
    data = fn(self.data.flatten(start_dim=0, end_dim=1), token_sizes).view((b, -1, *sizes))
    return R(data=data[:, +1:], token_sizes=duration.token_sizes)


C.seg = _cat_seg
L.seg = _left_seg
P.seg = _pack_seg
R.seg = _right_seg
