"""X.seg(duration, fn): reduce runs of tokens inside every sequence (the API of torchrua.segment,
reference segment.py:6-50).  `duration` is a container of run lengths (one sequence of runs per sequence
of `self`); `fn(rows, run_sizes) -> reduced` is normally one of torchrua_amd.reduce.segment_*, but any
callable with that contract works, which is what fixes the shape of the algorithm: `fn` wants ONE dense
row matrix and run sizes that cover it.

C : the runs of all sequences are already back to back — fn sees the payload as it is.
L/R: the padded storage [B, T] is handed to fn unchanged (no copy), with the padding of every row declared as
     one more run (trailing for L, leading for R); the column of results that belongs to it is cut off again.
     Its slots inside the result's own padding hold whatever fn makes of an empty run — the same values as in the
     reference for every segment_* but segment_last.
P : through C (two moves of the row mover), as the reference does.
"""
from torchrua_amd.layout import C, L, P, R, Z

__all__ = []  # methods are attached to the layout classes


def _seg_dense(self: C, duration: Z, fn) -> C:
    runs = duration.cat()
    return C(data=fn(self.data, runs.data), token_sizes=runs.token_sizes)


def _seg_padded(cls, pad_first: bool):
    """seg for a padded container; `pad_first` says on which side of a row its padding run sits."""

    def seg(self, duration: Z, fn):
        runs = duration.right(0) if pad_first else duration.left(0)        # [B, R] run lengths, zero-padded
        n_seq, n_steps = self.data.shape[:2]                                  # the PHYSICAL row grid fn will see
        n_runs = runs.data.size(1)
        # per row: R run lengths and the length of its padding, in storage order
        sizes = runs.data.new_empty((n_seq, n_runs + 1))
        pad_col, first_run = (0, 1) if pad_first else (n_runs, 0)
        sizes[:, first_run:first_run + n_runs] = runs.data
        sizes[:, pad_col] = n_steps - self.token_sizes
        reduced = fn(self.data.reshape((n_seq * n_steps,) + tuple(self.data.shape[2:])), sizes.reshape(-1))
        reduced = reduced.reshape((n_seq, n_runs + 1) + tuple(reduced.shape[1:]))
        return cls(data=reduced.narrow(1, first_run, n_runs), token_sizes=runs.token_sizes)

    return seg


def _seg_packed(self: P, duration: Z, fn) -> P:
    return _seg_dense(self.cat(), duration, fn).pack()


C.seg = _seg_dense
L.seg = _seg_padded(L, pad_first=False)
P.seg = _seg_packed
R.seg = _seg_padded(R, pad_first=True)
