"""A stand-in for the third-party `torchnyan` test helper the reference's tests import (it is
not vendored in the reference and not installed here).  Provides exactly the ten names those tests use, with
the semantics inferred from their call sites, so that the reference's test files can run UNMODIFIED against
either implementation — see scripts/run_reference_tests.py.  Written for this repository; not reference code."""
import torch
from hypothesis import strategies as st

from torchnyan.assertion import assert_close, assert_grad_close, assert_sequence_close  # noqa: F401

BATCH_SIZE = 24
TOKEN_SIZE = 50
FEATURE_DIM = 40
TINY_BATCH_SIZE = 5
TINY_TOKEN_SIZE = 11

device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')


def sizes(*maxes: int):
    """sizes(a) -> an int in [1, a];  sizes(a, b) -> a list (1..a long) of ints in [1, b];  and so on."""
    if len(maxes) == 1:
        return st.integers(min_value=1, max_value=maxes[0])
    return st.lists(sizes(*maxes[1:]), min_size=1, max_size=maxes[0])
