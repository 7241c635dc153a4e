"""Shared test plumbing: golden fixtures, oracle <-> torch conversion.  (tests may use oracle/.)"""
import json
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

if not os.path.exists(os.path.join(ROOT, 'oracle', 'librua_oracle.so')):
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)

from oracle import rua_oracle as orc  # noqa: E402

_small = None


def golden():
    """{case: {field: ndarray}} from tests/golden/small.npz (outputs of the reference itself)."""
    global _small
    if _small is None:
        out = {}
        for fname in ('small.npz', 'extra.npz', 'foreign.npz', 'r3.npz', 'r4.npz', 'r5.npz'):   # oracle/gen_golden.py: main / extra / foreign / round3 / round4 / round5
            z = np.load(os.path.join(GOLDEN, fname))
            for key in z.files:
                case, name = key.split('/', 1)
                out.setdefault(case, {})[name] = z[key]
        _small = out
    return _small


def fill_of(fields):
    """The fill value a `layout.*` fixture was padded with (oracle/gen_golden.py: layout_case): -1.5 for floating
    payloads, -7 for signed integers, and — round 5 — what the case stored for uint8 (7) and bool (True)."""
    dt = fields['data'].dtype
    if dt.kind == 'i':
        return -7
    if dt.kind == 'u' and dt.itemsize == 1:
        return int(fields['fill'])
    if dt.kind == 'b':
        return bool(fields['fill'])
    return float(fields['fill'])


def cotangent(shape, salt=0) -> np.ndarray:
    """The fixed cotangent of the gradient fixtures (oracle/gen_golden.py: cot_like), rebuilt from the shape."""
    n = int(np.prod(shape, dtype=np.int64))
    a = ((np.arange(n, dtype=np.float64) + 1.0 + salt) * 0.6180339887498949) % 1.0 - 0.5
    return a.astype(np.float32).reshape(shape)


def golden_sha():
    with open(os.path.join(GOLDEN, 'sha.json')) as f:
        return json.load(f)


def cases(prefix):
    return sorted(c for c in golden() if c.startswith(prefix))


def seq_from(fields, name, kind):
    """Rebuild an oracle Seq from golden fields '<name>.data' etc."""
    if kind == 'P':
        return orc.P(fields[f'{name}.data'], fields[f'{name}.batch_sizes'], fields[f'{name}.sorted_indices'],
                     fields[f'{name}.unsorted_indices'])
    return orc.Seq(kind, fields[f'{name}.data'], token_sizes=fields[f'{name}.token_sizes'])


def assert_seq_equal(actual, fields, name, kind, exact=True, rtol=0.0, atol=0.0):
    exp = seq_from(fields, name, kind)
    assert actual.kind == kind
    _cmp(actual.data, exp.data, exact, rtol, atol, f'{name}.data')
    for f in ('token_sizes', 'batch_sizes', 'sorted_indices', 'unsorted_indices'):
        e = getattr(exp, f)
        if e is not None:
            a = getattr(actual, f)
            assert a is not None, f'{name}.{f} missing'
            np.testing.assert_array_equal(np.asarray(a), e, err_msg=f'{name}.{f}')


def _cmp(a, e, exact, rtol, atol, what):
    a, e = np.asarray(a), np.asarray(e)
    assert a.shape == e.shape, f'{what}: shape {a.shape} vs {e.shape}'
    assert a.dtype == e.dtype, f'{what}: dtype {a.dtype} vs {e.dtype}'
    if exact:
        assert a.tobytes() == e.tobytes() or np.array_equal(a, e, equal_nan=True), f'{what}: not bit-exact'
    else:
        np.testing.assert_allclose(a, e, rtol=rtol, atol=atol, equal_nan=True, err_msg=what)


# ---- torch <-> numpy (bf16 travels as its uint16 bit pattern)
def to_np(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu().contiguous()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def to_torch(a: np.ndarray, device, bf16=False) -> torch.Tensor:
    if bf16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).to(device)
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def sha256(a) -> str:
    import hashlib
    if isinstance(a, torch.Tensor):
        a = to_np(a)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def seeded_inputs(seed, B, lo, hi, H, dtype):
    """SURVEY.md §8(d) input recipe (must match oracle/gen_golden.py: sha_configs.inputs)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (B,), generator=g)
    data = torch.randn(int(lens.sum()), H, generator=g).to(dtype)
    return lens, data
