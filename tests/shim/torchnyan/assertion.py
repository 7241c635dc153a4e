"""assert_close / assert_grad_close / assert_sequence_close as the reference's tests call them."""
import torch
from torch.nn.utils.rnn import PackedSequence


def assert_close(actual, expected, rtol=1e-4, atol=1e-5, **kwargs):
    kwargs.setdefault('check_stride', False)
    torch.testing.assert_close(actual, expected, rtol=rtol, atol=atol, **kwargs)


def assert_grad_close(actual, expected, inputs, rtol=1e-4, atol=1e-5):
    """Backpropagate one shared random cotangent through both results and compare the input gradients."""
    inputs = list(inputs)
    cotangent = torch.randn_like(expected)
    got = torch.autograd.grad(actual, inputs, cotangent, retain_graph=True, allow_unused=True)
    want = torch.autograd.grad(expected, inputs, cotangent, retain_graph=True, allow_unused=True)
    for g, w, x in zip(got, want, inputs):
        g = torch.zeros_like(x) if g is None else g
        w = torch.zeros_like(x) if w is None else w
        torch.testing.assert_close(g, w, rtol=rtol, atol=atol)


def assert_sequence_close(actual, expected, rtol=1e-4, atol=1e-5):
    """Same container type; payload close; every integer field equal."""
    assert type(actual) is type(expected), f'{type(actual).__name__} vs {type(expected).__name__}'
    if isinstance(expected, PackedSequence):
        fields = ('batch_sizes', 'sorted_indices', 'unsorted_indices')
    else:
        fields = ('token_sizes',)
    torch.testing.assert_close(actual.data, expected.data, rtol=rtol, atol=atol, check_stride=False)
    for name in fields:
        a, e = getattr(actual, name), getattr(expected, name)
        assert (a is None) == (e is None), name
        if e is not None:
            assert torch.equal(a.cpu(), e.cpu()), name
