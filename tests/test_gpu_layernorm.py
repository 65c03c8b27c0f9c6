"""msst_layernorm_fwd / _bwd (row a7 as an op of its own, SURVEY.md 8b export list) against the oracle's layer_norm
(reference nn.LayerNorm: vit_spatial_spectral.py:25 -- D = 96 -- and :194-195 -- D = 10 and D = 96), through the C-ABI."""
import pytest
import torch

from conftest import seed_all
from util import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,D", [(1, 96), (7, 96), (32, 96), (1280 * 4 + 5, 96), (327680, 96), (1, 10), (13, 10), (5120 * 3 + 1, 10), (9, 64), (3, 128), (17, 33)])
def test_layernorm_matches_oracle(rows, D):
    from maskedsst_amd.ops import layer_norm
    from oracle.model import layer_norm as oracle_ln
    seed_all(11)
    x = torch.randn(rows, D) * 1.7 + 0.3
    w = torch.randn(D) * 0.5 + 1.0
    b = torch.randn(D) * 0.2
    dy = torch.randn(rows, D)
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    yo = oracle_ln(xo, wo, bo)
    yo.backward(dy)
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    yd = layer_norm(xd, wd, bd)
    yd.backward(dy.cuda())
    torch.cuda.synchronize()
    assert relerr(yd, yo) <= 1e-5, relerr(yd, yo)
    assert relerr(xd.grad, xo.grad) <= 2e-5, relerr(xd.grad, xo.grad)
    # d gamma / d beta sum over all rows in fp32 (slab per workgroup, fixed order): tolerance grows with sqrt(rows)
    tol = 2e-5 if rows < 10000 else 2e-4
    assert relerr(wd.grad, wo.grad) <= tol, relerr(wd.grad, wo.grad)
    assert relerr(bd.grad, bo.grad) <= tol, relerr(bd.grad, bo.grad)


def test_layernorm_is_bit_reproducible_and_shaped():
    from maskedsst_amd.ops import layer_norm
    seed_all(3)
    x = torch.randn(4, 20, 64, 96).cuda().requires_grad_(True)
    w = torch.ones(96).cuda().requires_grad_(True)
    b = torch.zeros(96).cuda().requires_grad_(True)
    outs = []
    for _ in range(2):
        for t in (x, w, b):
            t.grad = None
        y = layer_norm(x, w, b)
        assert y.shape == x.shape
        y.square().sum().backward()
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()))
    for a, c in zip(*outs):
        assert torch.equal(a, c)


def test_layernorm_refuses_cpu_tensors():
    from maskedsst_amd.ops import layer_norm
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer_norm(torch.randn(4, 96), torch.ones(96), torch.zeros(96))
