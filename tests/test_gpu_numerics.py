"""SnakeBeta with the generator's own sin^2 routine (a hand-written range reduction, csrc/k_vocoder.hip) against
float64, far beyond the arguments the synthetic checkpoints produce: alpha up to +6 (exp = 403), |x| up to 20,
i.e. |x * exp(alpha)| up to 8000 (activations.py:107-120)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("alpha,beta", [(-2.0, 0.3), (0.0, 0.0), (1.5, -1.0), (3.0, 0.5), (4.5, -0.5), (6.0, 0.0), (6.0, -2.0)])
def test_snakebeta_large_arguments(alpha, beta):
    from bvcodec import _abi
    lib = _abi.load()
    g = torch.Generator().manual_seed(int(alpha * 10 + beta * 100 + 1000))
    x = torch.cat([torch.linspace(-20, 20, 200001), 20 * (2 * torch.rand(200000, generator=g) - 1),
                   torch.tensor([0.0, -0.0, 1e-30, 20.0, -20.0])]).float()
    y = torch.full_like(x, float("nan")).to(DEV)
    xd = x.to(DEV)
    _abi.check(lib.bvc_test_snakebeta(_abi.ptr(xd), x.numel(), float(alpha), float(beta), _abi.ptr(y),
                                      _abi.current_stream(DEV)))
    torch.cuda.synchronize()
    a32 = np.float32(np.exp(np.float64(alpha)))                       # the product's constants, as it derives them
    ib32 = np.float32(1.0) / (np.float32(np.exp(np.float64(beta))) + np.float32(1e-9))
    arg = (x.numpy() * a32).astype(np.float32)                       # the reference multiplies in float32 first
    ref = x.numpy().astype(np.float64) + np.float64(ib32) * np.sin(arg.astype(np.float64)) ** 2
    got = y.cpu().numpy().astype(np.float64)
    assert not np.isnan(got).any()
    err = np.abs(got - ref)
    tol = 2.5e-7 * float(ib32) + 1.2e-7 * np.maximum(1.0, np.abs(ref))          # sin^2 error scaled by 1/(e^beta) + output rounding
    worst = int(np.argmax(err - tol))
    assert (err <= tol).all(), (alpha, beta, float(x[worst]), float(arg[worst]), float(err[worst]), float(tol[worst]))
