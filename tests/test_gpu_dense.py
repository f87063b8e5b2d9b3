"""GPU parity of the few-row dense kernel (csrc/dense_small.hip) that carries the decoder's per-step epsilon-rule
contractions  x * (W^T (r / z~))  (models/gridTDmodel.py:744-765 `lrp_linear_eps`), through the C ABI
(lrpx_conv_mfma, taps = 1), against plain fp32/fp64 PyTorch on the CPU."""
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import ops as o
    return o


@pytest.mark.parametrize("rows,k,n,n_src", [(320, 512, 1536, 16), (37, 96, 64, 5), (2048, 1024, 160, 64)])
def test_dense_rel_few_rows(ops, rows, k, n, n_src):
    """out[row][j] = X[src(row)][j] * sum_i A[row][i] W[i][j]   (PACK_DENSE_T: W is (k rows, n cols))"""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(rows + k)
    a = torch.randn(rows, k, generator=g)
    w = torch.randn(k, n, generator=g) * 0.05
    x = torch.randn(n_src, n, generator=g)
    src = torch.randint(0, n_src, (rows,), generator=g).to(torch.int32)
    n_pad = -(-n // 32) * 32
    wp = ops.pack_weights(w.cuda(), k, n, 1, _lib.PACK_DENSE_T, 32)
    out = torch.full((rows, n), float("nan"), device="cuda")
    ops.conv_mfma(a.cuda(), wp, rows, 0, k, n_pad, 1, _lib.EPI_REL, pix_per_map=1, oc_split=n, x=x.cuda(),
                  map2img=src.cuda(), out0=out)
    torch.cuda.synchronize()
    want = x[src.long()].double() * (a.double() @ w.double())
    assert rel_err(out.cpu(), want) < 2e-6


@pytest.mark.parametrize("rows,k,n,relu", [(320, 512, 1536, 0), (16, 64, 96, 1), (320, 2048, 1536, 0), (100, 1056, 64, 0)])
def test_dense_plain_few_rows(ops, rows, k, n, relu):
    """out = A W^T + b (PACK_DENSE: W is (n rows, k cols)), optional ReLU"""
    from lrp_amd import _lib
    g = torch.Generator().manual_seed(rows * 3 + k)
    a = torch.randn(rows, k, generator=g)
    w = torch.randn(n, k, generator=g) * 0.05
    b = torch.randn(n, generator=g)
    n_pad = -(-n // 32) * 32
    wp = ops.pack_weights(w.cuda(), n, k, 1, _lib.PACK_DENSE, 32)
    out = torch.full((rows, n), float("nan"), device="cuda")
    ops.conv_mfma(a.cuda(), wp, rows, 0, k, n_pad, 1, _lib.EPI_PLAIN, pix_per_map=1, oc_split=n, relu=relu,
                  bias=b.cuda(), out0=out)
    torch.cuda.synchronize()
    want = a.double() @ w.double().t() + b.double()
    if relu:
        want = want.clamp(min=0)
    assert rel_err(out.cpu(), want) < 2e-6
