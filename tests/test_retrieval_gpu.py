"""GPU: RICES retrieval kernels against float64 numpy / torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from eavqa_amd import ops as o
    return o


def ref_topk(scores, k):
    """stable sort: descending value, ties by smaller column"""
    s = scores.double().cpu().numpy()
    idx = np.argsort(-s, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(s, idx, 1), idx


@pytest.mark.parametrize("rows,cols,k", [(5, 3000, 2048), (3, 2048, 2048), (7, 100000, 2048), (4, 1000, 7), (2, 70000, 1), (3, 333, 300)])
def test_topk_rows_exact(ops, rows, cols, k):
    g = torch.Generator().manual_seed(rows * 1000 + k)
    x = torch.randn(rows, cols, generator=g)
    x[0, : cols // 2] = x[0, 0]                  # a long run of exact ties straddling the threshold
    if rows > 1:
        x[1] = torch.round(x[1] * 4) / 4         # heavy ties everywhere, negative values and zeros of both signs
        x[1, 5] = -0.0
    val, idx = ops.topk_rows(x.to(DEV), k)
    wv, wi = ref_topk(x, k)
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.array_equal(val.cpu().numpy().astype(np.float64), wv)
    v2, i2 = ops.topk_rows(x.to(DEV), k)       # bitwise reproducible
    assert torch.equal(i2, idx) and torch.equal(v2, val)


def test_topk_rows_rejects(ops):
    from eavqa_amd._lib import EavqaError
    x = torch.zeros(2, 100, device=DEV)
    with pytest.raises(EavqaError):
        ops.topk_rows(x, 101)
    with pytest.raises(EavqaError):
        ops.topk_rows(torch.zeros(2, 5000, device=DEV), 4096)


def test_l2_normalize_rows(ops):
    x = torch.randn(9, 768, generator=torch.Generator().manual_seed(1))
    x[3] = 0.0
    y = ops.l2_normalize_rows_(x.clone().to(DEV)).cpu()
    want = x / x.norm(dim=1, keepdim=True).clamp(min=1e-30)
    want[3] = 0.0
    assert (y - want).abs().max().item() <= 1e-6
    assert torch.equal(y[3], torch.zeros(768))


def test_knn_inner_product_matches_float64(ops):
    from eavqa_amd.utils import rices
    g = torch.Generator().manual_seed(3)
    db, q = torch.randn(5000, 768, generator=g), torch.randn(300, 768, generator=g)
    D, I = rices.knn_inner_product(db.to(DEV), q.to(DEV), k=256, query_tile=128)
    dn, qn = db.double() / db.double().norm(dim=1, keepdim=True), q.double() / q.double().norm(dim=1, keepdim=True)
    S = qn @ dn.T
    wv, wi = torch.topk(S, 256, dim=1)
    assert (D.cpu().double() - wv).abs().max().item() <= 1e-5
    # same neighbours wherever float64 separates them by more than the fp32 error of a 768-term sum
    gaps = (wv[:, :-1] - wv[:, 1:]).abs()
    clear = torch.cat([gaps > 1e-5, torch.ones(300, 1, dtype=torch.bool)], 1) & torch.cat([torch.ones(300, 1, dtype=torch.bool), gaps > 1e-5], 1)
    assert (I.cpu()[clear] == wi[clear]).all()
    assert (I.cpu() == wi).float().mean().item() > 0.99
    # the inputs were not modified
    assert torch.equal(db, db.clone())
