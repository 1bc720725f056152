"""csrc/sort_rows.hip (dm_sort_rows_f32) against torch.sort(stable=True): the same permutation — ties, signed zeros,
infinities, every length class (one wave's worth, the 256- and 1 024-thread kernels, the 16 384 limit), strided rows."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.mark.parametrize('rows,n', [(1, 1), (1, 2), (3, 37), (2, 256), (1, 1000), (2, 2048), (2, 3000), (4, 9000), (1, 16384)])
@pytest.mark.parametrize('descending', [True, False])
def test_sort_rows_equals_torch_stable_sort(dev, rows, n, descending):
    from detmatch_amd import _lib
    g = torch.Generator().manual_seed(rows * 100003 + n)
    k = torch.randn(rows, n, generator=g)
    k = (k * 4).round() / 4                                  # many ties
    if n >= 8:
        k[:, 1], k[:, 2], k[:, 3], k[:, 5] = 0.0, -0.0, float('inf'), float('-inf')
    k = k.to(dev)
    want = torch.sort(k, dim=1, descending=descending, stable=True)[1]
    calls = _lib.SORT_ROWS_CALLS[0]
    got = _lib.sort_rows(k, descending=descending)
    assert _lib.SORT_ROWS_CALLS[0] == calls + 1 and got.dtype == torch.int64 and got.shape == want.shape
    assert torch.equal(got, want)
    if rows == 1:
        assert torch.equal(_lib.sort_rows(k[0], descending=descending), want[0])          # 1-D keys
    # a strided view of a wider matrix (rows with a pitch)
    wide = torch.cat([k, k.flip(1)], dim=1)
    assert torch.equal(_lib.sort_rows(wide[:, :n], descending=descending), want)


def test_sort_rows_falls_back_beyond_its_limit(dev):
    from detmatch_amd import _lib
    k = torch.randn(20000, device=dev)
    calls = _lib.SORT_ROWS_CALLS[0]
    assert torch.equal(_lib.sort_rows(k), torch.sort(k, descending=True, stable=True)[1])
    assert _lib.SORT_ROWS_CALLS[0] == calls                 # torch.sort took it
    assert _lib.lib().dm_sort_rows_max() == 16384
