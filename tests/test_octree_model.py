"""The data-parallel octree formulation (tests/octree_model.py) == the literal list-based oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import octree_model as M


def _random_case(rng, w, h, n):
    cells = rng.choice(w * h, size=min(n, w * h), replace=False)
    xs = (cells % w).astype(np.int32)
    ys = (cells // w).astype(np.int32)
    order = np.lexsort((xs, ys))  # row-major like FAST output (order is irrelevant to correctness)
    xs, ys = xs[order], ys[order]
    sc = rng.integers(7, 60, len(xs)).astype(np.int32)
    return xs, ys, sc


@pytest.mark.parametrize("seed", range(12))
def test_model_matches_oracle_random(seed):
    rng = np.random.default_rng(seed)
    w = int(rng.integers(40, 400)); h = int(rng.integers(30, 160))
    n = int(rng.integers(0, 1200))
    nf = int(rng.integers(1, 400))
    xs, ys, sc = _random_case(rng, w, h, n)
    a = O.distribute_octtree(xs, ys, sc, 16, 16 + w, 16, 16 + h, nf)
    b = M.distribute(xs, ys, sc, 16, 16 + w, 16, 16 + h, nf)
    assert a.tolist() == b.tolist()


def test_model_matches_oracle_clustered():
    rng = np.random.default_rng(99)
    # heavy clustering -> one-child splits and the "largest first" phase
    xs = np.concatenate([rng.integers(0, 12, 300), rng.integers(200, 260, 200)])
    ys = np.concatenate([rng.integers(0, 12, 300), rng.integers(40, 90, 200)])
    _, idx = np.unique(xs * 1000 + ys, return_index=True)
    xs, ys = xs[np.sort(idx)].astype(np.int32), ys[np.sort(idx)].astype(np.int32)
    sc = rng.integers(7, 255, len(xs)).astype(np.int32)
    for nf in (1, 5, 37, 100, 150, 1000):
        a = O.distribute_octtree(xs, ys, sc, 0, 300, 0, 100, nf)
        b = M.distribute(xs, ys, sc, 0, 300, 0, 100, nf)
        assert a.tolist() == b.tolist(), nf
