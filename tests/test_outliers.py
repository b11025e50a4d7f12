"""remove_outliers (radius outlier filter) on the GPU against oracle/outlier_oracle.py."""
import numpy as np
import pytest

from oracle import outlier_oracle as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.binding.Context(0)
    yield c
    c.close()


def _cloud_with_outliers(pkg, n, n_out, seed):
    pts = pkg.synthetic.sphere_shell(n, seed)
    rng = np.random.default_rng(seed)
    out = pts[:n_out].copy()
    out["x"], out["y"], out["z"] = [rng.uniform(0.0, 1.0, n_out).astype(np.float32) for _ in range(3)]
    both = np.concatenate([pts, out])
    return both[rng.permutation(len(both))]


@pytest.mark.parametrize("n,k,radius", [(20_000, 4, 0.02), (50_000, 8, 0.01), (5_000, 1, 0.05), (30_000, 30, 0.03)])
def test_remove_outliers_matches_oracle(pkg, ctx, n, k, radius):
    pts = _cloud_with_outliers(pkg, n, n // 50, n)
    got = ctx.remove_outliers(pts, k, radius)
    want = R.remove_outliers(pts, k, radius)
    assert got.tobytes() == want.tobytes()
    assert len(got) < len(pts)                      # some of the scattered points went
    assert len(got) > 0.6 * n                       # most of the surface stayed


def test_remove_outliers_edge_cases(pkg, ctx):
    pts = _cloud_with_outliers(pkg, 2_000, 40, 3)
    assert ctx.remove_outliers(pts, 0, 0.01).tobytes() == pts.tobytes()      # K = 0: filter off (impl.hpp:1844)
    assert len(ctx.remove_outliers(pts[:1], 1, 0.01)) == 0                    # a single point has no neighbour
    bad = pts.copy()
    bad["x"][::7] = np.nan
    got = ctx.remove_outliers(bad, 3, 0.05)
    assert got.tobytes() == R.remove_outliers(bad, 3, 0.05).tobytes()
    assert np.isfinite(got["x"]).all()
    dup = np.concatenate([pts[:10]] * 3)                                       # coincident points count as neighbours
    assert len(ctx.remove_outliers(dup, 2, 1e-4)) == 30
    with pytest.raises(pkg.binding.PccError):
        ctx.remove_outliers(pts, 3, 0.0)
