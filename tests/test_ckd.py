"""CKD bin enumeration / weights against the nested-loop restatement."""
import importlib

import numpy as np
import pytest

from oracle import ckd_oracle

ckd = importlib.import_module("radiativetransfer-sos_amd.ckd")


@pytest.mark.parametrize("nexp", [[1] * 8, [5, 1, 1, 1, 1, 1, 5, 1], [2, 3, 1, 1, 2, 1, 4, 1], [5, 5, 5, 1, 1, 1, 1, 1]])
def test_bin_order_and_weights(nexp):
    rng = np.random.default_rng(7)
    a = rng.uniform(0.01, 1.0, (5, 8))
    for g in range(8):                       # each gas's weights sum to ~1 like the tables (not exactly: renormalised)
        a[:nexp[g], g] /= a[:nexp[g], g].sum() * (1 + 1e-7 * g)
    ik, aik, s = ckd.ckd_bin_weights(nexp, a)
    ref_ik, ref_aik, ref_s = ckd_oracle.ckd_bins(nexp, a.tolist())
    assert ik.shape == (int(np.prod(nexp)), 8)
    assert [tuple(r) for r in ik.tolist()] == ref_ik          # gas 1 outermost ... gas 8 innermost
    assert s == ref_s                                         # bit-exact: same product and summation order
    assert np.array_equal(aik, np.array(ref_aik))
    assert abs(aik.sum() - 1.0) < 1e-13


def test_rejects_bad_tables():
    with pytest.raises(ValueError):
        ckd.ckd_bin_weights([1] * 7, np.ones((5, 8)))
    with pytest.raises(ValueError):
        ckd.ckd_bin_weights([6, 1, 1, 1, 1, 1, 1, 1], np.ones((5, 8)))
