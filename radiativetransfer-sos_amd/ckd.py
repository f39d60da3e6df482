"""CKD bin enumeration and weights of one wavelength (host side of the bin loop).

Mirrors the reference's 8 nested loops over the per-gas exponential terms (SOS_PROC.F:3381-3404 for the
normalisation sum, :3459-3487 for the per-bin weight): gas 1 (H2O) is the OUTERMOST loop, gas 8 (NO2) the innermost;
AIK = prod_g KDIS_AI(IK_g, g) evaluated left to right, then divided by the serial sum of all products.  The order
of the returned bins is the order in which the reference solves and aggregates them (SOS_AGGREGATE is called once per
bin in this order), which is also the order bins are sharded over ranks (dist.shard_range).

Gas order (SOS_PROC.F:3470-3477): H2O, CO2, O3, N2O, CO, CH4, O2, NO2.
"""
import numpy as np

GASES = ("H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "NO2")
NGAS = 8
NEXP_MAX = 5            # CTE_CKD_NAI_MAX: at most 5 exponential terms per gas (SOS.h)


def ckd_bin_weights(nexp, kdis_ai):
    """nexp[8] (ints >= 1), kdis_ai[NEXP_MAX][8] (KDIS_AI(IK, gas) of one wavelength, Fortran order [ik-1][gas-1]).
    Returns (ik[nb][8] int32 1-based term indices in solve order, aik[nb] float64 normalised weights, sum_aik)."""
    nexp = np.asarray(nexp, dtype=np.int64)
    a = np.asarray(kdis_ai, dtype=np.float64)
    if nexp.shape != (NGAS,) or a.ndim != 2 or a.shape[1] != NGAS:
        raise ValueError("nexp must have 8 entries and kdis_ai shape [n][8]")
    if (nexp < 1).any() or (nexp > a.shape[0]).any():
        raise ValueError("NEXP out of range")
    nb = int(np.prod(nexp))
    # C-order unravel of the flat bin index = gas 1 slowest, gas 8 fastest: the reference's loop nest
    ik = np.stack(np.unravel_index(np.arange(nb), tuple(int(n) for n in nexp)), axis=1).astype(np.int32)
    w = a[ik[:, 0], 0]
    for g in range(1, NGAS):
        w = w * a[ik[:, g], g]      # left-to-right product, one rounding per factor (SOS_PROC.F:3392-3395)
    s = 0.0
    for v in w:                     # serial sum in bin order (SOS_PROC.F:3397)
        s = s + float(v)
    return ik + 1, w / s, s
