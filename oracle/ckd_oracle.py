"""TEST INFRASTRUCTURE ONLY -- scalar restatement of the reference's CKD bin enumeration
(SOS_PROC.F:3381-3404 normalisation sum, :3459-3487 per-bin weight), written as the same 8 nested loops.
The reference code is inline in SOS_PROC (no callable entry point), so this restatement is pinned by reading only:
"parity unpinned" against reference OUTPUT for this helper (the solver/aggregate it feeds are pinned by goldens)."""


def ckd_bins(nexp, kdis_ai):
    def prod(i):
        v = kdis_ai[i[0] - 1][0]
        for g in range(1, 8):
            v = v * kdis_ai[i[g] - 1][g]
        return v
    idx = []
    for i1 in range(1, nexp[0] + 1):
        for i2 in range(1, nexp[1] + 1):
            for i3 in range(1, nexp[2] + 1):
                for i4 in range(1, nexp[3] + 1):
                    for i5 in range(1, nexp[4] + 1):
                        for i6 in range(1, nexp[5] + 1):
                            for i7 in range(1, nexp[6] + 1):
                                for i8 in range(1, nexp[7] + 1):
                                    idx.append((i1, i2, i3, i4, i5, i6, i7, i8))
    s = 0.0
    for i in idx:
        s = s + prod(i)
    return idx, [prod(i) / s for i in idx], s
