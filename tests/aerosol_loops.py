"""Host restatements used as checkers -- test infrastructure: the statement-for-statement loop form of SOS_DECOMPO_LEGENDRE
(src/SOS_AEROSOLS.F:3924-4390), against which the product's vectorised aerosols.decompo_legendre is checked bit for bit, and
the numpy form of SOS_GRANU (:4392-4820), which reproduces the reference's Aerosols.txt from its own MIE records and against
which the device kernel k_granu (the product path) is checked (tests/test_aerosols.py)."""
import importlib
import math

import numpy as np

_A = importlib.import_module("radiativetransfer-sos_amd.aerosols")
MU1_TRONCA, MU2_TRONCA, SEUIL_TRONCA, AerosolError, _seq_sum = _A.MU1_TRONCA, _A.MU2_TRONCA, _A.SEUIL_TRONCA, _A.AerosolError, _A._seq_sum


def decompo_legendre_loops(itronc, xmu, xhr, os_nb, p11_in, p12, p22, p33):
    """The statement-for-statement form of decompo_legendre (scalar loops), kept as what the vectorised form is tested against.
    SOS_DECOMPO_LEGENDRE: forward-peak truncation (log-linear extrapolation of P11 beyond acos(0.94), slope taken between
    acos(0.8) and acos(0.94)) and the expansions alpha, beta, gamma, zeta (0:os_nb), normalised by beta_0.
    Returns dict(alpha, beta, gamma, zeta, beta22, delta33, coef_tronca, itronc)."""
    w = len(xmu)
    n = (w - 1) // 2
    J = lambda j: j + n
    ttt = p11_in.copy()
    kk = np.arange(os_nb + 1)
    while True:
        p11 = ttt.copy()
        if itronc:
            k1 = next((j - 1 for j in range(1, n + 1) if xmu[J(j)] > MU1_TRONCA), None)
            k2 = next((j - 1 for j in range(1, n + 1) if xmu[J(j)] > MU2_TRONCA), None)
            if k1 is None or k2 is None:
                raise AerosolError("truncation angles outside the Mie angle set")
            aa = (math.log10(p11[J(k2)]) - math.log10(p11[J(k1)])) / (math.acos(xmu[J(k2)]) - math.acos(xmu[J(k1)]))
            x1, x2 = math.log10(p11[J(k2)]), math.acos(xmu[J(k2)])
            for j in range(k2 + 1, n + 1):
                p11[J(j)] = 10 ** (x1 + aa * (math.acos(xmu[J(j)]) - x2))
        beta11 = np.zeros(os_nb + 1)
        for j in range(-n, n + 1):
            if j == 0:
                continue
            x, xr = p11[J(j)] * xhr[J(j)], xmu[J(j)]
            plm, pl = 0., 1.
            for k in range(os_nb + 1):
                beta11[k] = beta11[k] + x * pl
                plm, pl = pl, ((2 * k + 1.) * xr * pl - k * plm) / (k + 1.)
        beta11 = (2 * kk + 1) * beta11 * .5
        coef = 2 * (1 - beta11[0]) if itronc else 0.0
        if itronc and coef < SEUIL_TRONCA:
            itronc = 0                    # truncation too small to matter: start again without it (SOS_AEROSOLS.F:4195-4214)
            continue
        break
    gamma12, beta22, delta33 = np.zeros(os_nb + 1), np.zeros(os_nb + 1), np.zeros(os_nb + 1)
    for j in range(-n, n + 1):
        if j == 0:
            continue
        xr = xmu[J(j)]
        pol = np.zeros(os_nb + 2)
        pol[2] = 3. * (1. - xr ** 2) / 2. / math.sqrt(6.0)
        xxx = xhr[J(j)] * p12[J(j)] * p11[J(j)] / ttt[J(j)]
        for k in range(2, os_nb + 1):
            d = (2. * k + 1.) / math.sqrt(1. * (k + 3.) * (k - 1.))
            e = math.sqrt(1. * (k + 2.) * (k - 2.)) / (2. * k + 1.)
            pol[k + 1] = d * (xr * pol[k] - e * pol[k - 1])
            gamma12[k] = gamma12[k] + xxx * pol[k]
        x = xhr[J(j)] * p22[J(j)] * (p11[J(j)] / ttt[J(j)])
        xx = xhr[J(j)] * p33[J(j)] * p11[J(j)] / ttt[J(j)]
        plm, pl = 0., 1.
        for k in range(os_nb + 1):
            beta22[k] = beta22[k] + x * pl
            delta33[k] = delta33[k] + xx * pl
            plm, pl = pl, ((2. * k + 1.) * xr * pl - k * plm) / (k + 1.)
    beta22 = beta22 * (2. * kk + 1.) * .5
    delta33 = delta33 * (2. * kk + 1.) * .5
    gamma12 = gamma12 * (2. * kk + 1.) * .5
    alp, zeta = np.zeros(os_nb + 1), np.zeros(os_nb + 1)
    f = np.float32
    for i in range(2, os_nb + 1):               # CO1, CO2, X2 are REAL*4 expressions (as in SOS_MAT_FRESNEL)
        co1 = float(f(4) * (f(2 * i) + f(1.)) / f(i) / (f(i) - f(1.)) / (f(i) + f(1.)) / (f(i) + f(2.)))
        co2 = float(f(i) * (f(i) - f(1.)) / ((f(i) + f(1.)) * (f(i) + f(2.))))
        co3 = co2 * delta33[i]
        co2 = co2 * beta22[i]
        nn, mm = int(i * .5), int((i - 1) * .5)
        s1 = s2 = s3 = s4 = 0.
        for j in range(1, nn + 1):
            x2 = float((f(i) - f(1.)) * (f(i) - f(1.)) - f(3.) * (f(2 * j) - f(1.)) * f(i - j))
            s1 = s1 + x2 * beta22[i - 2 * j]
            s2 = s2 + x2 * delta33[i - 2 * j]
        for j in range(0, mm + 1):
            x2 = float((f(i) - f(1.)) * (f(i) - f(1.)) - f(3.) * f(j) * (f(2 * i - 2 * j) - f(1.)))
            s3 = s3 + x2 * beta22[i - 2 * j - 1]
            s4 = s4 + x2 * delta33[i - 2 * j - 1]
        zeta[i] = co3 - co1 * (s2 - s3)
        alp[i] = co2 - co1 * (s1 - s4)
    z1 = beta11[0]
    return dict(alpha=alp / z1, beta=beta11 / z1, gamma=gamma12 / z1, zeta=zeta / z1, beta22=beta22 / z1, delta33=delta33 / z1,
                coef_tronca=float(coef), itronc=itronc)


def granu_host(rec, igranu, v1, v2, v3, wa):
    """SOS_GRANU: integral of the Mie records over the size distribution (igranu 1: log-normal, modal radius v1, ln-std v2;
    2: Junge, r0 = v1, slope v2, rmax = v3).  Returns kmat1, kmat2 (per particle), somme_nr, p11, p12, p33 [2N+1]."""
    alpha = rec["alpha"]                               # REAL*4 in the file
    a64 = alpha.astype(np.float64)
    r = a64 * wa / 2. / math.pi
    pas = np.full(len(alpha), np.float32(0.0001), dtype=np.float32)
    for lim, st in ((0.10, 0.001), (1.00, 0.01), (10., 0.05), (30., 0.10), (100., 1.00)):
        pas[alpha > np.float32(lim)] = np.float32(st)
    pas_prev = np.concatenate([[np.float32(0.0001)], pas[:-1]]).astype(np.float64)
    stop = a64 >= (rec["alphaf"] - pas_prev)
    nuse = int(np.argmax(stop)) if stop.any() else len(alpha)
    if igranu == 1:
        b = np.log(r / v1) / v2
        nr = np.exp(-b * b / 2.) / (r * v2 * math.sqrt(2 * math.pi))
    elif igranu == 2:
        over = r > v3
        if over[:nuse].any():
            nuse = int(np.argmax(over))
        nr = np.where(r <= v1, v1 ** (-v2), r ** (-v2))
    else:
        raise AerosolError("unknown size distribution %d" % igranu)
    sl = slice(0, nuse)
    # the record loop of SOS_GRANU accumulates in file order (SOS_AEROSOLS.F:4600-4618): sequential sums (cumsum), not numpy's
    # pairwise np.sum -- with the reference's own records this gives its Aerosols.txt digit for digit (tests/test_aerosols.py)
    seq0 = lambda a: np.cumsum(a, axis=0)[-1]
    pr = wa * pas[sl].astype(np.float64) / 2. / math.pi
    x1 = nr[sl] * pr * math.pi * r[sl] ** 2
    kmat1 = float(seq0(x1 * rec["qext"][sl].astype(np.float64)))
    x1s = rec["qsca"][sl].astype(np.float64) * x1
    kmat2 = float(seq0(x1s))
    p11 = seq0(rec["imie"][sl].astype(np.float64) * x1s[:, None]) / kmat2
    p12 = seq0(rec["qmie"][sl].astype(np.float64) * x1s[:, None]) / kmat2
    p33 = seq0(rec["umie"][sl].astype(np.float64) * x1s[:, None]) / kmat2
    somme_nr = float(seq0(nr[sl] * pr))
    return kmat1 / somme_nr, kmat2 / somme_nr, somme_nr, p11, p12, p33
