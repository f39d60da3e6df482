"""The keyword sets of the spectrum tests: the reference's end-to-end goldens (tests/golden/sos_proc_*.npz) as a list of
sos_proc calls, aerosols entering through the reference's own Aerosols.txt (-AER.UserFile)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RANDOM_CASES = sorted(f[len("sos_proc_"):-4] for f in os.listdir(GOLD) if f.startswith("sos_proc_rand_"))
# fixed goldens that need no extra inputs: configs 1, 2, 4, 5 (one wavenumber), CKD bands, land and sea surfaces, polar views
FIXED_CASES = ["cfg1_lambert", "cfg2_lnd_lambert", "cfg4_glitter_bilnd", "cfg5_ckd_maignan_25bins", "cfg5_roujean_maignan",
               "ckd_h2o_o2_25bins_flatsea", "ckd_o2a_5bins", "ckd_o2a_mode2", "flatsea_lnd", "flatsea_zout", "glitter_polar",
               "land_breon", "land_rondeaux", "land_roujean", "layer_1_3km_lnd", "nopolar_polar"]


def build(rs, workdir, names=None, resroot=False):
    """Returns (kwargs_list, goldens, coef_tronca overrides, rtols) for the named goldens (default: the 20 random keyword sets
    twice + the fixed cases: 56 calls)."""
    if names is None:
        names = RANDOM_CASES + FIXED_CASES + RANDOM_CASES
    kws, golds, coefs, rtols = [], [], [], []
    for k, name in enumerate(names):
        g = np.load(os.path.join(GOLD, "sos_proc_%s.npz" % name))
        user = {kk: (os.path.join(GOLD, v[8:]) if isinstance(v, str) and v.startswith("@GOLDEN/") else v)
                for kk, v in json.loads(str(g["user_json"])).items()}
        d = os.path.join(str(workdir), "%03d_%s" % (k, name))
        os.makedirs(d, exist_ok=True)
        user.update({"-SOS_Main.Log": "NO_LOG_FILE"})
        user.setdefault("-SOS.Flux", "NO_OUTPUT")
        if name.startswith("rand_"):
            user["-SOS.Flux"] = "NO_OUTPUT"
        if resroot:
            user["-SOS_Main.ResRoot"] = d
        else:
            user.pop("-SOS_Main.ResRoot", None)
        coef = None
        if user["-AER.AOTref"] != 0.0 and "aer_alpha" in g.files:
            f = os.path.join(d, "Aerosols_user.txt")
            rs.write_aerosols_file(f, {q: g["aer_" + q] for q in ("alpha", "beta", "gamma", "zeta", "a_tronc", "piztr", "piz")},
                                   *g["kmat"])
            user["-AER.UserFile"] = f
            coef = float(g["coef_tronca_userfile"]) if "coef_tronca_userfile" in g.files else 0.0
        kws.append(rs.sos_proc_kwargs(rs.update_parameters(rs.default_parameters(), user), trace=False))
        golds.append(g)
        coefs.append(coef)
        rtols.append(2e-7 if int(user.get("-SURF.Type", 0)) >= 3 else 1e-9)
    return kws, golds, coefs, rtols
