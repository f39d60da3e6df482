"""Diagnostic: the spectrum test's keyword list through sos_spectrum with a synchronisation and a progress line per stage
(SOS_SPECTRUM_DEBUG=1)."""
import importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SOS_ABS_ROOT"] = os.path.join(ROOT, "tests", "golden")
os.environ["SOS_SPECTRUM_DEBUG"] = "1"
import spectrum_cases
pkg = importlib.import_module("radiativetransfer-sos_amd")
rs = pkg.run_sos
names = sys.argv[1:] or None
kws, golds, coefs, rtols = spectrum_cases.build(rs, tempfile.mkdtemp(), names=names)
outs = rs.sos_spectrum(kws)
print("ok", len(outs))
