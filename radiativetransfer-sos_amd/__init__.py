"""radiativetransfer-sos_amd: MI355X-native drop-in for the SOS-ABS hot path (per-CKD-bin successive
orders of scattering, Cox-Munk glitter matrices, azimuth recomposition, bin aggregation).

Hand-written HIP kernels for gfx950 behind a plain C ABI (include/sosgpu.h, libsosgpu.so); this package is
the host-side mirror of the reference's binding/run_sos.py parameter surface.  The directory name is not
a Python identifier: import it with importlib.import_module("radiativetransfer-sos_amd").
"""
from . import synth  # noqa: F401  (pure numpy)

__all__ = ["synth", "capi", "solver", "build_ext", "SosContext"]


def __getattr__(name):
    import importlib
    if name in ("capi", "solver", "build_ext", "run_sos", "dist", "surface", "ckd", "absorption", "aerosols", "spectrum_pool"):
        return importlib.import_module("." + name, __name__)
    if name in ("SosContext", "SosBinError"):
        return getattr(importlib.import_module(".solver", __name__), name)
    raise AttributeError(name)
