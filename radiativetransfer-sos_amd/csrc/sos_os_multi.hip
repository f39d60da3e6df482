// csrc/sos_os_multi.hip -- the LDS-resident solver of sos_os.hip built a second time as k_sos_os_multi: every bin reads its
// wavelength context from a device table (SosBins::ctxs / ctx_of_bin) instead of the kernel argument, so that ONE launch
// covers the bins of many wavelengths (hyperspectral runs: 5-100 bins per wavelength fill a fraction of the 256 CUs).
// A separate translation unit, so that the single-wavelength kernels keep their code generation and the two build in parallel.
#define SOS_MULTI 1
#include "sos_os.hip"
