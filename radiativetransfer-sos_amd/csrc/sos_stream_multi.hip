// csrc/sos_stream_multi.hip -- the streamed-field solver of sos_stream.hip built a second time as k_sos_stream_multi, with a
// per-bin wavelength context (see sos_os_multi.hip).
#define SOS_MULTI 1
#include "sos_stream.hip"
