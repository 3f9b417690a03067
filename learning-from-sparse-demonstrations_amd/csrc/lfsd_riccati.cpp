// Second translation unit of a model library: the Riccati sweep, compiled with its own scheduler settings (runtime.TUNED_RICCATI; lfsd_internal.h).
#include "lfsd_internal.h"
#include "lfsd_riccati.inc"
