// Second translation unit of a model library: the Riccati sweep, compiled with the SLP vectoriser on (lfsd_internal.h).
#include "lfsd_internal.h"
#include "lfsd_riccati.inc"
