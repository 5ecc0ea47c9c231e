// amx_buildid.hip -- the identity of this build: sha256 of the sources, taken by the Makefile when the library is linked
// (include/amico_amd.h: amx_build_id).  A unit of its own so that any source change recompiles one line, not a solver.
#include "../../include/amico_amd.h"
#ifndef AMX_CSRC_HASH
#define AMX_CSRC_HASH "unknown"
#endif
extern "C" const char *amx_build_id(void) { return "amico_amd 1.0 csrc " AMX_CSRC_HASH; }
