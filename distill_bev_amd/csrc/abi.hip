// Library identity entry points of libdbev_hip.so (include/dbev_hip.h).
#include "common.h"

extern "C" int dbev_abi_version(void) { return DBEV_ABI_VERSION; }
extern "C" const char* dbev_target_arch(void) { return "gfx950"; }
