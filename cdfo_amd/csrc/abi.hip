// ABI bookkeeping for libcdfo_hip.so.
#include "common.h"

extern "C" int cdfo_abi_version(void) { return 1; }
extern "C" const char* cdfo_build_info(void) { return "libcdfo_hip gfx950 (CDNA4) " __DATE__ " " __TIME__; }
