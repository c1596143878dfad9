// Library identification + error plumbing.
#include "dz_common.h"

int g_dz_last_hip_error = 0;

extern "C" const char* dz_version(void) { return "dqnzoo_hip 0.1 (gfx950)"; }
extern "C" int dz_last_hip_error(void) { return g_dz_last_hip_error; }
extern "C" const char* dz_built_arch(void) { return "gfx950"; }
