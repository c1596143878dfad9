// Library identification + error plumbing.
#include "dz_common.h"

int g_dz_last_hip_error = 0;

extern "C" const char* dz_version(void) { return "dqnzoo_hip 0.1 (gfx950)"; }
extern "C" int dz_last_hip_error(void) { return g_dz_last_hip_error; }
extern "C" const char* dz_built_arch(void) { return "gfx950"; }

extern "C" int dz_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(dz_field_t);
    case 1: return (int)sizeof(dz_prio_sample_args_t);
    case 2: return (int)sizeof(dz_rainbow_layout_t);
    case 3: return (int)sizeof(dz_rainbow_args_t);
    default: return -1;
  }
}
