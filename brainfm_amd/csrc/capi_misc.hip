#include "bfm_common.h"
extern "C" const char* bfm_version(void) { return "brainfm_hip 0.1 gfx950"; }
