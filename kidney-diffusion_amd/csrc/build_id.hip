// kd_build_id(): the sha256 prefix of the sources this library was compiled from (build_id.py),
// written by the Makefile into $(OBJDIR)/build_id_gen.h.  This is the only translation unit that
// is recompiled for every source change; the header-dependency files (-MMD) rebuild the rest.
#include "build_id_gen.h"
#include "../../include/kd_engine.h"

extern "C" const char* kd_build_id(void) { return KD_BUILD_ID; }
