// Only compiled with `make -C kidney-diffusion_amd/csrc EXTRA=-DKD_EXPERIMENT` (see csrc/common.h): the A/B switches of the
// kernels and the plan builder read from the environment.  The product library never includes this file.
#pragma once
#include <stdlib.h>

static inline int kd_switch(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
