// cusp/version.h -- the version macros of the interface this layer mirrors (reference cusp/version.h: CUSP_VERSION 600 = 0.6.0, split into
// major / minor / subminor by division there), and the engine's own.
#pragma once
#include "detail/config.h"

#define CUSP_VERSION 600
#define CUSP_MAJOR_VERSION 0
#define CUSP_MINOR_VERSION 6
#define CUSP_SUBMINOR_VERSION 0
static_assert(CUSP_VERSION == CUSP_MAJOR_VERSION * 100000 + CUSP_MINOR_VERSION * 100 + CUSP_SUBMINOR_VERSION, "cusp/version.h: the parts must spell CUSP_VERSION");

// the MI355X engine behind device_memory: CMI_VERSION of include/cusp_mi355x.h at build time, cmi_version() of the loaded library at run time
#define CUSP_MI355X_ENGINE_VERSION CMI_VERSION
