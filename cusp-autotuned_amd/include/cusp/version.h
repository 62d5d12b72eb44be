// cusp/version.h -- CUSP_VERSION as the reference defines it (cusp/version.h:32-35: 600 = 0.6.0, the interface this layer mirrors), and the engine's own.
#pragma once
#include "detail/config.h"

#define CUSP_VERSION 600
#define CUSP_MAJOR_VERSION (CUSP_VERSION / 100000)
#define CUSP_MINOR_VERSION (CUSP_VERSION / 100 % 1000)
#define CUSP_SUBMINOR_VERSION (CUSP_VERSION % 100)
// the MI355X engine behind device_memory: CMI_VERSION of include/cusp_mi355x.h at build time, cmi_version() of the loaded library at run time
#define CUSP_MI355X_ENGINE_VERSION CMI_VERSION
