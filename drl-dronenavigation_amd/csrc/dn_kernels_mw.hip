// The multi-wave step kernels of the configuration without the observation normaliser, compiled WITH machine LICM
// (build.py gives this translation unit its own flags; see dn_launch_step_many in dn_kernels.hip for the measurements).
#define DN_TU 2
#include "dn_kernels.hip"
