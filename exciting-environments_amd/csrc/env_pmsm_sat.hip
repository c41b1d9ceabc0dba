// Kernel instantiations for the PMSM saturated (LUT) model — selected when excenv_props_t.pmsm_lut is set.
#include "launch.hpp"
namespace excenv {
EnvVTable vtable_pmsm_sat() { return EnvEntry<PmsmSat>::vtable(); }
}  // namespace excenv
