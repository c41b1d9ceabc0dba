// Kernel instantiations for one environment (its own translation unit so the six compile in parallel).
#include "launch.hpp"
namespace excenv {
EnvVTable vtable_msd() { return EnvEntry<MassSpringDamper>::vtable(); }
}  // namespace excenv
