// Kernel instantiations for one environment (its own translation unit so the six compile in parallel).
#include "launch.hpp"
namespace excenv {
EnvVTable vtable_acrobot() { return EnvEntry<Acrobot>::vtable(); }
}  // namespace excenv
