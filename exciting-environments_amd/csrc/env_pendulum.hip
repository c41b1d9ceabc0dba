// Kernel instantiations for one environment (its own translation unit so the six compile in parallel).
#include "launch.hpp"
namespace excenv {
EnvVTable vtable_pendulum() { return EnvEntry<Pendulum>::vtable(); }
}  // namespace excenv
