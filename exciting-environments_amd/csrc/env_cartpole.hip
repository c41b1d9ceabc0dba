// Kernel instantiations for one environment (its own translation unit so the six compile in parallel).
#include "launch.hpp"
namespace excenv {
EnvVTable vtable_cartpole() { return EnvEntry<CartPole>::vtable(); }
}  // namespace excenv
