// Force-included when building the UNMODIFIED reference with this image's libstdc++ 11, which
// predates C++23's std::unreachable (used at reference include/shapes.hpp:456).  SURVEY.md §8c.
#include <utility>
#if !defined(__cpp_lib_unreachable)
namespace std { [[noreturn]] inline void unreachable() { __builtin_unreachable(); } }
#endif
