// Library-wide pieces of the C ABI: version and the thread-local error string.
#include "hupr_common.h"

namespace hupr {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
unsigned long long* launch_counter() {
    static unsigned long long n = 0;
    return &n;
}
}  // namespace hupr

extern "C" int hupr_version(void) { return 100; }
extern "C" const char* hupr_last_error(void) { return hupr::error_buffer(); }
extern "C" unsigned long long hupr_launch_count(void) { return __atomic_load_n(hupr::launch_counter(), __ATOMIC_RELAXED); }
