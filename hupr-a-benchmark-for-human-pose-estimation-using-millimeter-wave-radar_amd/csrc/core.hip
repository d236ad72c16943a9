// Library-wide pieces of the C ABI: version and the thread-local error string.
#include "hupr_common.h"

namespace hupr {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace hupr

extern "C" int hupr_version(void) { return 100; }
extern "C" const char* hupr_last_error(void) { return hupr::error_buffer(); }
