// abi.hip -- version / error-text entry points of the C ABI (include/lina_gla.h).
#include <lina_dev.h>
#include "lina_common.h"

namespace lina {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace lina

extern "C" int lina_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* lina_last_error(void) { return lina::last_error_buf(); }
