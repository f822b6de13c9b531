// errors.hpp -- status codes, the thread-local error message and the early-return macro.  No HIP in here: the host-only
// translation units (container.hpp, rc_format.hpp, hostcoder.hip -- what reads untrusted bytes) build with a plain C++ compiler
// under AddressSanitizer / UBSan (tools/asan_host.sh) from this header alone.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gauspcc.h"

namespace gpcc {

extern thread_local char g_err[512];
extern thread_local long long g_launches;   // kernel launch sites passed by this thread (LAUNCH_CHECK): gpcc_debug_launches

inline int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

constexpr int MAXLV = 24;          // stored octree levels a container may announce (21 in practice: 21-bit coordinates)

#define GP_TRY(expr)                 \
    do {                             \
        int s_ = (expr);             \
        if (s_ != GPCC_OK) return s_; \
    } while (0)

}  // namespace gpcc
