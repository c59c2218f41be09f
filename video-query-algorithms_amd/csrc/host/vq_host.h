// What the host-only translation units (csrc/host/*.cc: no HIP include, buildable by plain g++ with sanitizers) share with the
// rest of the library: the error plumbing.  vq::last_error_ref() lives in vq_sim.hip for the product and in
// tests/sanitize/host_main.cc for the sanitizer builds.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>

namespace vq {
std::string& last_error_ref();

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

inline int host_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}
}  // namespace vq
