// Shared host-side helpers for libsncal.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/sncal.h"

namespace sncal {

void set_error(const char* fmt, ...);

#define SNCAL_CHECK_ARG(cond, ...)                                         \
    do { if (!(cond)) { ::sncal::set_error(__VA_ARGS__); return SNCAL_ERR_ARG; } } while (0)

#define SNCAL_CHECK_HIP(expr)                                              \
    do { hipError_t e_ = (expr);                                           \
         if (e_ != hipSuccess) {                                           \
             ::sncal::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
             return SNCAL_ERR_HIP; } } while (0)

#define SNCAL_CHECK_LAUNCH() SNCAL_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace sncal
