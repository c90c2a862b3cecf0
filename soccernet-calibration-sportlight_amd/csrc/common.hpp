// Shared host-side helpers for libsncal.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

// solve.hip: release the scratch block(s) sncal_calibrate keeps per (device, stream) -- one stream's, or all of them
int release_solve_scratch(hipStream_t st, bool all);

// Per-launch timing for sncal_hrnet_set_profiling: the plan executor arms a (start, stop) event pair before an op and
// the op's launch attaches them to its dispatch (hipExtLaunchKernelGGL: timestamps of the kernel's own start and
// completion, no extra marker packets in the queue).  SNCAL_LAUNCH_FIRST / _LAST split the pair over a two-kernel op.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
LaunchEvents& launch_events();

#define SNCAL_LAUNCH_EV(kernel, grid, block, lds, stream, e0_, e1_, ...)                                   \
    do { if ((e0_) || (e1_)) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, e0_, e1_, 0, __VA_ARGS__); \
         else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); } while (0)
#define SNCAL_LAUNCH(kernel, grid, block, lds, stream, ...)                                                 \
    do { ::sncal::LaunchEvents& le_ = ::sncal::launch_events(); hipEvent_t a_ = le_.start, b_ = le_.stop;   \
         le_.start = le_.stop = nullptr; SNCAL_LAUNCH_EV(kernel, grid, block, lds, stream, a_, b_, __VA_ARGS__); } while (0)
#define SNCAL_LAUNCH_FIRST(kernel, grid, block, lds, stream, ...)                                           \
    do { ::sncal::LaunchEvents& le_ = ::sncal::launch_events(); hipEvent_t a_ = le_.start; le_.start = nullptr; \
         SNCAL_LAUNCH_EV(kernel, grid, block, lds, stream, a_, (hipEvent_t) nullptr, __VA_ARGS__); } while (0)
#define SNCAL_LAUNCH_LAST(kernel, grid, block, lds, stream, ...)                                            \
    do { ::sncal::LaunchEvents& le_ = ::sncal::launch_events(); hipEvent_t b_ = le_.stop; le_.stop = nullptr;   \
         SNCAL_LAUNCH_EV(kernel, grid, block, lds, stream, (hipEvent_t) nullptr, b_, __VA_ARGS__); } while (0)

}  // namespace sncal
