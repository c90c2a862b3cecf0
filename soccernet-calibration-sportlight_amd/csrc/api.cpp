// libsncal.so: version + thread-local error string.
#include "common.hpp"
#include "x3.hpp"
#include <cstring>

namespace sncal {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
LaunchEvents& launch_events() { static thread_local LaunchEvents e; return e; }
}  // namespace sncal

extern "C" int sncal_version(void) { return SNCAL_VERSION; }
extern "C" const char* sncal_last_error(void) { return sncal::g_err; }
extern "C" const char* sncal_x3_name(void) { return SNCAL_X3_NAME; }

// A HIP stream whose kernels may only run on `cus_per_xcd` compute units of every XCD (hipExtStreamCreateWithCUMask; the KFD deals
// mask bit i to XCC i % 8, so bits [0, 8 n) are n CUs in each of the eight XCDs -- a mask that leaves an XCC empty is ignored for
// that XCC).  The frame pipeline puts the camera solves on such streams: a solve wavefront owns a SIMD's whole register file for up to
// hundreds of milliseconds (20000 Levenberg-Marquardt iterations at the reference's criterion), every kernel's workgroups are dealt
// to the XCDs round-robin by the hardware, and so ONE held CU costs every kernel 1 / 32 of its XCD's throughput (measured: 64 such
// waves anywhere on the chip +13.5 % on the network, confined to one CU per XCD +3.6 %; tools/dev/cumask_probe.py).
// NOTE the stream is a BLOCKING stream (HIP offers no flags with a CU mask): it synchronises with the legacy null stream in both
// directions.  Keep the null stream idle while such streams carry work (pipeline.py runs the network on its own non-blocking stream).
extern "C" int sncal_stream_create_cu_mask(int cus_per_xcd, void** out) {
    SNCAL_CHECK_ARG(out && cus_per_xcd >= 1 && cus_per_xcd <= 32, "sncal_stream_create_cu_mask: cus_per_xcd %d", cus_per_xcd);
    int dev = 0, cus = 0;
    SNCAL_CHECK_HIP(hipGetDevice(&dev));
    SNCAL_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    SNCAL_CHECK_ARG(cus % 8 == 0 && cus_per_xcd * 8 <= cus, "sncal_stream_create_cu_mask: %d CUs do not split into 8 XCDs of >= %d", cus, cus_per_xcd);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < 8 * cus_per_xcd; ++b) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t s = nullptr;
    SNCAL_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, 8, mask));
    *out = s;
    return SNCAL_OK;
}
extern "C" int sncal_stream_destroy(void* stream) {
    if (stream) {
        sncal::release_solve_scratch(sncal::as_stream(stream), false);      // the block sncal_calibrate kept for this stream goes with it
        SNCAL_CHECK_HIP(hipStreamDestroy(sncal::as_stream(stream)));
    }
    return SNCAL_OK;
}
// Free what the library holds outside its handles: the per-(device, stream) scratch blocks of sncal_calibrate's convenience form
// (synchronises those streams).  Network / decoder handles are released by their own destroy calls.
extern "C" int sncal_shutdown(void) {
    return sncal::release_solve_scratch(nullptr, true);
}
