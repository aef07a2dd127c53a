// Version / error strings of the C ABI (include/peclr_hip.h).
#include "common.hpp"

extern "C" int peclr_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* peclr_error_string(int code) {
    switch (code) {
        case PECLR_OK: return "ok";
        case PECLR_ERR_NULL: return "required pointer is null";
        case PECLR_ERR_SHAPE: return "unsupported or inconsistent shape";
        case PECLR_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned / not a multiple of 4";
        case PECLR_ERR_WORKSPACE: return "workspace too small";
        case PECLR_ERR_UNSUPPORTED: return "unsupported flag or layout";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown error";
}

// Identity of the hipGraph capture the stream is in (0 = not capturing).  The host side packs the per-step weight planes
// once per CAPTURE: a replayed graph must re-split the weights it is about to use whatever the host-side stamps said
// when it was recorded.
extern "C" int peclr_stream_capture_id(void* stream, unsigned long long* id_out) {
    if (!id_out) return PECLR_ERR_NULL;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipError_t e = hipStreamGetCaptureInfo(static_cast<hipStream_t>(stream), &st, &id);
    if (e != hipSuccess) return static_cast<int>(e);
    *id_out = (st == hipStreamCaptureStatusActive) ? (id ? id : 1ull) : 0ull;
    return PECLR_OK;
}
