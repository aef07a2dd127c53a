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
