#include "common.h"

extern "C" int dh_abi_version(void) { return DH_ABI_VERSION; }

extern "C" const char* dh_error_string(int code) {
    switch (code) {
        case DH_OK: return "ok";
        case DH_ERR_BAD_ARG: return "bad argument (null pointer, size or alignment contract violated)";
        case DH_ERR_UNSUPPORTED: return "unsupported dtype for this entry point";
        case DH_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown error code";
    }
}
