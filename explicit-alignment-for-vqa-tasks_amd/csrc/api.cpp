// ABI bookkeeping for libeavqa_hip.so (include/eavqa.h).
#include <hip/hip_runtime.h>
#include <string.h>
#include "eavqa.h"
#include "eavqa_test.h"

extern "C" int eavqa_abi_version(void) { return EAVQA_ABI_VERSION; }

extern "C" const char* eavqa_strerror(int code) {
    switch (code) {
        case EAVQA_OK: return "ok";
        case EAVQA_E_ARG: return "bad argument (null pointer or non-positive size)";
        case EAVQA_E_ALIGN: return "pointer or leading dimension not aligned as required";
        case EAVQA_E_SHAPE: return "shape not supported by the kernel";
        case EAVQA_E_DTYPE: return "unknown dtype or activation id";
        case EAVQA_E_LAUNCH: return "HIP launch failed";
        case EAVQA_E_ARCH: return "device is not gfx950 (MI355X)";
        default: return "unknown eavqa error code";
    }
}

extern "C" int eavqa_check_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return EAVQA_E_ARCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return EAVQA_E_ARCH;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? EAVQA_OK : EAVQA_E_ARCH;
}
