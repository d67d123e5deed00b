// ABI housekeeping: version, last-error text, device probe.
#include "surs_common.h"

extern "C" int surs_abi_version(void) { return SURS_ABI_VERSION; }

extern "C" const char *surs_last_error(void) { return surs::err_buf(); }

extern "C" int surs_device_info(int *cu_count, char *arch_out) {
    int dev = 0;
    SURS_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SURS_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch_out) {
        strncpy(arch_out, prop.gcnArchName, 31);
        arch_out[31] = 0;
    }
    return 0;
}
