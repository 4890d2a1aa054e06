// Library identity / device probe for libmotif_hip.so.
#include "common.h"
#include <string.h>

extern "C" int motif_abi_version(void) { return 2; }   // 2: MotifConvDesc.mma, motif_siren_pack_split + pre=2, splat row0

extern "C" int motif_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return MOTIF_OK;
}
