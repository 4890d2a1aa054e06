// Library identity / device probe for libmotif_hip.so.
#include "common.h"
#include <string.h>

extern "C" int motif_abi_version(void) { return 9; }   // 9: motif_instance_norm_affine_ws (PWCNet_light), status bit 2 / trap of motif_conv2d_chain_fwd; 8: motif_conv2d_chain_fwd / motif_conv2d_chain_ws_words (conv_wino.hip, chain mode); 7: range status word (MotifConvDesc.status, `status` argument of motif_dcn_v2_fused_fwd_multi / motif_siren_synth_fwd / motif_siren_synth_pre_fwd), scaled low activation part in the two-part fp16 form; 6: MotifConvDesc.mma = 7 and the two-part fp16 forms (conv_wino.hip, siren_split.hip: motif_siren_pack_split mode + 8, pre = 3, `pre` argument of motif_siren_synth_pre_fwd); 5: option pp_rp removed, conv_engine 5 / 6 (conv_wino.hip), packed 3x3 blobs carry a Winograd block; 4: motif_siren_imnet_add_fwd, g_lr = NULL in motif_splat_motif_pre_fwd, siren split blobs in turns; 3: motif_set_option / motif_get_option (2: MotifConvDesc.mma, motif_siren_pack_split + pre=2, splat row0)

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

// ---- tuning / test switches: environment read once, then motif_set_option only ---------------------------------------
#include <stdlib.h>
#include <ctype.h>
namespace {
const char* const kOptNames[MOTIF_OPT_COUNT] = {"conv_dbg", "conv_ck", "conv_nospec", "conv_engine", "lds_pad", "corr81",
                                                "dcn_nowin", "dcn_waves", "dcn_front_pad", "dcn_back_pad", "siren_stagger", "conv_novec", "conv_nodirect", "conv_wino_tr", "conv_wino_rpre", "conv_chain_wgs", "resize_narrow", "conv_direct_quads"};
struct OptTable {
    int v[MOTIF_OPT_COUNT];
    OptTable() {
        for (int i = 0; i < MOTIF_OPT_COUNT; ++i) {
            char env[64] = "MOTIF_";
            int k = 6;
            for (const char* c = kOptNames[i]; *c && k < 62; ++c) env[k++] = (char)toupper((unsigned char)*c);
            env[k] = 0;
            const char* e = getenv(env);
            v[i] = 0;
            if (e && *e) {
                if (i == MOTIF_OPT_CORR81) v[i] = !strcmp(e, "tiled") ? 1 : !strcmp(e, "small") ? 2 : atoi(e);
                else v[i] = atoi(e);
            }
        }
    }
};
OptTable& opt_table() { static OptTable t; return t; }
int opt_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < MOTIF_OPT_COUNT; ++i) if (!strcmp(name, kOptNames[i])) return i;
    return -1;
}
}  // namespace

int motif_opt(int id) { return opt_table().v[id]; }

extern "C" int motif_set_option(const char* name, int value) {
    const int i = opt_index(name);
    if (i < 0) return MOTIF_EINVAL;
    opt_table().v[i] = value;
    return MOTIF_OK;
}

extern "C" int motif_get_option(const char* name, int* value) {
    const int i = opt_index(name);
    if (i < 0 || !value) return MOTIF_EINVAL;
    *value = opt_table().v[i];
    return MOTIF_OK;
}
