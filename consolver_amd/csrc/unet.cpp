// placeholder until the UNet executor lands (replaced in the next milestone)
#include "common.h"
extern "C" {
int cs_unet_create(const CsUNetConfig*, CsUNet**) { CS_FAIL(CS_E_UNSUPPORTED, "unet not built yet"); }
void cs_unet_destroy(CsUNet*) {}
int cs_unet_set_weight(CsUNet*, const char*, const float*, const int64_t*, int) { return CS_E_UNSUPPORTED; }
int cs_unet_num_weights(const CsUNet*) { return 0; }
const char* cs_unet_weight_name(const CsUNet*, int, int64_t*, int*) { return nullptr; }
int cs_unet_finalize(CsUNet*) { return CS_E_UNSUPPORTED; }
size_t cs_unet_workspace_bytes(const CsUNet*, int) { return 0; }
double cs_unet_flops(const CsUNet*, int) { return 0; }
int cs_unet_forward(CsUNet*, const void*, int, int, const float*, int, const void*, void*, void*, size_t, int, void*) { return CS_E_UNSUPPORTED; }
int cs_unet_set_profiling(CsUNet*, int) { return CS_E_UNSUPPORTED; }
int cs_unet_profile_entries(const CsUNet*) { return 0; }
const char* cs_unet_profile_entry(const CsUNet*, int, double*, double*, double*, int*) { return nullptr; }
}
