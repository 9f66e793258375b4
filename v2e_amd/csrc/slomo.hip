// placeholder translation unit; real kernels follow
#include "common.h"
extern "C" {
int v2e_pack_conv_weight(const float *, float *, int, int, int, void *) { v2e_set_error("slomo not built"); return V2E_EINVAL; }
int v2e_conv2d_lrelu(const float *, int, const float *, int, int, const v2e_conv_desc *, float *, int, int, int, void *) { v2e_set_error("slomo not built"); return V2E_EINVAL; }
int64_t v2e_unet_workspace_bytes(int, int, int, int) { return 0; }
int v2e_unet_forward(const float *, int, const v2e_conv_desc *, int, float *, int, int, int, void *, void *) { v2e_set_error("slomo not built"); return V2E_EINVAL; }
int v2e_slomo_prep(const float *, const float *, const float *, const float *, int, int, int, int, float *, void *) { v2e_set_error("slomo not built"); return V2E_EINVAL; }
int v2e_slomo_fuse(const float *, const float *, const float *, const float *, const float *, int, int, int, int, float *, void *) { v2e_set_error("slomo not built"); return V2E_EINVAL; }
}
