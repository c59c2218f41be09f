// placeholder until the conv stack lands: every entry point reports VQ_E_UNSUPPORTED
#include "vq_common.h"
using namespace vq;
extern "C" {
int vq_tsn_create(const vq_tensor_desc*, int32_t, const vq_layer_desc*, int32_t, const float*, int64_t, int32_t, int32_t, int32_t, vq_tsn** out) { if (out) *out = nullptr; return fail(VQ_E_UNSUPPORTED, "TSN path not built yet"); }
int vq_tsn_destroy(vq_tsn*) { return VQ_OK; }
int vq_tsn_set_stream(vq_tsn*, void*) { return fail(VQ_E_UNSUPPORTED, "TSN path not built yet"); }
int vq_tsn_forward(vq_tsn*, const uint8_t*, int32_t, int32_t, int32_t, const float*, double*, float*) { return fail(VQ_E_UNSUPPORTED, "TSN path not built yet"); }
int vq_tsn_feat_devptr(vq_tsn*, void**, void**) { return fail(VQ_E_UNSUPPORTED, "TSN path not built yet"); }
int vq_tsn_read_tensor(vq_tsn*, int32_t, int32_t, float*) { return fail(VQ_E_UNSUPPORTED, "TSN path not built yet"); }
int vq_tsn_flops_per_crop(vq_tsn*, double*) { return fail(VQ_E_UNSUPPORTED, "TSN path not built yet"); }
}
