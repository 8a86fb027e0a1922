/* legacy_api.h — C-ABI of libbnn_hip_legacy.so: TEST-ONLY kernels that used to ride in libbnn_hip.so (ABI <= 11) as
 * independent implementations of the same results.  Nothing under bnn_amd/ loads this library; tests/helpers/legacy.py
 * binds it with ctypes.  Same conventions as include/bnn_hip.h (device pointers, status codes, stream as void*). */
#ifndef BNN_HIP_LEGACY_API_H_
#define BNN_HIP_LEGACY_API_H_
#include "../../../include/bnn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* The round-2 stem kernel (conv tile staged through LDS, pooled from there): what
 * bnn_hip_stem7x7_bn_relu_pool_pack_f32 computes, bit for bit, in flags 0 or BNN_HIP_STEM_FP16. */
int bnn_hip_legacy_stem_staged(const float* x, const float* w, const float* bn_scale, const float* bn_shift,
                               int N, int H, int W, int flags, float* out_f32, uint64_t* P, uint64_t* M, void* stream);

/* bnn_hip_bconv2d with the weight tile staged in LDS per 4-wave workgroup (3x3 tiled shapes, no zero weights). */
int bnn_hip_legacy_bconv2d_lds(const bnn_hip_conv_desc* d, const uint64_t* P, const uint64_t* M, const uint32_t* wbits,
                               const float* alpha, const float* bias, const float* post_scale, float* out,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif
