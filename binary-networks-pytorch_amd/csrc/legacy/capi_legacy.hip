// capi_legacy.hip — extern "C" boundary of the TEST-ONLY library (legacy_api.h).
#include <cstring>

#include "../bconv_core.h"
#include "legacy_api.h"

extern "C" {

int bnn_hip_legacy_stem_staged(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int N,
                               int H, int W, int flags, float* out_f32, uint64_t* P, uint64_t* M, void* stream) {
  if (!x || !w || !bn_scale || !bn_shift || N <= 0 || H <= 0 || W <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if ((!out_f32 && !P) || (P == nullptr) != (M == nullptr)) return BNN_HIP_ERR_INVALID_ARG;
  if (flags & ~BNN_HIP_STEM_FP16) return BNN_HIP_ERR_INVALID_ARG;     // no exact-fp32 mode in this kernel
  if ((long long)N * 64 * H * W / 16 * 4 > 0xFFFFFE00LL || (long long)N * 3 * H * W * 4 > 0xFFFFFE00LL)
    return BNN_HIP_ERR_TOO_LARGE;
  return bnn::launch_stem_split(x, w, bn_scale, bn_shift, N, H, W, (flags & BNN_HIP_STEM_FP16) != 0, out_f32, P, M,
                                static_cast<hipStream_t>(stream));
}

int bnn_hip_legacy_bconv2d_lds(const bnn_hip_conv_desc* d, const uint64_t* P, const uint64_t* M, const uint32_t* wbits,
                               const float* alpha, const float* bias, const float* post_scale, float* out,
                               void* stream) {
  if (!d || !P || !M || !wbits || !alpha || !out) return BNN_HIP_ERR_INVALID_ARG;
  if (d->N <= 0 || d->C <= 0 || d->H <= 0 || d->W <= 0 || d->O <= 0 || d->KH != 3 || d->KW != 3 ||
      d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h != 1 || d->dil_w != 1)
    return BNN_HIP_ERR_UNSUPPORTED;
  const int Ho = (d->H + 2 * d->pad_h - 3) / d->stride_h + 1, Wo = (d->W + 2 * d->pad_w - 3) / d->stride_w + 1;
  if (Ho <= 0 || Wo <= 0) return BNN_HIP_ERR_INVALID_ARG;
  if ((long long)d->N * d->O * Ho * Wo > (1LL << 30) - 1) return BNN_HIP_ERR_TOO_LARGE;
  bnn::ConvP p;
  std::memset(&p, 0, sizeof(p));
  p.P = reinterpret_cast<const uint32_t*>(P);
  p.M = reinterpret_cast<const uint32_t*>(M);
  p.W = wbits;
  p.alpha = alpha; p.bias = bias; p.scale = post_scale; p.out = out;
  p.N = d->N; p.H = d->H; p.Wd = d->W; p.Ho = Ho; p.Wo = Wo; p.O = d->O; p.C = d->C;
  p.KH = 3; p.KW = 3; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w; p.dh = 1; p.dw = 1;
  p.cw32 = 2 * ((d->C + 63) / 64);
  p.cwc = (p.cw32 % 4 == 0) ? 4 : 2;          // == choose_cwc() of the product library for 3x3 weights
  p.nchunk = p.cw32 / p.cwc;
  p.npix = d->N * Ho * Wo;
  p.c_off = 0; p.c_tot = d->O;
  return bnn::launch_bconv_lds(p, static_cast<hipStream_t>(stream));
}

}  // extern "C"
