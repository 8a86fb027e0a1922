// pack_weight.hip — XNOR-Net weight binarisation into the kernel-facing bit layout.
//
// Replaces bnn/ops.py:129-140 (XNORWeightBinarizer.forward) and bnn/ops.py:116-127
// (_compute_alpha).  The reference recomputes `sign(W) * mean|W|` on every forward
// (bnn/layers/conv.py:92); here it runs once per weight version and its outputs
// (1 bit + 1 mask bit per weight, one float per output channel) are cached by the host.
//
// One wave (64 lanes) per output channel.  Reductions are defined precisely so that
// the CPU oracle reproduces them bit-for-bit:
//   * lane l accumulates, in double, the elements k = l, l+64, l+128, ... (ascending),
//   * the 64 partial sums are combined by an xor-butterfly (32,16,8,4,2,1),
//   * the total is divided in double and rounded once to float.
// centre:  mean[t] over input channels for each tap t (reference: x.mean(1), ops.py:131)
// value :  v = w - mean[t] in float (reference: x.sub(mean), ops.py:132)
// alpha :  mean over all (c,t) of |v|            (reference: ops.py:119, L1 norm / n)
// bit   :  v > 0 ; mask: v != 0 and not NaN      (reference: SignActivation, ops.py:66,136)
#include "bnn_dev.h"

namespace bnn {

constexpr int kMaxTaps = 1024;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

__global__ __launch_bounds__(64) void pack_weight_kernel(const float* __restrict__ w, int O, int C,
                                                         int taps, int cwc, int nchunk, int center,
                                                         int compute_alpha,
                                                         uint32_t* __restrict__ wbits,
                                                         uint32_t* __restrict__ wnz,
                                                         float* __restrict__ alpha,
                                                         int32_t* __restrict__ zero_flag) {
  __shared__ float mean[kMaxTaps];
  const int o = blockIdx.x;
  const int lane = threadIdx.x;
  const int ob = o / kOCB, j = o % kOCB;
  const int per_o = taps * cwc;  // words per (o, chunk)
  const int nwords = nchunk * per_o;
  uint32_t* wb = wbits + ((size_t)ob * nchunk * kOCB + j) * per_o;
  uint32_t* wz = wnz + ((size_t)ob * nchunk * kOCB + j) * per_o;
  const size_t chunk_stride = (size_t)kOCB * per_o;

  if (o >= O) {  // padding channel: all-zero weights, alpha 0
    for (int d = lane; d < nwords; d += kWave) {
      const int ch = d / per_o, rem = d - ch * per_o;
      wb[ch * chunk_stride + rem] = 0u;
      wz[ch * chunk_stride + rem] = 0u;
    }
    if (lane == 0) alpha[o] = 0.0f;
    return;
  }

  const float* wo = w + (size_t)o * C * taps;
  const int K = C * taps;

  for (int t = 0; t < taps; ++t) {
    float m = 0.0f;
    if (center) {
      double s = 0.0;
      for (int c = lane; c < C; c += kWave) s += (double)wo[(size_t)c * taps + t];
      s = wave_sum(s);
      m = (float)(s / (double)C);
    }
    if (lane == 0) mean[t] = m;
  }
  __syncthreads();

  if (compute_alpha) {
    double s = 0.0;
    for (int k = lane; k < K; k += kWave) {
      const int t = k % taps;
      const float v = wo[k] - mean[t];
      s += (double)fabsf(v);
    }
    s = wave_sum(s);
    if (lane == 0) alpha[o] = (float)(s / (double)K);
  } else if (lane == 0) {
    alpha[o] = 1.0f;
  }

  bool any_zero = false;
  for (int d = lane; d < nwords; d += kWave) {
    const int ch = d / per_o, rem = d - ch * per_o;
    const int t = rem / cwc, cw = rem - t * cwc;
    const int c0 = (ch * cwc + cw) * 32;
    const float m = mean[t];
    uint32_t bits = 0u, nz = 0u;
    for (int b = 0; b < 32; ++b) {
      const int c = c0 + b;
      if (c < C) {
        const float v = wo[(size_t)c * taps + t] - m;
        const bool pos = is_pos(v), neg = is_neg(v);
        bits |= (pos ? 1u : 0u) << b;
        nz |= ((pos || neg) ? 1u : 0u) << b;
        any_zero |= !(pos || neg);
      }
    }
    wb[ch * chunk_stride + rem] = bits;
    wz[ch * chunk_stride + rem] = nz;
  }
  if (__any(any_zero) && lane == 0) atomicOr(zero_flag, 1);
}

int launch_pack_weight(const float* w, int O, int C, int KH, int KW, int center, int compute_alpha,
                       const bnn_hip_wlayout& L, uint32_t* wbits, uint32_t* wnz, float* alpha,
                       int32_t* zero_flag, hipStream_t stream) {
  if (L.taps > kMaxTaps) return BNN_HIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(L.o_pad), dim3(64), 0, stream, w, O, C, L.taps, L.cwc,
                     L.nchunk, center, compute_alpha, wbits, wnz, alpha, zero_flag);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
