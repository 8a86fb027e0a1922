// xnor_train.hip — the XNOR-Net weight binarizer of a training step, forward value and backward, one kernel each.
//
// Reference: XNORWeightBinarizer.forward (bnn/ops.py:129-140) evaluated under autograd,
//     Wc = W - mean_c(W)            (center_weights)          alpha = mean |Wc| per output channel (bnn/ops.py:116-127)
//     What = SignActivation(Wc) * alpha                        SignActivation.backward: grad * 1[|Wc| < 1] (bnn/ops.py:68-73)
// which torch runs as ~6 element-wise / reduction kernels forward and ~8 backward per layer (19 layers: 1.2 ms of a
// 21 ms ResNet-18 step).  Here: one workgroup per output channel, reductions in double in a fixed order (thread-strided
// partial sums, xor butterfly per wave, the wave sums in wave order): deterministic; alpha agrees with pack_weight.hip's
// to the last bit or two of the double sum (both round the same mean to fp32).
//     forward :  What[o,c,t] = sign(Wc) * alpha[o]
//     backward:  dWc = dWhat * alpha * 1[|Wc| < 1] + sign(Wc) * (sum_{c,t} dWhat * sign(Wc)) / K        (compute_alpha)
//                dWc = dWhat * 1[|Wc| < 1]                                                              (otherwise)
//                dW  = dWc - mean_c(dWc)  when centred, else dWc
#include "bnn_dev.h"

namespace bnn {

namespace xt {
constexpr int kMaxTaps = 1024;
constexpr int NT = 256;  // one 4-wave workgroup per output channel (round 4a: one wave — 64 waves on 1024 SIMDs for a
                         // 64-channel layer, every load a serial round trip: 15-26 us for 0.15-9 MB)
// sum over the workgroup, every thread gets it: lane-strided partials -> xor butterfly per wave -> the four wave sums added
// in wave order (a fixed order: deterministic)
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  __syncthreads();                         // `red` may still be read from the sum before
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float sgn(float v) { return is_pos(v) ? 1.0f : is_neg(v) ? -1.0f : 0.0f; }

// mean[t] over the input channels (0 when not centred) into LDS; returns alpha
__device__ __forceinline__ float centre_and_alpha(const float* __restrict__ wo, int C, int taps, int center,
                                                  int compute_alpha, float* mean, double* red) {
  const int tid = threadIdx.x, K = C * taps;
  if (center) {
    for (int t = 0; t < taps; ++t) {
      double s = 0.0;
      for (int c = tid; c < C; c += NT) s += (double)wo[(size_t)c * taps + t];
      const float m = (float)(block_sum(s, red) / (double)C);
      if (tid == 0) mean[t] = m;
    }
  } else {
    for (int t = tid; t < taps; t += NT) mean[t] = 0.0f;
  }
  __syncthreads();
  if (!compute_alpha) return 1.0f;
  double s = 0.0;
  for (int k = tid, t = tid % taps; k < K; k += NT, t = (t + NT) % taps) s += (double)fabsf(wo[k] - mean[t]);
  return (float)(block_sum(s, red) / (double)K);
}
}  // namespace xt

__global__ __launch_bounds__(xt::NT) void xnor_what_kernel(const float* __restrict__ w, int C, int taps, int center,
                                                           int compute_alpha, float* __restrict__ what,
                                                           float* __restrict__ alpha_out) {
  __shared__ float mean[xt::kMaxTaps];
  __shared__ double red[4];
  const int o = blockIdx.x, tid = threadIdx.x, K = C * taps;
  const float* wo = w + (size_t)o * K;
  const float alpha = xt::centre_and_alpha(wo, C, taps, center, compute_alpha, mean, red);
  if (tid == 0 && alpha_out) alpha_out[o] = alpha;
  for (int k = tid, t = tid % taps; k < K; k += xt::NT, t = (t + xt::NT) % taps)
    what[(size_t)o * K + k] = xt::sgn(wo[k] - mean[t]) * alpha;
}

// dwhat: `splits` partial slabs [splits][O][C][taps] (the split-K partial sums of the weight-gradient kernel, added here
// in slab order — one pass less than summing them first), or the gradient itself with splits == 1.
__global__ __launch_bounds__(xt::NT) void xnor_weight_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dwhat,
                                                                 int splits, size_t slab, int C, int taps, int center,
                                                                 int compute_alpha, float* __restrict__ dw) {
  __shared__ float mean[xt::kMaxTaps];
  __shared__ float dmean[xt::kMaxTaps];
  __shared__ double red[4];
  const int o = blockIdx.x, tid = threadIdx.x, K = C * taps;
  const float* wo = w + (size_t)o * K;
  float* dwo = dw + (size_t)o * K;
  // dL/dWhat of this channel, summed over the slabs, parked in dw (every element is read back by the thread that wrote
  // it, or behind a barrier)
  for (int k = tid; k < K; k += xt::NT) {
    const float* p = dwhat + (size_t)o * K + k;
    float a = p[0];
    for (int s = 1; s < splits; ++s) a += p[(size_t)s * slab];
    dwo[k] = a;
  }
  const float alpha = xt::centre_and_alpha(wo, C, taps, center, compute_alpha, mean, red);   // (barriers inside)
  float gk = 0.0f;      // (sum dWhat * sign(Wc)) / K: the gradient that reaches Wc through alpha
  if (compute_alpha) {
    double s = 0.0;
    for (int k = tid, t = tid % taps; k < K; k += xt::NT, t = (t + xt::NT) % taps)
      s += (double)dwo[k] * (double)xt::sgn(wo[k] - mean[t]);
    gk = (float)(xt::block_sum(s, red) / (double)K);
  }
  auto dvc = [&](int k, int t) {
    const float v = wo[k] - mean[t];
    const float ste = fabsf(v) < 1.0f ? dwo[k] * alpha : 0.0f;
    return ste + xt::sgn(v) * gk;
  };
  if (center) {      // dW = dWc - mean over the input channels of dWc, per tap
    for (int t = 0; t < taps; ++t) {
      double s = 0.0;
      for (int c = tid; c < C; c += xt::NT) s += (double)dvc(c * taps + t, t);
      const float m = (float)(xt::block_sum(s, red) / (double)C);
      if (tid == 0) dmean[t] = m;
    }
    __syncthreads();
  }
  for (int k = tid, t = tid % taps; k < K; k += xt::NT, t = (t + xt::NT) % taps)
    dwo[k] = dvc(k, t) - (center ? dmean[t] : 0.0f);
}

// What the input-gradient kernel needs of the weights, straight from W (round 5; was xnor_what_kernel -> grad_alpha_kernel
// -> grad_pack_weight_kernel: three launches per layer and backward pass, 57 per ResNet-18 step): alpha[o] and
// sign(Wc) in the MFMA B-fragment order of csrc/grad.hip,
//     Bp[ob][tap][cs][lane][e] = sign(Wc[o = 32 ob + 8 (lane >> 4) + e][c = 16 cs + (lane & 15)][tap])     (bf16)
// One workgroup per output channel of the PADDED range (32 * ceil(O / 32)): channels past O and input channels past C
// are written as zeros, so the buffer needs no clearing.  The same centring / alpha code as xnor_what_kernel: the same
// bits as the three-launch form for finite weights (alpha there: max |What| = alpha, or 0 when every sign is 0 — see
// below; a NaN weight made that form's What NaN throughout, i.e. all signs 0 and alpha 0 — here the channel keeps the
// signs of its finite weights and alpha = NaN reaches the input gradient).
__global__ __launch_bounds__(xt::NT) void xnor_grad_pack_kernel(const float* __restrict__ w, int O, int C, int taps, int center,
                                                                int compute_alpha, int CS, unsigned short* __restrict__ Bp,
                                                                float* __restrict__ alpha_out) {
  __shared__ float mean[xt::kMaxTaps];
  __shared__ double red[4];
  __shared__ int any_nz;
  const int o = blockIdx.x, tid = threadIdx.x, K = C * taps;
  const int ob = o >> 5, lg = (o & 31) >> 3, e = o & 7;
  const bool live = o < O;
  float alpha = 0.0f;
  if (tid == 0) any_nz = 0;
  const float* wo = w + (size_t)(live ? o : 0) * K;
  if (live) alpha = xt::centre_and_alpha(wo, C, taps, center, compute_alpha, mean, red);   // (block-uniform branch)
  else __syncthreads();
  int nz = 0;
  const int KP = CS * 16 * taps;                  // the channel range padded to whole 16-channel fragments
  for (int k = tid; k < KP; k += xt::NT) {
    const int c = k / taps, t = k - c * taps;
    unsigned short b = 0;
    if (live && c < C) {
      const float v = wo[(size_t)c * taps + t] - mean[t];
      b = is_pos(v) ? (unsigned short)0x3F80 : is_neg(v) ? (unsigned short)0xBF80 : (unsigned short)0;
      nz |= b != 0;
    }
    const size_t blk = ((size_t)ob * taps + t) * CS + (c >> 4);
    Bp[(blk * 64 + (c & 15) + 16 * lg) * 8 + e] = b;
  }
  if (nz) any_nz = 1;                             // (benign race: every writer stores 1)
  __syncthreads();
  // the three-launch form derived alpha as max |What|: 0 for a channel whose signs are all 0
  if (live && tid == 0) alpha_out[o] = any_nz ? alpha : 0.0f;
}

int launch_xnor_grad_pack(const float* w, int O, int C, int ks, int center, int compute_alpha, void* packed, float* alpha,
                          hipStream_t s) {
  const int taps = ks * ks, CS = (C + 15) / 16, OB = (O + 31) / 32;
  hipLaunchKernelGGL(xnor_grad_pack_kernel, dim3((unsigned)(OB * 32)), dim3(xt::NT), 0, s, w, O, C, taps, center, compute_alpha,
                     CS, static_cast<unsigned short*>(packed), alpha);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_xnor_what(const float* w, int O, int C, int taps, int center, int compute_alpha, float* what, float* alpha,
                     hipStream_t s) {
  if (taps > xt::kMaxTaps) return BNN_HIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(xnor_what_kernel, dim3((unsigned)O), dim3(xt::NT), 0, s, w, C, taps, center, compute_alpha, what, alpha);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_xnor_weight_bwd(const float* w, const float* dwhat, int splits, int O, int C, int taps, int center,
                           int compute_alpha, float* dw, hipStream_t s) {
  if (taps > xt::kMaxTaps) return BNN_HIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(xnor_weight_bwd_kernel, dim3((unsigned)O), dim3(xt::NT), 0, s, w, dwhat, splits,
                     (size_t)O * C * taps, C, taps, center, compute_alpha, dw);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
