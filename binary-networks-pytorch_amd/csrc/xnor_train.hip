// xnor_train.hip — the XNOR-Net weight binarizer of a training step, forward value and backward, one kernel each.
//
// Reference: XNORWeightBinarizer.forward (bnn/ops.py:129-140) evaluated under autograd,
//     Wc = W - mean_c(W)            (center_weights)          alpha = mean |Wc| per output channel (bnn/ops.py:116-127)
//     What = SignActivation(Wc) * alpha                        SignActivation.backward: grad * 1[|Wc| < 1] (bnn/ops.py:68-73)
// which torch runs as ~6 element-wise / reduction kernels forward and ~8 backward per layer (19 layers: 1.2 ms of a
// 21 ms ResNet-18 step).  Here: one wave per output channel, reductions in double with the fixed order of
// pack_weight.hip (lane-strided partial sums + xor butterfly), so What carries the same alpha the forward kernels use.
//     forward :  What[o,c,t] = sign(Wc) * alpha[o]
//     backward:  dWc = dWhat * alpha * 1[|Wc| < 1] + sign(Wc) * (sum_{c,t} dWhat * sign(Wc)) / K        (compute_alpha)
//                dWc = dWhat * 1[|Wc| < 1]                                                              (otherwise)
//                dW  = dWc - mean_c(dWc)  when centred, else dWc
#include "bnn_dev.h"

namespace bnn {

namespace xt {
constexpr int kMaxTaps = 1024;
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
__device__ __forceinline__ float sgn(float v) { return is_pos(v) ? 1.0f : is_neg(v) ? -1.0f : 0.0f; }

// mean[t] over the input channels (0 when not centred) into LDS; returns alpha
__device__ __forceinline__ float centre_and_alpha(const float* __restrict__ wo, int C, int taps, int center,
                                                  int compute_alpha, float* mean) {
  const int lane = threadIdx.x, K = C * taps;
  for (int t = 0; t < taps; ++t) {
    float m = 0.0f;
    if (center) {
      double s = 0.0;
      for (int c = lane; c < C; c += kWave) s += (double)wo[(size_t)c * taps + t];
      m = (float)(wave_sum(s) / (double)C);
    }
    if (lane == 0) mean[t] = m;
  }
  __syncthreads();
  if (!compute_alpha) return 1.0f;
  double s = 0.0;
  for (int k = lane; k < K; k += kWave) s += (double)fabsf(wo[k] - mean[k % taps]);
  return (float)(wave_sum(s) / (double)K);
}
}  // namespace xt

__global__ __launch_bounds__(64) void xnor_what_kernel(const float* __restrict__ w, int C, int taps, int center,
                                                       int compute_alpha, float* __restrict__ what,
                                                       float* __restrict__ alpha_out) {
  __shared__ float mean[xt::kMaxTaps];
  const int o = blockIdx.x, lane = threadIdx.x, K = C * taps;
  const float* wo = w + (size_t)o * K;
  const float alpha = xt::centre_and_alpha(wo, C, taps, center, compute_alpha, mean);
  if (lane == 0 && alpha_out) alpha_out[o] = alpha;
  for (int k = lane; k < K; k += kWave) what[(size_t)o * K + k] = xt::sgn(wo[k] - mean[k % taps]) * alpha;
}

__global__ __launch_bounds__(64) void xnor_weight_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dwhat,
                                                             int C, int taps, int center, int compute_alpha,
                                                             float* __restrict__ dw) {
  __shared__ float mean[xt::kMaxTaps];
  __shared__ float dmean[xt::kMaxTaps];
  const int o = blockIdx.x, lane = threadIdx.x, K = C * taps;
  const float* wo = w + (size_t)o * K;
  const float* go = dwhat + (size_t)o * K;
  float* dwo = dw + (size_t)o * K;
  const float alpha = xt::centre_and_alpha(wo, C, taps, center, compute_alpha, mean);
  float gk = 0.0f;      // (sum dWhat * sign(Wc)) / K: the gradient that reaches Wc through alpha
  if (compute_alpha) {
    double s = 0.0;
    for (int k = lane; k < K; k += kWave) s += (double)go[k] * (double)xt::sgn(wo[k] - mean[k % taps]);
    gk = (float)(xt::wave_sum(s) / (double)K);
  }
  auto dvc = [&](int k) {
    const float v = wo[k] - mean[k % taps];
    const float ste = fabsf(v) < 1.0f ? go[k] * alpha : 0.0f;
    return ste + xt::sgn(v) * gk;
  };
  if (center) {      // dW = dWc - mean over the input channels of dWc, per tap
    for (int t = 0; t < taps; ++t) {
      double s = 0.0;
      for (int c = lane; c < C; c += kWave) s += (double)dvc(c * taps + t);
      const float m = (float)(xt::wave_sum(s) / (double)C);
      if (lane == 0) dmean[t] = m;
    }
    __syncthreads();
  }
  for (int k = lane; k < K; k += kWave) dwo[k] = dvc(k) - (center ? dmean[k % taps] : 0.0f);
}

int launch_xnor_what(const float* w, int O, int C, int taps, int center, int compute_alpha, float* what, float* alpha,
                     hipStream_t s) {
  if (taps > xt::kMaxTaps) return BNN_HIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(xnor_what_kernel, dim3((unsigned)O), dim3(64), 0, s, w, C, taps, center, compute_alpha, what, alpha);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_xnor_weight_bwd(const float* w, const float* dwhat, int O, int C, int taps, int center, int compute_alpha,
                           float* dw, hipStream_t s) {
  if (taps > xt::kMaxTaps) return BNN_HIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(xnor_weight_bwd_kernel, dim3((unsigned)O), dim3(64), 0, s, w, dwhat, C, taps, center, compute_alpha, dw);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
