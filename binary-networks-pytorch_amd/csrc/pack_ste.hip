// pack_ste.hip — what the backward of a binary convolution needs from its input, in 3 bits per element.
//
// Reference: the autograd graph of bnn/layers/conv.py:90-97 keeps the fp32 input x alive for the backward of
// SignActivation (bnn/ops.py:68-73: grad * 1[|x| < 1]) and the fp32 sign(x) for the weight gradient.  Both are
// functions of three bits per element:
//     P = x > 0,  M = x < 0        (sign(x): the planes bnn_hip_pack_act_f32 writes, same format)
//     T = |x| < 1                  (the hard-tanh straight-through mask; NaN -> 0, like masked_fill(x.abs() >= 1, 0))
// written here in ONE pass over x as three [N][ceil(C/64)][H][W] uint64 planes — 32 -> 3 bits per element of saved
// state for every binary layer of a training step (csrc/grad.hip reads them: dgrad the T plane, wgrad P and M).
// HBM-bound: 4 B read + 3 bits written per element.  Same thread mapping as pack_act.hip.
#include "bnn_dev.h"

namespace bnn {

template <int VP>
__global__ __launch_bounds__(256) void pack_ste_kernel(const float* __restrict__ x, int C, int HW, long long npix,
                                                       int cw64, uint64_t* __restrict__ P, uint64_t* __restrict__ M,
                                                       uint64_t* __restrict__ T) {
  struct alignas(4 * VP) V { float v[VP]; };
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long pix0 = t * VP;
  if (pix0 >= npix) return;
  const int g = blockIdx.y;
  const int n = (int)(pix0 / HW);
  const int r = (int)(pix0 - (long long)n * HW);
  const float* xb = x + ((size_t)n * C) * HW + r;
  uint32_t pw[2][VP], mw[2][VP], tw[2][VP];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int v = 0; v < VP; ++v) { pw[h][v] = 0u; mw[h][v] = 0u; tw[h][v] = 0u; }
    const int c0 = g * 64 + h * 32;
    for (int b = 0; b < 32 && c0 + b < C; ++b) {
      const V xv = *reinterpret_cast<const V*>(xb + (size_t)(c0 + b) * HW);
#pragma unroll
      for (int v = 0; v < VP; ++v) {
        const float u = xv.v[v];
        pw[h][v] |= (is_pos(u) ? 1u : 0u) << b;
        mw[h][v] |= (is_neg(u) ? 1u : 0u) << b;
        tw[h][v] |= (fabsf(u) < 1.0f ? 1u : 0u) << b;
      }
    }
  }
  const size_t o = ((size_t)n * cw64 + g) * HW + r;
#pragma unroll
  for (int v = 0; v < VP; ++v) {
    P[o + v] = (uint64_t)pw[0][v] | ((uint64_t)pw[1][v] << 32);
    M[o + v] = (uint64_t)mw[0][v] | ((uint64_t)mw[1][v] << 32);
    T[o + v] = (uint64_t)tw[0][v] | ((uint64_t)tw[1][v] << 32);
  }
}

int launch_pack_ste(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M, uint64_t* T,
                    hipStream_t stream) {
  const int HW = H * W;
  const long long npix = (long long)N * HW;
  const int cw64 = (C + 63) / 64;
  const bool a4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0, a2 = (reinterpret_cast<uintptr_t>(x) & 7) == 0;
  auto grid = [&](long long nthr) { return dim3((unsigned)((nthr + 255) / 256), (unsigned)cw64); };
  if (HW % 4 == 0 && a4)
    hipLaunchKernelGGL(pack_ste_kernel<4>, grid(npix / 4), dim3(256), 0, stream, x, C, HW, npix, cw64, P, M, T);
  else if (HW % 2 == 0 && a2)
    hipLaunchKernelGGL(pack_ste_kernel<2>, grid(npix / 2), dim3(256), 0, stream, x, C, HW, npix, cw64, P, M, T);
  else
    hipLaunchKernelGGL(pack_ste_kernel<1>, grid(npix), dim3(256), 0, stream, x, C, HW, npix, cw64, P, M, T);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
