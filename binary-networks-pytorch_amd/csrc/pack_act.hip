// pack_act.hip — sign() of an fp32 NCHW activation tensor as two pixel-major bit planes.
//
// Replaces bnn/ops.py:63-66 (SignActivation.forward: `input.sign()`) as called by
// bnn/ops.py:151-152 (BasicInputBinarizer.forward).  The reference materialises a full
// fp32 tensor of {-1,0,+1}; here the same information is 2 bits per element:
//   P bit = x > 0, M bit = x < 0, neither = 0 / -0 / NaN  (torch.sign semantics).
//
// HBM-bound: reads 4 B/element, writes 2 bits/element (+2 B/pixel of nzc).
// Mapping: one thread owns VP consecutive pixels of one image and walks all channels,
// so every global load is a coalesced 4*VP-byte-per-lane row segment of one channel
// plane, and every store is a whole uint64 word of a pixel.
#include "bnn_dev.h"

namespace bnn {

template <int VP>
struct PixVec;
template <>
struct PixVec<1> { using type = float; };
template <>
struct PixVec<2> { using type = float2; };
template <>
struct PixVec<4> { using type = float4; };

template <int VP>
__global__ __launch_bounds__(256) void pack_act_kernel(const float* __restrict__ x, int C, int HW,
                                                       long long npix, int cw64,
                                                       uint64_t* __restrict__ P,
                                                       uint64_t* __restrict__ M,
                                                       uint16_t* __restrict__ nzc) {
  using V = typename PixVec<VP>::type;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long pix0 = t * VP;
  if (pix0 >= npix) return;
  const int n = (int)(pix0 / HW);
  const int r = (int)(pix0 - (long long)n * HW);
  const float* xb = x + ((size_t)n * C) * HW + r;

  int cnt[VP];
#pragma unroll
  for (int v = 0; v < VP; ++v) cnt[v] = 0;

  for (int g = 0; g < cw64; ++g) {
    uint32_t pw[2][VP], mw[2][VP];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int v = 0; v < VP; ++v) { pw[h][v] = 0u; mw[h][v] = 0u; }
      const int c0 = g * 64 + h * 32;
      if (c0 + 32 <= C) {
        // full 32-channel word: walk channels high -> low so that shifting left leaves
        // channel c0+b in bit b.
#pragma unroll 8
        for (int b = 31; b >= 0; --b) {
          const V xv = *reinterpret_cast<const V*>(xb + (size_t)(c0 + b) * HW);
          const float* xs = reinterpret_cast<const float*>(&xv);
#pragma unroll
          for (int v = 0; v < VP; ++v) {
            pw[h][v] = (pw[h][v] << 1) | (is_pos(xs[v]) ? 1u : 0u);
            mw[h][v] = (mw[h][v] << 1) | (is_neg(xs[v]) ? 1u : 0u);
          }
        }
      } else {
        for (int b = 0; b < 32 && c0 + b < C; ++b) {
          const V xv = *reinterpret_cast<const V*>(xb + (size_t)(c0 + b) * HW);
          const float* xs = reinterpret_cast<const float*>(&xv);
#pragma unroll
          for (int v = 0; v < VP; ++v) {
            pw[h][v] |= (is_pos(xs[v]) ? 1u : 0u) << b;
            mw[h][v] |= (is_neg(xs[v]) ? 1u : 0u) << b;
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VP; ++v) {
      const uint64_t pq = (uint64_t)pw[0][v] | ((uint64_t)pw[1][v] << 32);
      const uint64_t mq = (uint64_t)mw[0][v] | ((uint64_t)mw[1][v] << 32);
      P[(size_t)(pix0 + v) * cw64 + g] = pq;
      M[(size_t)(pix0 + v) * cw64 + g] = mq;
      cnt[v] += __builtin_popcountll(pq | mq);
    }
  }
#pragma unroll
  for (int v = 0; v < VP; ++v) nzc[pix0 + v] = (uint16_t)cnt[v];
}

int launch_pack_act(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                    uint16_t* nzc, hipStream_t stream) {
  const int HW = H * W;
  const long long npix = (long long)N * HW;
  const int cw64 = (C + 63) / 64;
  const bool a16 = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  const bool a8 = (reinterpret_cast<uintptr_t>(x) & 7u) == 0;
  if (HW % 4 == 0 && a16) {
    const long long nthr = npix / 4;
    hipLaunchKernelGGL(pack_act_kernel<4>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0,
                       stream, x, C, HW, npix, cw64, P, M, nzc);
  } else if (HW % 2 == 0 && a8) {
    const long long nthr = npix / 2;
    hipLaunchKernelGGL(pack_act_kernel<2>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0,
                       stream, x, C, HW, npix, cw64, P, M, nzc);
  } else {
    hipLaunchKernelGGL(pack_act_kernel<1>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                       stream, x, C, HW, npix, cw64, P, M, nzc);
  }
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
