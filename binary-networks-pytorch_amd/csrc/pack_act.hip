// pack_act.hip — sign() of an fp32 NCHW activation tensor as two pixel-major bit planes.
//
// Replaces bnn/ops.py:63-66 (SignActivation.forward: `input.sign()`) as called by
// bnn/ops.py:151-152 (BasicInputBinarizer.forward).  The reference materialises a full
// fp32 tensor of {-1,0,+1}; here the same information is 2 bits per element:
//   P bit = x > 0, M bit = x < 0, neither = 0 / -0 / NaN  (torch.sign semantics).
//
// HBM-bound: reads 4 B/element, writes 2 bits/element.
// Mapping: one thread owns VP consecutive pixels of one image and ONE 64-channel group
// (blockIdx.y), so every global load is a coalesced 4*VP-byte-per-lane row segment of one
// channel plane, every store is VP contiguous uint64 words of the [n][group][y][x] output
// plane, and small images with many channels still fill the chip (parallelism = pixels/VP x
// channel groups).
#include <hip/hip_fp16.h>

#include "bnn_dev.h"

namespace bnn {

// VP consecutive pixels of one channel, loaded with ONE instruction (4..16 bytes per lane).
// T = float, or __half for `.half()` models (sign() of an fp16 value is the sign of its exact fp32 widening).
template <typename T, int VP>
struct alignas(sizeof(T) * VP) PixVec {
  T v[VP];
};
// The activation tensor is read exactly once: non-temporal loads (no L2 / MALL allocation).  Measured on the config-2
// tensor (411 MB, warm clocks): 79-82 us = 5.1 TB/s with plain loads, 68.8 us = 5.97 TB/s non-temporal.
template <typename V>
__device__ __forceinline__ V load_once(const V* p) {
  if constexpr (sizeof(V) == 16) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const u4*>(p)));
  } else if constexpr (sizeof(V) == 8) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const u2*>(p)));
  } else if constexpr (sizeof(V) == 4) {
    return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p)));
  } else {
    return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p)));
  }
}
__device__ __forceinline__ float widen(float v) { return v; }
__device__ __forceinline__ float widen(__half v) { return __half2float(v); }

// AFF: v = fmaf(x, bn_a[c], bn_b[c]) first (eval-mode BatchNorm in front of a pre-activation block's
// binary conv: res_block.py:148, hierarchical_block.py:39); relu != 0: planes of sign(max(v, 0)).
template <int VP, bool AFF = false, typename T = float>
__global__ __launch_bounds__(256) void pack_act_kernel(const T* __restrict__ x, int C, int HW,
                                                       long long npix, int cw64,
                                                       uint64_t* __restrict__ P,
                                                       uint64_t* __restrict__ M,
                                                       const float* __restrict__ bn_a = nullptr,
                                                       const float* __restrict__ bn_b = nullptr,
                                                       int relu = 0) {
  using V = PixVec<T, VP>;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long pix0 = t * VP;
  if (pix0 >= npix) return;
  const int g = blockIdx.y;
  const int n = (int)(pix0 / HW);
  const int r = (int)(pix0 - (long long)n * HW);
  const T* xb = x + ((size_t)n * C) * HW + r;

  uint32_t pw[2][VP], mw[2][VP];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int v = 0; v < VP; ++v) { pw[h][v] = 0u; mw[h][v] = 0u; }
    const int c0 = g * 64 + h * 32;
    if (c0 + 32 <= C) {
      // full 32-channel word: walk channels high -> low so that shifting left leaves
      // channel c0+b in bit b.
#pragma unroll 16
      for (int b = 31; b >= 0; --b) {
        const V xv = load_once(reinterpret_cast<const V*>(xb + (size_t)(c0 + b) * HW));
        float xs[VP];
#pragma unroll
        for (int v = 0; v < VP; ++v) xs[v] = widen(xv.v[v]);
        const float ca = AFF && bn_a ? bn_a[c0 + b] : 1.0f, cb = AFF && bn_a ? bn_b[c0 + b] : 0.0f;
#pragma unroll
        for (int v = 0; v < VP; ++v) {
          const float u = (AFF && bn_a) ? fmaf(xs[v], ca, cb) : xs[v];
          pw[h][v] = (pw[h][v] << 1) | (is_pos(u) ? 1u : 0u);
          mw[h][v] = (mw[h][v] << 1) | ((!(AFF && relu) && is_neg(u)) ? 1u : 0u);
        }
      }
    } else {
      for (int b = 0; b < 32 && c0 + b < C; ++b) {
        const V xv = load_once(reinterpret_cast<const V*>(xb + (size_t)(c0 + b) * HW));
        float xs[VP];
#pragma unroll
        for (int v = 0; v < VP; ++v) xs[v] = widen(xv.v[v]);
        const float ca = AFF && bn_a ? bn_a[c0 + b] : 1.0f, cb = AFF && bn_a ? bn_b[c0 + b] : 0.0f;
#pragma unroll
        for (int v = 0; v < VP; ++v) {
          const float u = (AFF && bn_a) ? fmaf(xs[v], ca, cb) : xs[v];
          pw[h][v] |= (is_pos(u) ? 1u : 0u) << b;
          mw[h][v] |= ((!(AFF && relu) && is_neg(u)) ? 1u : 0u) << b;
        }
      }
    }
  }
  // planes are [n][group][y][x] uint64: the VP words of this thread are contiguous
  const size_t o = ((size_t)n * cw64 + g) * HW + r;
#pragma unroll
  for (int v = 0; v < VP; ++v) {
    P[o + v] = (uint64_t)pw[0][v] | ((uint64_t)pw[1][v] << 32);
    M[o + v] = (uint64_t)mw[0][v] | ((uint64_t)mw[1][v] << 32);
  }
}

template <typename T>
static int launch_pack_act_t(const T* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                             hipStream_t stream) {
  const int HW = H * W;
  const long long npix = (long long)N * HW;
  const int cw64 = (C + 63) / 64;
  const bool a4 = (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(T) - 1)) == 0;
  const bool a2 = (reinterpret_cast<uintptr_t>(x) & (2 * sizeof(T) - 1)) == 0;
  auto grid = [&](long long nthr) { return dim3((unsigned)((nthr + 255) / 256), (unsigned)cw64); };
  const float* none = nullptr;
  if (HW % 4 == 0 && a4)
    hipLaunchKernelGGL((pack_act_kernel<4, false, T>), grid(npix / 4), dim3(256), 0, stream, x, C, HW, npix, cw64,
                       P, M, none, none, 0);
  else if (HW % 2 == 0 && a2)
    hipLaunchKernelGGL((pack_act_kernel<2, false, T>), grid(npix / 2), dim3(256), 0, stream, x, C, HW, npix, cw64,
                       P, M, none, none, 0);
  else
    hipLaunchKernelGGL((pack_act_kernel<1, false, T>), grid(npix), dim3(256), 0, stream, x, C, HW, npix, cw64, P, M,
                       none, none, 0);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_pack_act(const float* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                    hipStream_t stream) {
  return launch_pack_act_t<float>(x, N, C, H, W, P, M, stream);
}

int launch_pack_act_f16(const void* x, int N, int C, int H, int W, uint64_t* P, uint64_t* M,
                        hipStream_t stream) {
  return launch_pack_act_t<__half>(static_cast<const __half*>(x), N, C, H, W, P, M, stream);
}

int launch_bn_act_pack(const float* x, int N, int C, int H, int W, const float* bn_a, const float* bn_b,
                       int relu, uint64_t* P, uint64_t* M, hipStream_t stream) {
  const int HW = H * W;
  const long long npix = (long long)N * HW;
  const int cw64 = (C + 63) / 64;
  const bool a16 = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  const bool a8 = (reinterpret_cast<uintptr_t>(x) & 7u) == 0;
  auto grid = [&](long long nthr) { return dim3((unsigned)((nthr + 255) / 256), (unsigned)cw64); };
  if (HW % 4 == 0 && a16)
    hipLaunchKernelGGL((pack_act_kernel<4, true>), grid(npix / 4), dim3(256), 0, stream, x, C, HW, npix, cw64,
                       P, M, bn_a, bn_b, relu);
  else if (HW % 2 == 0 && a8)
    hipLaunchKernelGGL((pack_act_kernel<2, true>), grid(npix / 2), dim3(256), 0, stream, x, C, HW, npix, cw64,
                       P, M, bn_a, bn_b, relu);
  else
    hipLaunchKernelGGL((pack_act_kernel<1, true>), grid(npix), dim3(256), 0, stream, x, C, HW, npix, cw64, P,
                       M, bn_a, bn_b, relu);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// AvgPool2d(k, stride=k, ceil_mode=True, count_include_pad=False) + sign() in one pass: the
// shortcut branch of a down-sampling stage (bnn/models/resnet.py:128-133).  The sign of an
// average is the sign of the sum, so the divisor never matters and clipped windows at the
// ragged edge need no special case beyond skipping out-of-image taps.
// One thread = one output pixel x one 32-channel word.
__global__ __launch_bounds__(256) void avgpool_pack_kernel(const float* __restrict__ x, int C,
                                                           int H, int W, int k, int Ho, int Wo,
                                                           long long npix_out, int cw32,
                                                           uint32_t* __restrict__ P,
                                                           uint32_t* __restrict__ M) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= npix_out) return;
  const int word = blockIdx.y;
  const int hw = Ho * Wo;
  const int n = (int)(q / hw);
  const int r = (int)(q - (long long)n * hw);
  const int oy = r / Wo, ox = r - oy * Wo;
  uint32_t pw = 0u, mw = 0u;
  for (int b = 0; b < 32; ++b) {
    const int c = word * 32 + b;
    if (c >= C) break;
    const float* xc = x + ((size_t)n * C + c) * H * W;
    float s = 0.0f;
    for (int dy = 0; dy < k; ++dy) {
      const int iy = oy * k + dy;
      if (iy >= H) break;
      for (int dx = 0; dx < k; ++dx) {
        const int ix = ox * k + dx;
        if (ix >= W) break;
        s += xc[(size_t)iy * W + ix];
      }
    }
    pw |= (is_pos(s) ? 1u : 0u) << b;
    mw |= (is_neg(s) ? 1u : 0u) << b;
  }
  const size_t o = ((((size_t)n * (cw32 >> 1) + (word >> 1)) * hw + r) << 1) + (word & 1);
  P[o] = pw;
  M[o] = mw;
}

// The same shortcut input when the tensor is NON-NEGATIVE (a ReLU output — every ResNet stage transition) and its
// sign planes already exist: an average of non-negative values is positive iff one of them is, so
//     sign(AvgPool_k(x)) = OR over the k x k window of the P plane,   M = 0,
// computed from 2 bits per element instead of re-reading the fp32 tensor (206 MB -> 6.4 MB for the 64-channel
// 56x56 stage at batch 256: 38 us -> launch-bound).  Exact for finite inputs (a sum of non-negative floats is
// positive iff a term is); a NaN element makes the reference's average NaN (sign 0) while the OR ignores it.
__global__ __launch_bounds__(256) void orpool_packed_kernel(const uint64_t* __restrict__ P, int H, int W, int k,
                                                            int Ho, int Wo, long long nwords,
                                                            uint64_t* __restrict__ outP,
                                                            uint64_t* __restrict__ outM) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // (n * cw64 + g, oy, ox)
  if (q >= nwords) return;
  const int hw = Ho * Wo;
  const long long plane = q / hw;
  const int r = (int)(q - plane * hw);
  const int oy = r / Wo, ox = r - oy * Wo;
  const uint64_t* src = P + plane * H * W;
  uint64_t acc = 0;
  for (int dy = 0; dy < k; ++dy) {
    const int iy = oy * k + dy;
    if (iy >= H) break;
    for (int dx = 0; dx < k; ++dx) {
      const int ix = ox * k + dx;
      if (ix >= W) break;
      acc |= src[(size_t)iy * W + ix];
    }
  }
  outP[q] = acc;
  outM[q] = 0;
}

int launch_orpool_packed(const uint64_t* P, int N, int C, int H, int W, int k, uint64_t* outP, uint64_t* outM,
                         hipStream_t stream) {
  const int Ho = (H + k - 1) / k, Wo = (W + k - 1) / k;
  const long long nwords = (long long)N * ((C + 63) / 64) * Ho * Wo;
  hipLaunchKernelGGL(orpool_packed_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, stream, P, H, W, k,
                     Ho, Wo, nwords, outP, outM);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// k == 2 on even images with W % 4 == 0 (every ResNet stage transition): one thread produces two
// adjacent outputs from one aligned float4 per input row.  Same tap order as the scalar kernel.
__global__ __launch_bounds__(256) void avgpool2_pack_kernel(const float* __restrict__ x, int C,
                                                            int H, int W, int Ho, int Wo,
                                                            long long npairs, int cw32,
                                                            uint32_t* __restrict__ P,
                                                            uint32_t* __restrict__ M) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // (n, oy, ox/2)
  if (q >= npairs) return;
  const int word = blockIdx.y;
  const int wp = Wo >> 1;
  const int n = (int)(q / ((long long)Ho * wp));
  const int rr = (int)(q - (long long)n * Ho * wp);
  const int oy = rr / wp, t = rr - oy * wp;
  const int hw = Ho * Wo;
  const int r = oy * Wo + 2 * t;
  uint32_t pw0 = 0u, mw0 = 0u, pw1 = 0u, mw1 = 0u;
  for (int b = 0; b < 32; ++b) {
    const int c = word * 32 + b;
    if (c >= C) break;
    const float* row = x + (((size_t)n * C + c) * H + 2 * oy) * W + 4 * t;
    const float4 u = *reinterpret_cast<const float4*>(row);
    const float4 v = *reinterpret_cast<const float4*>(row + W);
    const float s0 = ((u.x + u.y) + v.x) + v.y;
    const float s1 = ((u.z + u.w) + v.z) + v.w;
    pw0 |= (is_pos(s0) ? 1u : 0u) << b;
    mw0 |= (is_neg(s0) ? 1u : 0u) << b;
    pw1 |= (is_pos(s1) ? 1u : 0u) << b;
    mw1 |= (is_neg(s1) ? 1u : 0u) << b;
  }
  const size_t o = ((((size_t)n * (cw32 >> 1) + (word >> 1)) * hw + r) << 1) + (word & 1);
  P[o] = pw0; M[o] = mw0;
  P[o + 2] = pw1; M[o + 2] = mw1;
}

// k == 2 on even images whose width is not a multiple of 4 (14x14 -> 7x7): one aligned float2 per
// input row and output.  Same tap order as the scalar kernel.
__global__ __launch_bounds__(256) void avgpool2_pack_f2_kernel(const float* __restrict__ x, int C,
                                                               int H, int W, int Ho, int Wo,
                                                               long long npix_out, int cw32,
                                                               uint32_t* __restrict__ P,
                                                               uint32_t* __restrict__ M) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= npix_out) return;
  const int word = blockIdx.y;
  const int hw = Ho * Wo;
  const int n = (int)(q / hw);
  const int r = (int)(q - (long long)n * hw);
  const int oy = r / Wo, ox = r - oy * Wo;
  uint32_t pw = 0u, mw = 0u;
  const float* base = x + (((size_t)n * C + (size_t)word * 32) * H + 2 * oy) * W + 2 * ox;
  const size_t cstride = (size_t)H * W;
#pragma unroll 8
  for (int b = 0; b < 32; ++b) {
    if (word * 32 + b >= C) break;
    const float2 u = *reinterpret_cast<const float2*>(base + b * cstride);
    const float2 v = *reinterpret_cast<const float2*>(base + b * cstride + W);
    const float s = ((u.x + u.y) + v.x) + v.y;
    pw |= (is_pos(s) ? 1u : 0u) << b;
    mw |= (is_neg(s) ? 1u : 0u) << b;
  }
  const size_t o = ((((size_t)n * (cw32 >> 1) + (word >> 1)) * hw + r) << 1) + (word & 1);
  P[o] = pw;
  M[o] = mw;
}

int launch_avgpool_pack(const float* x, int N, int C, int H, int W, int k, uint64_t* P, uint64_t* M,
                        hipStream_t stream) {
  const int Ho = (H + k - 1) / k, Wo = (W + k - 1) / k;
  const long long npix = (long long)N * Ho * Wo;
  const int cw32 = 2 * ((C + 63) / 64);
  if (k == 2 && H % 2 == 0 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
    const long long npairs = npix / 2;
    hipLaunchKernelGGL(avgpool2_pack_kernel, dim3((unsigned)((npairs + 255) / 256), (unsigned)cw32),
                       dim3(256), 0, stream, x, C, H, W, Ho, Wo, npairs, cw32,
                       reinterpret_cast<uint32_t*>(P), reinterpret_cast<uint32_t*>(M));
    return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  }
  if (k == 2 && H % 2 == 0 && W % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0) {
    hipLaunchKernelGGL(avgpool2_pack_f2_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)cw32),
                       dim3(256), 0, stream, x, C, H, W, Ho, Wo, npix, cw32,
                       reinterpret_cast<uint32_t*>(P), reinterpret_cast<uint32_t*>(M));
    return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(avgpool_pack_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)cw32),
                     dim3(256), 0, stream, x, C, H, W, k, Ho, Wo, npix, cw32,
                     reinterpret_cast<uint32_t*>(P), reinterpret_cast<uint32_t*>(M));
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// AvgPool2d(2, 2) in front of a pre-activation stage + the sign planes of up to TWO BatchNorm branches of its result in one
// pass (a hierarchical-block stage behind a pool, bnn_amd/models/resnet.py: the block's bn1 -> act -> conv1 and its
// shortcut's bn -> conv1x1 both binarise the pooled tensor):
//     t  = (((x00 + x01) + x10) + x11) / 4        ATen's avg_pool2d: window summed row by row, then divided
//     u1 = fmaf(t, a1[c], b1[c]) -> P1 / M1 (relu1: M1 = 0);   u2 = fmaf(t, a2[c], b2[c]) -> P2 / M2
// The fp32 tensor t is written only when `out` is given (nobody reads it when both consumers take planes).
// A workgroup = 64 consecutive output pixels x one 32-channel word; wave w of it = channels 8w .. 8w+7 of the word (lane =
// pixel: coalesced float2 loads, 16 in flight per lane); the four 8-bit pieces meet in LDS.  Even H and W.
// (large tensors — the 56x56 -> 28x28 transition at batch 128 — stream at 4.9 TB/s in the one-thread-per-word form below;
// this one is for the smaller transitions, where that form has too few threads: 19.4 -> 14.3 us and 18.4 -> 11.0 us)
__global__ __launch_bounds__(256) void avgpool2_bn_pack2_kernel(
    const float* __restrict__ x, int C, int H, int W, int Ho, int Wo, long long npix_out, int cw32,
    const float* __restrict__ a1, const float* __restrict__ b1, int relu1, uint32_t* __restrict__ P1,
    uint32_t* __restrict__ M1, const float* __restrict__ a2, const float* __restrict__ b2, int relu2,
    uint32_t* __restrict__ P2, uint32_t* __restrict__ M2, float* __restrict__ out) {
  __shared__ uint32_t pieces[4][64];  // [plane][pixel]: byte w = the piece of wave w
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long q0 = (long long)blockIdx.x * 64 + lane;
  const long long q = q0 < npix_out ? q0 : npix_out - 1;  // lanes past the end copy the last pixel (nothing stored)
  const int word = blockIdx.y;
  const int hw = Ho * Wo;
  const int n = (int)(q / hw);
  const int r = (int)(q - (long long)n * hw);
  const int oy = r / Wo, ox = r - oy * Wo;
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) pieces[k][lane] = 0u;
  }
  __syncthreads();
  uint32_t p1 = 0u, m1 = 0u, p2 = 0u, m2 = 0u;
  const int c0 = word * 32 + wv * 8;
  const float* base = x + (((size_t)n * C + (size_t)c0) * H + 2 * oy) * W + 2 * ox;
  const size_t cstride = (size_t)H * W;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int c = c0 + b;
    if (c >= C) break;
    const float2 u = *reinterpret_cast<const float2*>(base + b * cstride);
    const float2 v = *reinterpret_cast<const float2*>(base + b * cstride + W);
    const float t = (((u.x + u.y) + v.x) + v.y) * 0.25f;   // (x / 4 and x * 0.25 are the same float for every x)
    if (out && q0 < npix_out) out[((size_t)n * C + c) * hw + r] = t;
    const float u1 = fmaf(t, a1[c], b1[c]);
    p1 |= (is_pos(u1) ? 1u : 0u) << b;
    m1 |= ((!relu1 && is_neg(u1)) ? 1u : 0u) << b;
    if (a2) {
      const float u2 = fmaf(t, a2[c], b2[c]);
      p2 |= (is_pos(u2) ? 1u : 0u) << b;
      m2 |= ((!relu2 && is_neg(u2)) ? 1u : 0u) << b;
    }
  }
  unsigned char* pb = reinterpret_cast<unsigned char*>(&pieces[0][0]);
  pb[(0 * 64 + lane) * 4 + wv] = (unsigned char)p1;
  pb[(1 * 64 + lane) * 4 + wv] = (unsigned char)m1;
  pb[(2 * 64 + lane) * 4 + wv] = (unsigned char)p2;
  pb[(3 * 64 + lane) * 4 + wv] = (unsigned char)m2;
  __syncthreads();
  if (wv == 0 && q0 < npix_out) {
    const size_t o = ((((size_t)n * (cw32 >> 1) + (word >> 1)) * hw + r) << 1) + (word & 1);
    P1[o] = pieces[0][lane];
    M1[o] = pieces[1][lane];
    if (a2) {
      P2[o] = pieces[2][lane];
      M2[o] = pieces[3][lane];
    }
  }
}

// One thread = one output pixel x one 32-channel word.
__global__ __launch_bounds__(256) void avgpool2_bn_pack2_wide_kernel(
    const float* __restrict__ x, int C, int H, int W, int Ho, int Wo, long long npix_out, int cw32,
    const float* __restrict__ a1, const float* __restrict__ b1, int relu1, uint32_t* __restrict__ P1,
    uint32_t* __restrict__ M1, const float* __restrict__ a2, const float* __restrict__ b2, int relu2,
    uint32_t* __restrict__ P2, uint32_t* __restrict__ M2, float* __restrict__ out) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= npix_out) return;
  const int word = blockIdx.y;
  const int hw = Ho * Wo;
  const int n = (int)(q / hw);
  const int r = (int)(q - (long long)n * hw);
  const int oy = r / Wo, ox = r - oy * Wo;
  uint32_t p1 = 0u, m1 = 0u, p2 = 0u, m2 = 0u;
  const float* base = x + (((size_t)n * C + (size_t)word * 32) * H + 2 * oy) * W + 2 * ox;
  const size_t cstride = (size_t)H * W;
#pragma unroll 8
  for (int b = 0; b < 32; ++b) {
    const int c = word * 32 + b;
    if (c >= C) break;
    const float2 u = *reinterpret_cast<const float2*>(base + b * cstride);
    const float2 v = *reinterpret_cast<const float2*>(base + b * cstride + W);
    const float t = (((u.x + u.y) + v.x) + v.y) * 0.25f;
    if (out) out[((size_t)n * C + c) * hw + r] = t;
    const float u1 = fmaf(t, a1[c], b1[c]);
    p1 |= (is_pos(u1) ? 1u : 0u) << b;
    m1 |= ((!relu1 && is_neg(u1)) ? 1u : 0u) << b;
    if (a2) {
      const float u2 = fmaf(t, a2[c], b2[c]);
      p2 |= (is_pos(u2) ? 1u : 0u) << b;
      m2 |= ((!relu2 && is_neg(u2)) ? 1u : 0u) << b;
    }
  }
  const size_t o = ((((size_t)n * (cw32 >> 1) + (word >> 1)) * hw + r) << 1) + (word & 1);
  P1[o] = p1;
  M1[o] = m1;
  if (a2) {
    P2[o] = p2;
    M2[o] = m2;
  }
}

int launch_avgpool2_bn_pack2(const float* x, int N, int C, int H, int W, const float* a1, const float* b1, int relu1,
                             uint64_t* P1, uint64_t* M1, const float* a2, const float* b2, int relu2, uint64_t* P2,
                             uint64_t* M2, float* out, hipStream_t stream) {
  const int Ho = H / 2, Wo = W / 2;
  const long long npix = (long long)N * Ho * Wo;
  const int cw32 = 2 * ((C + 63) / 64);
  if (npix * cw32 >= 150000LL) {  // enough threads to fill the chip with one per (pixel, word)
    hipLaunchKernelGGL(avgpool2_bn_pack2_wide_kernel, dim3((unsigned)((npix + 255) / 256), (unsigned)cw32), dim3(256), 0,
                       stream, x, C, H, W, Ho, Wo, npix, cw32, a1, b1, relu1, reinterpret_cast<uint32_t*>(P1),
                       reinterpret_cast<uint32_t*>(M1), a2, b2, relu2, reinterpret_cast<uint32_t*>(P2),
                       reinterpret_cast<uint32_t*>(M2), out);
    return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(avgpool2_bn_pack2_kernel, dim3((unsigned)((npix + 63) / 64), (unsigned)cw32), dim3(256), 0, stream,
                     x, C, H, W, Ho, Wo, npix, cw32, a1, b1, relu1, reinterpret_cast<uint32_t*>(P1),
                     reinterpret_cast<uint32_t*>(M1), a2, b2, relu2, reinterpret_cast<uint32_t*>(P2),
                     reinterpret_cast<uint32_t*>(M2), out);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// Tail of the real-valued stem in ONE pass over the stem conv's output
// (bnn/models/resnet.py:150-153: bn1 -> relu -> maxpool, then the first binary conv's sign()):
//     v = fma(x, bn_a[c], bn_b[c])   eval-mode BatchNorm (optional)
//     m = max over the k x k window (padding taps ignored, like nn.MaxPool2d)
//     y = relu(m)                    relu and max commute, so ReLU runs once per output
//     out_f32 = y ; P/M = sign(y)
// Reads the big tensor once (HBM-bound) instead of four times (BN, ReLU, pool, pack).
// One thread = one output pixel x one 32-channel word; consecutive lanes = consecutive ox, so
// fp32 stores are coalesced and the stride-2 window reads of neighbouring lanes share lines.
__global__ __launch_bounds__(256) void bn_relu_maxpool_pack_kernel(
    const float* __restrict__ x, int C, int H, int W, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int relu, int k, int stride, int pad, int Ho, int Wo,
    long long npix_out, int cw32, float* __restrict__ out, uint32_t* __restrict__ P,
    uint32_t* __restrict__ M) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= npix_out) return;
  const int word = blockIdx.y;
  const int hw = Ho * Wo;
  const int n = (int)(q / hw);
  const int r = (int)(q - (long long)n * hw);
  const int oy = r / Wo, ox = r - oy * Wo;
  const int y0 = oy * stride - pad, x0 = ox * stride - pad;
  uint32_t pw = 0u, mw = 0u;
  for (int b = 0; b < 32; ++b) {
    const int c = word * 32 + b;
    if (c >= C) break;
    const float* xc = x + ((size_t)n * C + c) * H * W;
    const float a = bn_a ? bn_a[c] : 1.0f, sh = bn_b ? bn_b[c] : 0.0f;
    float m = -INFINITY;
    for (int dy = 0; dy < k; ++dy) {
      const int iy = y0 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int ix = x0 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        m = fmaxf(m, fmaf(xc[(size_t)iy * W + ix], a, sh));
      }
    }
    if (relu) m = fmaxf(m, 0.0f);
    if (out) out[((size_t)n * C + c) * hw + r] = m;
    pw |= (is_pos(m) ? 1u : 0u) << b;
    mw |= (is_neg(m) ? 1u : 0u) << b;
  }
  if (P) {
    const size_t o = ((((size_t)n * (cw32 >> 1) + (word >> 1)) * hw + r) << 1) + (word & 1);
    P[o] = pw;
    M[o] = mw;
  }
}

// The ResNet stem's geometry (3x3 window, stride 2, pad 1, W % 4 == 0) with vector loads: one
// thread produces TWO horizontally adjacent outputs from one aligned float4 + one scalar per
// input row, i.e. 6 loads per channel for 2 outputs instead of 18 stride-2 dword loads.
__global__ __launch_bounds__(256) void bn_relu_maxpool3s2_pack_kernel(
    const float* __restrict__ x, int C, int H, int W, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int relu, int Ho, int Wo, long long npairs, int cw32,
    float* __restrict__ out, uint32_t* __restrict__ P, uint32_t* __restrict__ M) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // (n, oy, ox/2)
  if (q >= npairs) return;
  const int word = blockIdx.y;
  const int wp = Wo >> 1;
  const int n = (int)(q / ((long long)Ho * wp));
  const int rr = (int)(q - (long long)n * Ho * wp);
  const int oy = rr / wp, t = rr - oy * wp;
  const int hw = Ho * Wo;
  const int r = oy * Wo + 2 * t;
  uint32_t pw0 = 0u, mw0 = 0u, pw1 = 0u, mw1 = 0u;
  for (int b = 0; b < 32; ++b) {
    const int c = word * 32 + b;
    if (c >= C) break;
    const float* xc = x + ((size_t)n * C + c) * H * W;
    const float a = bn_a ? bn_a[c] : 1.0f, sh = bn_b ? bn_b[c] : 0.0f;
    float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy - 1 + dy;
      if ((unsigned)iy < (unsigned)H) {
        const float* row = xc + (size_t)iy * W + 4 * t;
        const float4 v = *reinterpret_cast<const float4*>(row);
        const float e = t > 0 ? fmaf(row[-1], a, sh) : -INFINITY;
        const float v0 = fmaf(v.x, a, sh), v1 = fmaf(v.y, a, sh), v2 = fmaf(v.z, a, sh),
                    v3 = fmaf(v.w, a, sh);
        m0 = fmaxf(m0, fmaxf(e, fmaxf(v0, v1)));
        m1 = fmaxf(m1, fmaxf(v1, fmaxf(v2, v3)));
      }
    }
    if (relu) { m0 = fmaxf(m0, 0.0f); m1 = fmaxf(m1, 0.0f); }
    if (out) *reinterpret_cast<float2*>(out + ((size_t)n * C + c) * hw + r) = make_float2(m0, m1);
    pw0 |= (is_pos(m0) ? 1u : 0u) << b;
    mw0 |= (is_neg(m0) ? 1u : 0u) << b;
    pw1 |= (is_pos(m1) ? 1u : 0u) << b;
    mw1 |= (is_neg(m1) ? 1u : 0u) << b;
  }
  if (P) {
    const size_t o = ((((size_t)n * (cw32 >> 1) + (word >> 1)) * hw + r) << 1) + (word & 1);
    P[o] = pw0; M[o] = mw0;
    P[o + 2] = pw1; M[o + 2] = mw1;
  }
}

int launch_bn_relu_maxpool_pack(const float* x, int N, int C, int H, int W, const float* bn_a,
                                const float* bn_b, int relu, int k, int stride, int pad, float* out,
                                uint64_t* P, uint64_t* M, hipStream_t stream) {
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const long long npix = (long long)N * Ho * Wo;
  const int cw32 = 2 * ((C + 63) / 64);
  if (k == 3 && stride == 2 && pad == 1 && W % 4 == 0 && Wo == W / 2 &&
      (reinterpret_cast<uintptr_t>(x) & 15u) == 0 &&
      (!out || (reinterpret_cast<uintptr_t>(out) & 7u) == 0)) {
    const long long npairs = npix / 2;
    hipLaunchKernelGGL(bn_relu_maxpool3s2_pack_kernel,
                       dim3((unsigned)((npairs + 255) / 256), (unsigned)cw32), dim3(256), 0, stream, x,
                       C, H, W, bn_a, bn_b, relu, Ho, Wo, npairs, cw32, out,
                       reinterpret_cast<uint32_t*>(P), reinterpret_cast<uint32_t*>(M));
    return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(bn_relu_maxpool_pack_kernel,
                     dim3((unsigned)((npix + 255) / 256), (unsigned)cw32), dim3(256), 0, stream, x, C,
                     H, W, bn_a, bn_b, relu, k, stride, pad, Ho, Wo, npix, cw32, out,
                     reinterpret_cast<uint32_t*>(P), reinterpret_cast<uint32_t*>(M));
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
