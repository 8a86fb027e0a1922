// stem.hip — the real-valued first layer of the reference's ResNets as ONE kernel on the fp32 matrix
// cores:   conv 7x7 / stride 2 / pad 3 (3 -> 64)  ->  BatchNorm(eval)  ->  ReLU  ->  MaxPool 3x3 / 2 / 1
//          ->  fp32 NCHW  +  sign() bit planes for the first binary conv.
// Reference graph: bnn/models/resnet.py:93-96,150-153 (conv1, bn1, relu, maxpool); the layer is kept
// real-valued by the recipes (examples/cifar10.py:71), so it is the one place MFMA is used.
//
// Implicit GEMM per workgroup (4 waves): M = 17x15 conv pixels (the 8x7 pooled tile + its 1-pixel pool
// halo, 255 -> 16 sub-tiles of 16), N = 64 output channels, K = 3*7*7 = 147 (-> 37 steps of 4) with
// v_mfma_f32_16x16x4_f32: exact fp32 multiply-adds (bit-for-bit a k-ordered fmaf chain), so the result
// agrees with the reference's fp32 convolution to rounding.  The input patch (39x35x3 floats), the whole
// weight matrix and the post-BN/ReLU conv tile live in LDS; conv outputs never touch HBM (the library
// path writes and re-reads 822 MB at batch 256).  Persistent workgroups: weights are staged once.
#include "bnn_dev.h"

namespace bnn {

namespace stem {
constexpr int CIN = 3, KS = 7, COUT = 64, K = CIN * KS * KS;  // 147
constexpr int KSTEPS = (K + 3) / 4;                              // 37
constexpr int PTH = 8, PTW = 7;                                  // pooled tile
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1;              // conv tile 17 x 15 (pool halo incl.)
constexpr int MPIX = CTH * CTW;                                  // 255
constexpr int ITH = 2 * CTH + 5, ITW = 2 * CTW + 5;              // input patch 39 x 35
constexpr int ICH = ITH * ITW;                                   // 1365 floats per channel
constexpr int NIN = CIN * ICH;                                   // 4095
constexpr int SM = 257;                                          // stage row stride (bank-conflict pad)
constexpr int NT = 256;
// LDS carve (floats).  The conv tile is staged 32 channels at a time into the space the input patch
// occupied during the GEMM, which keeps a workgroup at 70.8 KB: TWO workgroups per CU, so one can
// run its matrix phase while the other loads / pools.
constexpr int HALF = 32;                                         // channels per staging round
constexpr int OFF_IN = 0;                                        // input patch, later the stage
constexpr int OFF_STAGE = 0;
constexpr int OFF_W = HALF * SM;              // [K4 = 148][16][4]  -> one ds_read_b128 per B fragment set
constexpr int LDS_FLOATS = OFF_W + KSTEPS * 4 * COUT;
static_assert(HALF * SM >= NIN, "stage region must also hold the input patch");
}  // namespace stem

using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(stem::NT, 2) void stem_conv_bn_relu_pool_pack_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int N, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
    int tiles_x, int per_xcd, float* __restrict__ out, uint64_t* __restrict__ P,
    uint64_t* __restrict__ M) {
  using namespace stem;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* in_t = lds + OFF_IN;
  float* w_t = lds + OFF_W;
  float* stage = lds + OFF_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;

  // ---- once per workgroup: weights as B fragments
  // w_t[(k*16 + j)*4 + t] = w[o = 16 t + j][k]   (k >= K: 0)
  for (int e = tid; e < KSTEPS * 4 * COUT; e += NT) {
    const int k = e >> 6, j = (e >> 2) & 15, t = e & 3;
    w_t[e] = k < K ? w[(size_t)(16 * t + j) * K + k] : 0.0f;
  }
  // this lane's A offsets: k = 4*ks + (lane>>4) -> (c, ky, kx) -> offset inside the input patch.
  // Kept in registers (37 VGPRs; one wave per SIMD has the whole file) so that the fully unrolled
  // K loop has no dependent LDS lookup and the compiler can prefetch operands across steps.
  int ko[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int k = ks * 4 + lg;
    const int kk = k < K ? k : 0;
    const int c = kk / (KS * KS), rem = kk - c * KS * KS, ky = rem / KS, kx = rem - ky * KS;
    ko[ks] = c * ICH + ky * ITW + kx;
  }
  // per-lane BN constants for the 4 channel sub-tiles of the accumulator layout (col = lane & 15)
  float ba[4], bb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { ba[t] = bn_a[16 * t + li]; bb[t] = bn_b[16 * t + li]; }
  // per-lane A bases: conv pixel m = (wave*4 + s)*16 + li  ->  patch offset 2*cy*ITW + 2*cx
  int abase[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    int m = (wave * 4 + s) * 16 + li;
    if (m >= MPIX) m = MPIX - 1;
    const int cy = m / CTW, cx = m - cy * CTW;
    abase[s] = 2 * cy * ITW + 2 * cx;
  }

  const int ntiles = N * tiles_y * tiles_x;
  // Tile order: workgroup b sits on XCD b % 8 (observed placement, used for speed only).  Each XCD
  // walks ONE contiguous eighth of the tile list, so the tiles that share an output cache line
  // (neighbours in x) and an input halo are handled by the same L2 within a short time window.
  for (int seq = blockIdx.x; seq < per_xcd * 8; seq += gridDim.x) {
    const int tile = (seq & 7) * per_xcd + (seq >> 3);
    if (tile >= ntiles) continue;
    const int n = tile / (tiles_y * tiles_x);
    const int tr = tile - n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int py0 = ty * PTH, px0 = tx * PTW;        // pooled origin
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // conv origin (pool pad 1)
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;  // input origin (conv pad 3)

    for (int e = tid; e < NIN; e += NT) {
      const int c = e / ICH, rem = e - c * ICH, r = rem / ITW, col = rem - r * ITW;
      const int iy = iy0 + r, ix = ix0 + col;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      in_t[e] = ok ? x[(((size_t)n * CIN + c) * H + iy) * W + ix] : 0.0f;
    }
    __syncthreads();

    // ---- implicit GEMM on the matrix cores
    f32x4 acc[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k = ks * 4 + lg;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(w_t + ((size_t)k * 16 + li) * 4);
      float av[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float a = in_t[abase[s] + ko[ks]];
        av[s] = (ks < KSTEPS - 1 || k < K) ? a : 0.0f;  // the padded k column must not inject inf*0
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[t], acc[s][t], 0, 0, 0);
    }

    __syncthreads();  // every wave is done reading the input patch: its space becomes the stage

    // ---- two rounds of 32 channels: BN + ReLU -> LDS, then 3x3 / stride-2 max pool out of LDS
    // D layout: col = lane&15 (channel in sub-tile), row = 4*(lane>>4)+r
    const int p = tid >> 2, q = tid & 3;  // pooling role: pixel p (< 56), 8-channel group q
    const int ly = p / PTW, lx = p - ly * PTW;
    const int py = py0 + ly, px = px0 + lx;
    const bool live = p < PTH * PTW && py < Hp && px < Wp;
    const int mb = (2 * ly) * CTW + 2 * lx;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = (wave * 4 + s) * 16 + lg * 4 + r;
          if (m < MPIX) {
            const int cy = m / CTW, cx = m - cy * CTW;
            const bool inside = (unsigned)(cy0 + cy) < (unsigned)Hc && (unsigned)(cx0 + cx) < (unsigned)Wc;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
              const int t = 2 * half + tt;
              const float v = fmaxf(fmaf(acc[s][t][r], ba[t], bb[t]), 0.0f);
              // positions outside the conv output are MaxPool padding: 0 never beats a ReLU output
              stage[(16 * tt + li) * SM + m] = inside ? v : 0.0f;
            }
          }
        }
      }
      __syncthreads();
      uint32_t bits = 0u;
      if (p < PTH * PTW) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int cl = q * 8 + cc;
          const float* sp = stage + cl * SM + mb;
          const float m0 = fmaxf(fmaxf(sp[0], sp[1]), sp[2]);
          const float m1 = fmaxf(fmaxf(sp[CTW], sp[CTW + 1]), sp[CTW + 2]);
          const float m2 = fmaxf(fmaxf(sp[2 * CTW], sp[2 * CTW + 1]), sp[2 * CTW + 2]);
          const float v = fmaxf(fmaxf(m0, m1), m2);
          if (live && out) out[(((size_t)n * COUT + HALF * half + cl) * Hp + py) * Wp + px] = v;
          bits |= (is_pos(v) ? 1u : 0u) << cc;
        }
      }
      // the four 8-bit groups of a pixel sit in adjacent lanes: one 32-bit half of the uint64 word
      const uint32_t b1 = __shfl_down(bits, 1, 64), b2 = __shfl_down(bits, 2, 64),
                     b3 = __shfl_down(bits, 3, 64);
      if (live && q == 0 && P) {
        const size_t o = (((size_t)n * Hp + py) * Wp + px) * 2 + half;  // cw64 == 1 for 64 channels
        reinterpret_cast<uint32_t*>(P)[o] = bits | (b1 << 8) | (b2 << 16) | (b3 << 24);
        reinterpret_cast<uint32_t*>(M)[o] = 0u;  // nothing is negative after ReLU
      }
      __syncthreads();  // stage is rewritten by the next round / the next tile's input patch
    }
  }
}

int launch_stem(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                int W, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  using namespace stem;
  const int Hc = (H + 6 - KS) / 2 + 1, Wc = (W + 6 - KS) / 2 + 1;
  const int Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW;
  const long long ntiles = (long long)N * tiles_y * tiles_x;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  }
  const int per_xcd = (int)((ntiles + 7) / 8);
  const long long want = 2LL * cus;  // two resident workgroups per CU
  const unsigned grid = (unsigned)(ntiles < want ? ((ntiles + 7) / 8 * 8) : want);
  const size_t lds_bytes = (size_t)LDS_FLOATS * sizeof(float);
  static bool attr_set[64] = {false};  // >64 KB of dynamic LDS needs the opt-in, once per device
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_bn_relu_pool_pack_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(stem_conv_bn_relu_pool_pack_kernel, dim3(grid), dim3(NT), lds_bytes, stream, x, w,
                     bn_a, bn_b, N, H, W, Hc, Wc, Hp, Wp, tiles_y, tiles_x, per_xcd, out, P, M);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
