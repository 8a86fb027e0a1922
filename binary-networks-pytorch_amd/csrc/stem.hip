// stem.hip — the real-valued first layer of the reference's ResNets as ONE kernel on the matrix cores:
//     conv 7x7 / stride 2 / pad 3 (3 -> 64)  ->  BatchNorm(eval)  ->  ReLU  ->  MaxPool 3x3 / 2 / 1
//     ->  fp32 NCHW  +  sign() bit planes for the first binary conv.
// Reference graph: bnn/models/resnet.py:93-96,150-153 (conv1, bn1, relu, maxpool); the layer is kept
// real-valued by the recipes (examples/cifar10.py:71), so it is the one place MFMA is used.
//
// Implicit GEMM per workgroup (8 waves): M = 17x15 conv pixels (the 8x7 pooled tile + its 1-pixel pool
// halo, 255 -> 16 sub-tiles of 16, two per wave), N = 64 output channels, K = 3*7*7 = 147.
// Two arithmetic modes:
//   SPLIT (default): every fp32 operand is split into two fp16 halves (x = hi + lo, 22 mantissa bits)
//     and the product is formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32
//     accumulation: measured error 2.8e-7 relative (the rounding class of an fp32 convolution) at
//     3/16 of the matrix time of the fp32 instruction.
//   exact (BNN_HIP_STEM_EXACT_FP32): v_mfma_f32_16x16x4_f32, bit-for-bit a k-ordered fp32 fmaf chain.
// The input patch (39x35x3 floats), the whole weight matrix and the post-BN/ReLU conv tile live in LDS;
// conv outputs never touch HBM (the library path writes and re-reads 822 MB at batch 256).  Persistent
// workgroups (one per CU): weights are staged once; the NEXT tile's input patch is fetched into registers
// while the matrix phase of the current tile runs; tiles are walked in XCD-contiguous order.
#include "bnn_dev.h"

namespace bnn {

namespace stem {
constexpr int CIN = 3, KS = 7, COUT = 64, K = CIN * KS * KS;  // 147
constexpr int KSTEPS = (K + 3) / 4;                              // 37 steps of 4  (fp32 instruction)
constexpr int KSTEPS16 = (K + 31) / 32;                          // 5 steps of 32  (fp16 instruction)
constexpr int PTH = 8, PTW = 7;                                  // pooled tile
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1;              // conv tile 17 x 15 (pool halo incl.)
constexpr int MPIX = CTH * CTW;                                  // 255
constexpr int ITH = 2 * CTH + 5, ITW = 2 * CTW + 5;              // input patch 39 x 35
constexpr int ICH = ITH * ITW;                                   // 1365 floats per channel
constexpr int NIN = CIN * ICH;                                   // 4095
constexpr int SM = 257;                                          // stage row stride (bank-conflict pad)
constexpr int NT = 512, NWAVE = NT / 64, SUB = 16 / NWAVE;       // 8 waves, 2 M-sub-tiles each
constexpr int PER_T = (NIN + NT - 1) / NT;                       // input floats per thread (8)
// LDS carve (floats)
constexpr int OFF_IN = 0;
constexpr int OFF_STAGE = 4096;
constexpr int OFF_W = OFF_STAGE + COUT * SM;
constexpr int W_FLOATS_F32 = KSTEPS * 4 * COUT;                  // [148][16][4] floats
constexpr int W_FLOATS_SPLIT = KSTEPS16 * 4 * 64 * 8;            // hi + lo halves, each [ks][t][lane][8]
constexpr int lds_floats(bool split) { return OFF_W + (split ? W_FLOATS_SPLIT : W_FLOATS_F32); }
}  // namespace stem

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

template <bool SPLIT>
__global__ __launch_bounds__(stem::NT, 2) void stem_conv_bn_relu_pool_pack_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int N, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
    int tiles_x, int per_xcd, float* __restrict__ out, uint64_t* __restrict__ P,
    uint64_t* __restrict__ M) {
  using namespace stem;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* in_t = lds + OFF_IN;
  float* stage = lds + OFF_STAGE;
  float* w_t = lds + OFF_W;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;

  // ---- once per workgroup: weights as B fragments; this lane's A offsets (k -> patch offset).
  // Offsets live in registers so that the fully unrolled K loop has no dependent LDS lookup.
  constexpr int NKO = SPLIT ? KSTEPS16 * 8 : KSTEPS;
  int ko[NKO];
  auto patch_off = [](int k) {
    const int kk = k < K ? k : 0;
    const int c = kk / (KS * KS), rem = kk - c * KS * KS, ky = rem / KS, kx = rem - ky * KS;
    return c * ICH + ky * ITW + kx;
  };
  if constexpr (SPLIT) {
    // f16 16x16x32: lane holds 8 consecutive k of row/col (lane&15): k = 32*ks + 8*(lane>>4) + e
    // LDS: hi halves then lo halves, each [ks][t][lane][8]  -> one 16-byte read per (ks, t, part)
    _Float16* wh = reinterpret_cast<_Float16*>(w_t);
    _Float16* wl = wh + KSTEPS16 * 4 * 64 * 8;
    for (int idx = tid; idx < KSTEPS16 * 4 * 64 * 8; idx += NT) {
      const int e = idx & 7, ln = (idx >> 3) & 63, t = (idx >> 9) & 3, ks = idx >> 11;
      const int k = 32 * ks + 8 * (ln >> 4) + e;
      const float v = k < K ? w[(size_t)(16 * t + (ln & 15)) * K + k] : 0.0f;
      const _Float16 h = (_Float16)v;
      wh[idx] = h;
      wl[idx] = (_Float16)(v - (float)h);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS16; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) ko[ks * 8 + e] = patch_off(32 * ks + 8 * lg + e);
  } else {
    // f32 16x16x4: w_t[(k*16 + j)*4 + t] = w[o = 16 t + j][k]   (k >= K: 0)
    for (int e = tid; e < KSTEPS * 4 * COUT; e += NT) {
      const int k = e >> 6, j = (e >> 2) & 15, t = e & 3;
      w_t[e] = k < K ? w[(size_t)(16 * t + j) * K + k] : 0.0f;
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) ko[ks] = patch_off(ks * 4 + lg);
  }
  // per-lane BN constants for the 4 channel sub-tiles of the accumulator layout (col = lane & 15)
  float ba[4], bb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { ba[t] = bn_a[16 * t + li]; bb[t] = bn_b[16 * t + li]; }
  // per-lane A bases: conv pixel m = (wave*SUB + s)*16 + li  ->  patch offset 2*cy*ITW + 2*cx
  int abase[SUB];
#pragma unroll
  for (int s = 0; s < SUB; ++s) {
    int m = (wave * SUB + s) * 16 + li;
    if (m >= MPIX) m = MPIX - 1;
    const int cy = m / CTW, cx = m - cy * CTW;
    abase[s] = 2 * cy * ITW + 2 * cx;
  }
  // pooling role of this thread: pooled pixel p (< 56) x 8-channel group q
  const int pp = tid >> 3, pq = tid & 7;
  const int ply = pp / PTW, plx = pp - ply * PTW;
  const int pmb = (2 * ply) * CTW + 2 * plx;

  const int ntiles = N * tiles_y * tiles_x;
  const int nseq = per_xcd * 8;
  // Tile order: workgroup b sits on XCD b % 8 (observed placement, used for speed only).  Each XCD
  // walks ONE contiguous eighth of the tile list, so the tiles that share an output cache line
  // (neighbours in x) and an input halo are handled by the same L2 within a short time window.
  auto tile_of = [&](int seq) { return (seq & 7) * per_xcd + (seq >> 3); };
  // input patch of tile `tile` -> registers (zeros outside the image = the conv's zero padding)
  float nxt[PER_T];
  auto fetch = [&](int tile) {
    const bool valid = tile < ntiles;
    const int tl = valid ? tile : 0;
    const int n = tl / (tiles_y * tiles_x);
    const int tr = tl - n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int iy0 = 2 * (2 * ty * PTH - 1) - 3, ix0 = 2 * (2 * tx * PTW - 1) - 3;
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int e = tid + u * NT;
      const int c = e / ICH, rem = e - c * ICH, r = rem / ITW, col = rem - r * ITW;
      const int iy = iy0 + r, ix = ix0 + col;
      const bool ok = valid && e < NIN && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      nxt[u] = ok ? x[(((size_t)n * CIN + c) * H + iy) * W + ix] : 0.0f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int e = tid + u * NT;
      if (e < NIN) in_t[e] = nxt[u];
    }
  };

  int seq = blockIdx.x;
  if (seq < nseq) { fetch(tile_of(seq)); commit(); }
  for (; seq < nseq; seq += gridDim.x) {
    const int tile = tile_of(seq);
    const bool valid = tile < ntiles;  // workgroup-uniform
    __syncthreads();                   // in_t holds this tile's patch; stage is free again
    const int nseq_next = seq + gridDim.x;
    if (nseq_next < nseq) fetch(tile_of(nseq_next));  // global loads fly during the matrix phase
    const int tl = valid ? tile : 0;
    const int n = tl / (tiles_y * tiles_x);
    const int tr = tl - n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int py0 = ty * PTH, px0 = tx * PTW;        // pooled origin
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // conv origin (pool pad 1)

    // ---- implicit GEMM on the matrix cores
    f32x4 acc[SUB][4];
#pragma unroll
    for (int s = 0; s < SUB; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (SPLIT) {
      const _Float16* wh = reinterpret_cast<const _Float16*>(w_t);
      const _Float16* wl = wh + KSTEPS16 * 4 * 64 * 8;
#pragma unroll
      for (int ks = 0; ks < KSTEPS16; ++ks) {
        half8 bh[4], bl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bh[t] = *reinterpret_cast<const half8*>(wh + ((ks * 4 + t) * 64 + lane) * 8);
          bl[t] = *reinterpret_cast<const half8*>(wl + ((ks * 4 + t) * 64 + lane) * 8);
        }
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
          half8 ah, al;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = in_t[abase[s] + ko[ks * 8 + e]];
            if (32 * ks + 8 * 3 + 7 >= K) a = (32 * ks + 8 * lg + e < K) ? a : 0.0f;  // padded k: no inf*0
            const _Float16 h = (_Float16)a;
            ah[e] = h;
            al[e] = (_Float16)(a - (float)h);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[t], acc[s][t], 0, 0, 0);
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[t], acc[s][t], 0, 0, 0);
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[t], acc[s][t], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int k = ks * 4 + lg;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(w_t + ((size_t)k * 16 + li) * 4);
        float av[SUB];
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
          const float a = in_t[abase[s] + ko[ks]];
          av[s] = (ks < KSTEPS - 1 || k < K) ? a : 0.0f;  // the padded k column must not inject inf*0
        }
#pragma unroll
        for (int s = 0; s < SUB; ++s)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[t], acc[s][t], 0, 0, 0);
      }
    }

    // ---- BN + ReLU, conv tile -> LDS.  D layout: col = lane&15 (channel in sub-tile), row = 4*(lane>>4)+r
#pragma unroll
    for (int s = 0; s < SUB; ++s) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (wave * SUB + s) * 16 + lg * 4 + r;
        if (m < MPIX) {
          const int cy = m / CTW, cx = m - cy * CTW;
          const bool inside = (unsigned)(cy0 + cy) < (unsigned)Hc && (unsigned)(cx0 + cx) < (unsigned)Wc;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float v = fmaxf(fmaf(acc[s][t][r], ba[t], bb[t]), 0.0f);
            // positions outside the conv output are MaxPool padding: 0 never beats a ReLU output
            stage[(16 * t + li) * SM + m] = inside ? v : 0.0f;
          }
        }
      }
    }
    __syncthreads();  // conv tile staged; every wave is done reading in_t
    if (nseq_next < nseq) commit();  // next tile's patch: registers -> LDS (overlaps the pooling)

    // ---- 3x3 / stride-2 max pool out of LDS, fp32 store + sign planes (M plane is 0 after ReLU)
    const int py = py0 + ply, px = px0 + plx;
    const bool live = valid && pp < PTH * PTW && py < Hp && px < Wp;
    uint32_t bits = 0u;
    if (live) {
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const int ch = pq * 8 + cc;
        const float* sp = stage + ch * SM + pmb;
        const float m0 = fmaxf(fmaxf(sp[0], sp[1]), sp[2]);
        const float m1 = fmaxf(fmaxf(sp[CTW], sp[CTW + 1]), sp[CTW + 2]);
        const float m2 = fmaxf(fmaxf(sp[2 * CTW], sp[2 * CTW + 1]), sp[2 * CTW + 2]);
        const float v = fmaxf(fmaxf(m0, m1), m2);
        if (out) out[(((size_t)n * COUT + ch) * Hp + py) * Wp + px] = v;
        bits |= (is_pos(v) ? 1u : 0u) << cc;
      }
    }
    if (P) {  // the 8 lanes of a pixel each hold byte `pq` of its 64-channel word (cw64 == 1)
      uint32_t wd = bits << (8 * (pq & 3));
      wd |= (uint32_t)__shfl_xor((int)wd, 1);
      wd |= (uint32_t)__shfl_xor((int)wd, 2);
      const uint32_t other = (uint32_t)__shfl_xor((int)wd, 4);
      if (live && pq == 0) {
        const size_t o = ((size_t)n * Hp + py) * Wp + px;
        P[o] = (uint64_t)wd | ((uint64_t)other << 32);
        M[o] = 0;  // nothing is negative after ReLU
      }
    }
  }
}

template <bool SPLIT>
static int launch_stem_t(const float* x, const float* w, const float* bn_a, const float* bn_b, int N,
                         int H, int W, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  using namespace stem;
  const int Hc = (H + 6 - KS) / 2 + 1, Wc = (W + 6 - KS) / 2 + 1;
  const int Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW;
  const long long ntiles = (long long)N * tiles_y * tiles_x;
  const int cus = current_device_cus();
  const int per_xcd = (int)((ntiles + 7) / 8);
  const long long want = cus;  // one resident workgroup (8 waves) per CU
  const unsigned grid = (unsigned)(ntiles < want ? ((ntiles + 7) / 8 * 8) : want);
  const size_t lds_bytes = (size_t)lds_floats(SPLIT) * sizeof(float);
  // > 64 KB of dynamic LDS needs the opt-in: per device and per kernel, so it is set on every launch (no mutable
  // global state in a re-entrant API; the call is a table write on the host)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_bn_relu_pool_pack_kernel<SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_bytes) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  hipLaunchKernelGGL(stem_conv_bn_relu_pool_pack_kernel<SPLIT>, dim3(grid), dim3(NT), lds_bytes, stream,
                     x, w, bn_a, bn_b, N, H, W, Hc, Wc, Hp, Wp, tiles_y, tiles_x, per_xcd, out, P, M);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_stem(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                int W, int flags, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  if (flags & BNN_HIP_STEM_EXACT_FP32) return launch_stem_t<false>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream);
  return launch_stem_rows(x, w, bn_a, bn_b, N, H, W, (flags & BNN_HIP_STEM_FP16) != 0, out, P, M, stream);
}

}  // namespace bnn
