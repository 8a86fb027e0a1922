// stem_lean.hip — the fused stem (stem_split.hip) re-cut so that it SHARES the CU with other kernels.
//
// stem_split.hip uses 8 waves x 255 VGPRs per CU: while it runs, no wave of any other kernel fits on the
// chip, and with two batches in flight (PipelinedInference) its ~330 us are pure serial time although it keeps
// the vector ALU only ~30 % busy.  This variant runs ONE wave per SIMD (4-wave workgroups, <= 256 VGPRs), so
// half of every SIMD's register file stays free: the integer-ALU-bound binary convolutions of the other batch
// are co-resident and issue into the cycles the stem spends waiting on LDS, the matrix cores and HBM.
// Measured (tools/exp_twostream.py, batch 256): alone 560 us instead of 330; with two batches in flight the
// step takes 1.411 ms against 1.414 ms with the dense variant — co-residency does hide the extra 230 us, but
// the convolutions that share the SIMD run with half the registers (2-3 waves instead of 4-8) and lose what
// the stem's idle cycles give.  Kept as an option (BNN_HIP_STEM_SHARE_CU), not the default.
//
// Same arithmetic (fp16 hi/lo split, 3 MFMAs, fp32 accumulate) and the same results bit for bit; what changes:
//   * 4 waves = 4 pixel groups; each wave walks BOTH channel halves of its 4 sub-tiles one after the other;
//   * the B fragments (weights, hi/lo) live in LDS ([half][k-step][tile][hi|lo][lane][8]) and are re-read per
//     k-step — 16 live registers instead of 96;
//   * patch rows are dense (36 halves) so that patch + weights + staged tile fit in 136 KB of LDS;
//   * pooling walks its 64 channels in two passes of 32.
#include "bnn_dev.h"

namespace bnn {

namespace stem3 {
constexpr int CIN = 3, KS = 7, COUT = 64;
constexpr int KROWS = 24, KSTEPS = KROWS / 4;
constexpr int PTH = 8, PTW = 7;
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1;
constexpr int MPIX = CTH * CTW;
constexpr int ITH = 2 * CTH + 5;
constexpr int ITWP = 36;
constexpr int ROWH = ITWP;
constexpr int ICHP = ITH * ROWH;
constexpr int NINP = CIN * ICHP;
constexpr int NROW = CIN * ITH;
constexpr int NPC = ITWP / 2;
constexpr int SC = COUT + 4;
constexpr int NT = 256;
constexpr int RSTEP = NT / NPC;                    // 14 rows per sweep (252 fetching threads)
constexpr int PER_T = (NROW + RSTEP - 1) / RSTEP;  // 9 column pairs per thread
constexpr int SUBS = 4, TT = 2, NH = 2;
constexpr int WFRAG = 64 * 8;                       // halves per (half, k-step, tile, part) fragment
constexpr int OFF_HI = 0;
constexpr int OFF_LO = OFF_HI + ((NINP * 2 + 15) / 16) * 16;
constexpr int OFF_STAGE = OFF_LO + ((NINP * 2 + 15) / 16) * 16;
constexpr int OFF_W = OFF_STAGE + (MPIX + 1) * SC * 4;
constexpr int OFF_BITS = OFF_W + NH * KSTEPS * TT * 2 * WFRAG * 2;
constexpr int LDS_BYTES = OFF_BITS + 2 * PTH * PTW * 8;
}  // namespace stem3

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__global__ __launch_bounds__(stem3::NT, 2) void stem_lean_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int N, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
    int tiles_x, int per_xcd, float* __restrict__ out, uint64_t* __restrict__ P,
    uint64_t* __restrict__ M) {
  using namespace stem3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* hiP = reinterpret_cast<_Float16*>(lds_raw + OFF_HI);
  _Float16* loP = reinterpret_cast<_Float16*>(lds_raw + OFF_LO);
  float* stage = reinterpret_cast<float*>(lds_raw + OFF_STAGE);
  _Float16* wl = reinterpret_cast<_Float16*>(lds_raw + OFF_W);
  uint8_t* bits = lds_raw + OFF_BITS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;  // wave = pixel group (4 sub-tiles of 16 conv pixels)
  const int li = lane & 15, lg = lane >> 4;

  // ---- once: weights as MFMA B fragments in LDS.  Fragment (nh, ks, tt, part): lane l holds
  // B[k = 8*(l>>4) + e][j = l&15] = w[o = 32*nh + 16*tt + (l&15)][c][ky][kx = e], (c, ky) = row 4*ks + (l>>4).
  for (int idx = tid; idx < NH * KSTEPS * TT * WFRAG; idx += NT) {
    const int e = idx & 7, l = (idx >> 3) & 63, f = idx >> 9;  // f = (nh*KSTEPS + ks)*TT + tt
    const int tt = f % TT, ks = (f / TT) % KSTEPS, nh = f / (TT * KSTEPS);
    const int krow = 4 * ks + (l >> 4);
    const int c = krow / KS, ky = krow - c * KS;
    const int o = 32 * nh + 16 * tt + (l & 15);
    const float v = (krow < CIN * KS && e < KS) ? w[((size_t)(o * CIN + c) * KS + ky) * KS + e] : 0.0f;
    const _Float16 h = (_Float16)v;
    wl[(2 * f) * WFRAG + l * 8 + e] = h;
    wl[(2 * f + 1) * WFRAG + l * 8 + e] = (_Float16)(v - (float)h);
  }
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int krow = 4 * ks + lg;
    const int c = krow / KS, ky = krow - c * KS;
    koff[ks] = krow < CIN * KS ? c * ICHP + ky * ROWH : 0;
  }
  int abase[SUBS];
#pragma unroll
  for (int i = 0; i < SUBS; ++i) {
    int m = (SUBS * wave + i) * 16 + li;
    if (m >= MPIX) m = MPIX - 1;
    const int cy = m / CTW, cx = m - cy * CTW;
    abase[i] = 2 * cy * ROWH + 2 * cx;
  }
  const int fpc = tid % NPC, frow0 = tid / NPC;
  const bool fetcher = tid < NPC * RSTEP;
  int f_goff[PER_T];
#pragma unroll
  for (int u = 0; u < PER_T; ++u) {
    const int R = frow0 + RSTEP * u;
    const int c = R / ITH, r = R - c * ITH;
    f_goff[u] = (fetcher && R < NROW) ? (c * H + r) * W + 2 * fpc : -1;
  }
  const int pchl = lane & 7, pplx = lane >> 3;

  const int ntiles = N * tiles_y * tiles_x;
  const int nseq = per_xcd * 8;
  auto tile_of = [&](int seq) { return (seq & 7) * per_xcd + (seq >> 3); };

  float nx0[PER_T], nx1[PER_T];
  auto fetch = [&](int tile) {
    const bool valid = tile < ntiles;
    const int tl = valid ? tile : 0;
    const int n = tl / (tiles_y * tiles_x);
    const int tr = tl - n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int iy0 = 2 * (2 * ty * PTH - 1) - 3, ix0 = 2 * (2 * tx * PTW - 1) - 3;
    const float* xb = x + (size_t)n * CIN * H * W + (ptrdiff_t)iy0 * W + ix0;
    const int ix = ix0 + 2 * fpc;
    const bool okc0 = valid && (unsigned)ix < (unsigned)W;
    const bool okc1 = valid && 2 * fpc + 1 < ITWP - 1 && (unsigned)(ix + 1) < (unsigned)W;  // col 35: zero
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int R = frow0 + RSTEP * u;
      const int r = R - (R >= 2 * ITH ? 2 * ITH : R >= ITH ? ITH : 0);
      const bool okr = f_goff[u] >= 0 && (unsigned)(iy0 + r) < (unsigned)H;
      nx0[u] = (okr && okc0) ? xb[f_goff[u]] : 0.0f;
      nx1[u] = (okr && okc1) ? xb[f_goff[u] + 1] : 0.0f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int R = frow0 + RSTEP * u;
      if (fetcher && R < NROW) {
        half2v h, l;
        h[0] = (_Float16)nx0[u];
        h[1] = (_Float16)nx1[u];
        l[0] = (_Float16)(nx0[u] - (float)h[0]);
        l[1] = (_Float16)(nx1[u] - (float)h[1]);
        reinterpret_cast<half2v*>(hiP)[R * (ROWH / 2) + fpc] = h;
        reinterpret_cast<half2v*>(loP)[R * (ROWH / 2) + fpc] = l;
      }
    }
  };
  auto load_a = [&](const _Float16* plane, int off) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(plane) + (off >> 1);
    u32x4 v;
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    return __builtin_bit_cast(half8, v);
  };

  int prev_n = -1, prev_py0 = 0, prev_px0 = 0, buf = 0;
  auto flush_bits = [&](int b) {
    if (P && prev_n >= 0 && tid < PTH * PTW) {
      const int ply = tid / PTW, plx = tid - ply * PTW;
      const int py = prev_py0 + ply, px = prev_px0 + plx;
      if (py < Hp && px < Wp) {
        const size_t o = ((size_t)prev_n * Hp + py) * Wp + px;
        P[o] = *reinterpret_cast<const uint64_t*>(bits + (b * PTH * PTW + tid) * 8);
        M[o] = 0;  // nothing is negative after ReLU
      }
    }
  };

  int seq = blockIdx.x;
  if (seq < nseq) { fetch(tile_of(seq)); commit(); }
  for (; seq < nseq; seq += gridDim.x) {
    const int tile = tile_of(seq);
    const bool valid = tile < ntiles;
    __syncthreads();  // patch (and, the first time, the weights) in LDS; `stage` free again
    flush_bits(buf ^ 1);
    const int seq_next = seq + gridDim.x;
    if (seq_next < nseq) fetch(tile_of(seq_next));
    const int tl = valid ? tile : 0;
    const int n = tl / (tiles_y * tiles_x);
    const int tr = tl - n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int py0 = ty * PTH, px0 = tx * PTW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;

#pragma unroll 1
    for (int nh = 0; nh < NH; ++nh) {
      f32x4 acc[SUBS][TT];
#pragma unroll
      for (int i = 0; i < SUBS; ++i)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        __builtin_amdgcn_sched_barrier(0);  // keep the k-steps apart: hoisting all B/A loads spills 160 VGPRs
        half8 bh[TT], bl[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const int f = (nh * KSTEPS + ks) * TT + tt;
          bh[tt] = *reinterpret_cast<const half8*>(wl + (2 * f) * WFRAG + lane * 8);
          bl[tt] = *reinterpret_cast<const half8*>(wl + (2 * f + 1) * WFRAG + lane * 8);
        }
#pragma unroll
        for (int ip = 0; ip < SUBS; ip += 2) {
          half8 ah[2], al[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            ah[d] = load_a(hiP, abase[ip + d] + koff[ks]);
            al[d] = load_a(loP, abase[ip + d] + koff[ks]);
          }
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[d], bh[tt], acc[ip + d][tt], 0, 0, 0);
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[d], bl[tt], acc[ip + d][tt], 0, 0, 0);
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[d], bh[tt], acc[ip + d][tt], 0, 0, 0);
        }
      }
      // BN + ReLU -> staged tile [pixel][channel]
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const int ch = 32 * nh + 16 * tt + li;
        const float ba = bn_a[ch], bb = bn_b[ch];
#pragma unroll
        for (int i = 0; i < SUBS; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = (SUBS * wave + i) * 16 + lg * 4 + r;
            if (m < MPIX) {
              const int cy = m / CTW, cx = m - cy * CTW;
              const bool inside = (unsigned)(cy0 + cy) < (unsigned)Hc && (unsigned)(cx0 + cx) < (unsigned)Wc;
              const float v = fmaxf(fmaf(acc[i][tt][r], ba, bb), 0.0f);
              stage[m * SC + ch] = inside ? v : 0.0f;
            }
          }
        }
      }
    }
    __syncthreads();  // conv tile staged; every wave is done reading the patch
    if (seq_next < nseq) commit();

    // 3x3 / stride-2 max pool, 64 channels in two passes of 32 (8 per wave)
    const int px = px0 + pplx;
    const bool col_live = valid && pplx < PTW && px < Wp;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int pch = 32 * pass + wave * 8 + pchl;
      float hm[CTH];
      const float* sp = stage + (2 * (pplx < PTW ? pplx : 0)) * SC + pch;
#pragma unroll
      for (int r = 0; r < CTH; ++r)
        hm[r] = fmaxf(fmaxf(sp[(r * CTW) * SC], sp[(r * CTW + 1) * SC]), sp[(r * CTW + 2) * SC]);
#pragma unroll
      for (int ply = 0; ply < PTH; ++ply) {
        const int py = py0 + ply;
        const bool live = col_live && py < Hp;
        const float v = fmaxf(fmaxf(hm[2 * ply], hm[2 * ply + 1]), hm[2 * ply + 2]);
        if (live && out) out[(((size_t)n * COUT + pch) * Hp + py) * Wp + px] = v;
        if (P) {  // lanes 8*plx .. 8*plx+7 hold the 8 channels of byte 4*pass + wave of pixel (ply, plx)
          const unsigned long long mask = __ballot(live && is_pos(v));
          if (pchl == 0 && pplx < PTW)
            bits[((buf * PTH + ply) * PTW + pplx) * 8 + 4 * pass + wave] = (uint8_t)(mask >> (8 * pplx));
        }
      }
    }
    prev_n = valid ? n : -1;
    prev_py0 = py0;
    prev_px0 = px0;
    buf ^= 1;
  }
  __syncthreads();
  flush_bits(buf ^ 1);
}

int launch_stem_lean(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                     int W, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  using namespace stem3;
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
  const long long want = cus;  // one 4-wave workgroup per CU: one wave per SIMD
  const unsigned grid = (unsigned)(ntiles < want ? ((ntiles + 7) / 8 * 8) : want);
  static bool attr_set[64] = {false};
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_lean_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(stem_lean_kernel, dim3(grid), dim3(NT), LDS_BYTES, stream, x, w, bn_a, bn_b, N, H, W,
                     Hc, Wc, Hp, Wp, tiles_y, tiles_x, per_xcd, out, P, M);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
