// TEST-ONLY since round 4 (libbnn_hip_legacy.so; never loaded by the product path): the round-2 stem kernel, kept as an
// independent implementation that tests/test_gpu_fused.py compares the default kernel (csrc/stem_rows.hip) with,
// bit for bit.  Until ABI 11 it rode in libbnn_hip.so behind BNN_HIP_STEM_STAGED.
// stem_split.hip — default arithmetic of the fused stem (see stem.hip for the layer and the exact mode):
// fp32 operands split into fp16 hi + lo, product = hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with
// fp32 accumulation (measured 3e-7 relative to an fp64 reference: the rounding class of an fp32 conv).
// Operand range: |x| and |w| below 65504 (images and conv weights are O(1)); inf/NaN inputs give NaN.
//
// What differs from a textbook implicit GEMM, and why (all measured on MI355X, tools/bench_stem.py):
//  * the split is done ONCE per input element when the patch is written to LDS (two fp16 planes), not
//    per use: the matrix phase then has no conversion VALU work at all (it was the bottleneck);
//  * K is laid out as 24 rows of 8: row = (c, ky), column = kx (kx = 7 and rows 21..23 carry zero
//    weights).  A lane's 8 consecutive k of the MFMA A operand are then 8 ADJACENT halves of the patch:
//    two ds_read2_b32 instead of 8 gathers + packing;
//  * B (weights) never goes through LDS: a wave owns 32 of the 64 output channels and keeps their
//    hi/lo fragments for all 6 k-steps in 96 VGPRs for the life of the persistent workgroup;
//  * 4 waves = 2 pixel groups (5 sub-tiles of 16 conv pixels each) x 2 channel halves; TWO such
//    workgroups per CU, each on its own 4 x 8 pooled tile, so that one's matrix phase can run beside the
//    other's fetch / pooling / stores (one 8-wave workgroup on an 8 x 7 tile: 329 us vs 306 us, round 2).
//    The accumulators go through in two passes (4 + 1 sub-tiles) to stay inside 256 VGPRs.
//    (Round-1 alternative that lost, 540 vs 440 us: two workgroups that SHARE a tile, one channel half
//    each — the patch is then fetched and split twice.)
//  * the next tile's patch is fetched into registers during the matrix phase; tiles are walked in
//    XCD-contiguous order;
//  * max-pool reads each staged conv value ~3x instead of 9x: a thread owns one pooled COLUMN of one
//    channel (9 row maxima -> 4 outputs); the sign bits of 8 channels are gathered with one ballot and
//    leave as whole 64-bit words one tile later (byte stores from several waves into one word are slow).
#include "../bnn_dev.h"

#ifndef BNN_STEM_ABL  // timing ablations only (wrong results): 1 matrix, 2 epilogue, 4 pooling, 8 fetch
#define BNN_STEM_ABL 0
#endif

namespace bnn {

namespace stem2 {
constexpr int CIN = 3, KS = 7, COUT = 64;
constexpr int KROWS = 24, KSTEPS = KROWS / 4;        // 6 k-steps of 32 (4 rows of 8)
// Two INDEPENDENT 4-wave workgroups per CU, each on a 4 x 8 pooled tile: the phases of one (fetch, matrix,
// BN, pooling + stores) are serialised by its barriers, but the CU interleaves them with the other
// workgroup's.  One 8-wave workgroup on an 8 x 7 tile: 329 us; this: 306 us (same GPU, batch 256).
constexpr int PTH = 4, PTW = 8;                      // pooled tile
constexpr int NT = 256, SUBS = 5;                    // threads; 16-pixel sub-tiles per wave
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1;  // conv tile 9 x 17 (pool halo included)
constexpr int MPIX = CTH * CTW;                      // 153
constexpr int ITH = 2 * CTH + 5;                     // 23 input rows
constexpr int ITWP = 2 * CTW + 6;                    // 39 input columns + 1 zero column (kx = 7)
// Patch rows are stored ROWH = 96 halves (48 dwords = 16 mod 32 banks) apart: the four k-rows a wave reads
// at once (lane>>4 -> consecutive ky) then fall into alternating halves of the 32 LDS banks, so the two
// 16-lane groups of each half-wave never collide.  (Dense rows of 18 dwords: +50 % LDS cycles on the
// A-operand reads, measured with SQ_LDS_BANK_CONFLICT.)
constexpr int ROWH = 96;
constexpr int ICHP = ITH * ROWH;                     // halves per channel plane
constexpr int NINP = CIN * ICHP;                     // halves per plane (13 KB)
constexpr int NROW = CIN * ITH;                      // 69 patch rows
constexpr int NPC = ITWP / 2;                        // 20 column pairs per row
// Staged conv tile: [conv pixel m][channel], row stride SC floats.  SC = 68 (= 4 mod 64) makes both sides
// conflict-free: the MFMA D layout writes (channel = lane&15, pixel = 4*(lane>>4)+r) -> bank lane&15 + 16*(lane>>4),
// the pooling threads read (channel = lane&7 [+8*wave], column = lane>>3) -> bank lane&7 + 8*(lane>>3).
// (The channel-major layout it replaces spent 55 % of the LDS-active cycles in bank conflicts: PMC.)
constexpr int SC = COUT + 4;
constexpr int NW = NT / 64, MG = NW / 2;             // waves; pixel groups
constexpr int PJ = COUT / (NW * 8);                  // pooling passes (8 channels per wave and pass)
constexpr int RSTEP = NT / NPC;                      // 12 rows per sweep (240 fetching threads)
constexpr int PER_T = (NROW + RSTEP - 1) / RSTEP;    // 6 column pairs per thread
constexpr int TT = 2, SPASS = 4;                     // channel tiles per wave; sub-tiles per accumulator pass
// LDS carve (bytes)
constexpr int OFF_HI = 0;
constexpr int OFF_LO = OFF_HI + ((NINP * 2 + 15) / 16) * 16;
constexpr int OFF_STAGE = OFF_LO + ((NINP * 2 + 15) / 16) * 16;
constexpr int OFF_BITS = OFF_STAGE + (MPIX + 1) * SC * 4;  // sign bytes of the tile: [2][32 pixels][8]
constexpr int LDS_BYTES = OFF_BITS + 2 * PTH * PTW * 8;
}  // namespace stem2

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

// Buffer addressing for the output streams: 128-bit descriptor in SGPRs + ONE per-lane byte offset (tile-invariant,
// computed once) + a wave-uniform byte offset per store in an SGPR (`soffset`).  With plain pointers every store cost
// a 64-bit multiply-add chain in the vector ALU (~7 instructions; the kernel's time outside the MFMAs is VALU issue).
// Dead lanes carry an out-of-range offset (the store is dropped); a null tensor is a descriptor with zero records.
#if defined(__HIP_DEVICE_COMPILE__)
using StemRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ StemRsrc stem_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ void stem_st(StemRsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void stem_st2(StemRsrc r, unsigned voff, unsigned soff, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}
#else  // the host pass of hipcc only parses these
struct StemRsrc {};
__device__ __forceinline__ StemRsrc stem_rsrc(const void*, unsigned) { return {}; }
__device__ __forceinline__ void stem_st(StemRsrc, unsigned, unsigned, float) {}
__device__ __forceinline__ void stem_st2(StemRsrc, unsigned, unsigned, u32x2) {}
#endif
constexpr unsigned kStemOOB = 0xFFFFFFF0u;  // beyond every descriptor's num_records

// HALF: plain fp16 operands, one MFMA per product (BNN_HIP_STEM_FP16: the "fp16 MFMA stem" of BASELINE config
// 5) — 1/3 of the matrix work, ~5e-4 relative error; the lo planes / fragments are then dead code.
template <bool HALF>
__global__ __launch_bounds__(stem2::NT, 2) void stem_split_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int N, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
    int tiles_x, int per_xcd, float* __restrict__ out, uint64_t* __restrict__ P,
    uint64_t* __restrict__ M, unsigned out_bytes, unsigned plane_bytes) {
  using namespace stem2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* hiP = reinterpret_cast<_Float16*>(lds_raw + OFF_HI);
  _Float16* loP = reinterpret_cast<_Float16*>(lds_raw + OFF_LO);
  float* stage = reinterpret_cast<float*>(lds_raw + OFF_STAGE);
  uint8_t* bits = lds_raw + OFF_BITS;  // double-buffered: tile t's words leave during tile t+1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int mg = wave % MG, nh = wave / MG;  // pixel group (SUBS sub-tiles), channel half

  // ---- once: B fragments (hi, lo) of this wave's 2 channel tiles x 6 k-steps, in registers.
  // MFMA 16x16x32 B operand: lane holds B[k = 8*lg + e][j = li], e = 0..7  ->  row 4*ks + lg, kx = e.
  half8 bh[KSTEPS][TT], bl[KSTEPS][TT];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int krow = 4 * ks + lg;
    const int c = krow / KS, ky = krow - c * KS;
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int o = 32 * nh + 16 * tt + li;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (krow < CIN * KS && e < KS) ? w[((size_t)(o * CIN + c) * KS + ky) * KS + e] : 0.0f;
        const _Float16 h = (_Float16)v;
        bh[ks][tt][e] = h;
        bl[ks][tt][e] = (_Float16)(v - (float)h);
      }
    }
  }
  // A operand: lane holds A[i = li][k = 8*lg + e] = patch[c][2*cy + ky][2*cx + e] of conv pixel
  // m = 16*sub + li.  Offsets in halves; everything is even, so reads are 4-byte aligned.
  int koff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int krow = 4 * ks + lg;
    const int c = krow / KS, ky = krow - c * KS;
    koff[ks] = krow < CIN * KS ? c * ICHP + ky * ROWH : 0;  // zero-weight rows: any valid address
  }
  int abase[SUBS];
#pragma unroll
  for (int i = 0; i < SUBS; ++i) {
    int m = (SUBS * mg + i) * 16 + li;
    if (m >= MPIX) m = MPIX - 1;
    const int cy = m / CTW, cx = m - cy * CTW;
    abase[i] = 2 * cy * ROWH + 2 * cx;
  }
  // BN constants of the accumulator layout (column = li -> channel 32*nh + 16*tt + li)
  float ba[TT], bb[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    ba[tt] = bn_a[32 * nh + 16 * tt + li];
    bb[tt] = bn_b[32 * nh + 16 * tt + li];
  }
  // fetch role: column pair `fpc` of patch rows frow0 + 28*u (row = c*39 + r)
  const int fpc = tid % NPC, frow0 = tid / NPC;
  const bool fetcher = tid < NPC * RSTEP;
  // pooling role: one pooled column (PTH outputs) of one channel, PJ times; 8 channels x 8 columns per wave and pass
  const int pchl = lane & 7, pplx = lane >> 3;  // channel within the wave's byte, pooled column (7 = idle)
  const int pch0 = wave * 8 + pchl;
  const StemRsrc r_out = stem_rsrc(out, out_bytes), r_P = stem_rsrc(P, plane_bytes), r_M = stem_rsrc(M, plane_bytes);
  // per-lane byte offsets of the output streams relative to the tile's (wave-uniform) origin
  const unsigned out_lane = (unsigned)((pch0 * Hp) * Wp + pplx) * 4u;          // channel pch0, pooled column pplx
  const unsigned flush_lane = (unsigned)((tid / PTW) * Wp + tid % PTW) * 8u;    // pixel (tid / PTW, tid % PTW) of a tile

  const int ntiles = N * tiles_y * tiles_x;
  const int nseq = per_xcd * 8;
  // Tile order: workgroup b sits on XCD b % 8 (observed placement, used for speed only).  Each XCD
  // walks ONE contiguous eighth of the tile list: x-neighbours (shared halo, shared output lines) meet
  // in the same L2 within a short time.
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
    const float* xl = xb + 2 * fpc;
    int fr = frow0;
    asm volatile("" : "+v"(fr));  // per-row offsets are recomputed per tile: hoisted, they only get spilled
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int R = fr + RSTEP * u;
      const int c = (R >= 2 * ITH) + (R >= ITH), r = R - c * ITH;
      const int goff = (c * H + r) * W;  // c*H*W + r*W
      const bool okr = fetcher && R < NROW && (unsigned)(iy0 + r) < (unsigned)H;
      nx0[u] = (okr && okc0) ? xl[goff] : 0.0f;
      nx1[u] = (okr && okc1) ? xl[goff + 1] : 0.0f;
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
        if constexpr (!HALF) reinterpret_cast<half2v*>(loP)[R * (ROWH / 2) + fpc] = l;
      }
    }
  };
  auto load_a = [&](const _Float16* plane, int off) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(plane) + (off >> 1);
    u32x4 v;
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    return __builtin_bit_cast(half8, v);
  };

  // sign words of the PREVIOUS tile: its waves left 8 bytes per pixel in LDS; PTH*PTW threads send them
  // as whole 64-bit words (byte stores from several waves into one word cost ~80 us at batch 256)
  int prev_n = -1, prev_py0 = 0, prev_px0 = 0, buf = 0;
  auto flush_bits = [&](int b) {
    if (!(BNN_STEM_ABL & 32) && prev_n >= 0 && tid < PTH * PTW) {  // wave 0 only
      const int ply = tid / PTW, plx = tid - ply * PTW;
      const bool ok = prev_py0 + ply < Hp && prev_px0 + plx < Wp;
      const unsigned soff = (unsigned)((prev_n * Hp + prev_py0) * Wp + prev_px0) * 8u;
      const u32x2 word = *reinterpret_cast<const u32x2*>(bits + (b * PTH * PTW + tid) * 8);
      stem_st2(r_P, ok ? flush_lane : kStemOOB, soff, word);
      stem_st2(r_M, ok ? flush_lane : kStemOOB, soff, u32x2{0u, 0u});  // nothing is negative after ReLU
    }
  };

#ifdef BNN_STEM_TIMING  // per-phase cycle sums of every wave's lane 0, dumped over the M plane (debug builds)
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define STEM_T(k) { const unsigned long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; }
#else
#define STEM_T(k)
#endif
  int seq = blockIdx.x;
  if (seq < nseq) { fetch(tile_of(seq)); commit(); }
  for (; seq < nseq; seq += gridDim.x) {
    const int tile = tile_of(seq);
    const bool valid = tile < ntiles;  // workgroup-uniform
    STEM_T(5)
    __syncthreads();                   // this tile's patch is in LDS; `stage` is free again
    STEM_T(0)
    flush_bits(buf ^ 1);
    const int seq_next = seq + gridDim.x;
#if !(BNN_STEM_ABL & 8)
    if (seq_next < nseq) fetch(tile_of(seq_next));  // global loads fly during the matrix phase
#endif
    const int tl = valid ? tile : 0;
    const int n = tl / (tiles_y * tiles_x);
    const int tr = tl - n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int py0 = ty * PTH, px0 = tx * PTW;        // pooled origin
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // conv origin (pool pad 1)

    // ---- implicit GEMM: SUBS sub-tiles x 2 channel tiles x 6 k-steps x (lo*hi + hi*lo + hi*hi).
    // Two sub-tiles at a time, product-type major: 4 independent accumulators between two MFMAs
    // that touch the same one.
    // The sub-tiles go through in passes of at most SPASS (accumulators of one pass live at a time).
    // ---- BN + ReLU, conv tile -> LDS.  D layout: column = li (channel), row = 4*lg + r (pixel).
#pragma unroll
    for (int s0 = 0; s0 < SUBS; s0 += SPASS) {
      if (s0) __builtin_amdgcn_sched_barrier(0);  // one pass's accumulators at a time
      f32x4 acc[SPASS][TT];
#pragma unroll
      for (int i = 0; i < SPASS; ++i)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !(BNN_STEM_ABL & 1)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int ip = 0; ip < SPASS; ip += 2) {
          if (s0 + ip >= SUBS) continue;
          half8 ah[2], al[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            if (s0 + ip + d >= SUBS) continue;
            ah[d] = load_a(hiP, abase[s0 + ip + d] + koff[ks]);
            if constexpr (!HALF) al[d] = load_a(loP, abase[s0 + ip + d] + koff[ks]);
          }
          if constexpr (!HALF) {
#pragma unroll
            for (int d = 0; d < 2 && s0 + ip + d < SUBS; ++d)
#pragma unroll
              for (int tt = 0; tt < TT; ++tt)
                acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[d], bh[ks][tt], acc[ip + d][tt], 0, 0, 0);
#pragma unroll
            for (int d = 0; d < 2 && s0 + ip + d < SUBS; ++d)
#pragma unroll
              for (int tt = 0; tt < TT; ++tt)
                acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[d], bl[ks][tt], acc[ip + d][tt], 0, 0, 0);
          }
#pragma unroll
          for (int d = 0; d < 2 && s0 + ip + d < SUBS; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[d], bh[ks][tt], acc[ip + d][tt], 0, 0, 0);
        }
      }
#endif
#if !(BNN_STEM_ABL & 2)
      // three quarters of the tiles lie entirely inside the conv output: no per-pixel range tests there
      const bool interior = cy0 >= 0 && cx0 >= 0 && cy0 + CTH <= Hc && cx0 + CTW <= Wc;  // workgroup-uniform
      float* sdst = stage + ((SUBS * mg) * 16 + lg * 4) * SC + 32 * nh + li;
      if (interior) {
        // BN of two accumulator rows at a time: one v_pk_fma_f32 instead of two v_fma_f32.  The ReLU is applied AFTER
        // the max-pool (max and relu commute exactly; padding positions are staged as 0): 8 clamps per thread there
        // instead of 40 here
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < SPASS; ++i)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
              if (s0 + i >= SUBS) continue;
              const f2 y = __builtin_elementwise_fma(f2{acc[i][tt][rp], acc[i][tt][rp + 1]}, f2{ba[tt], ba[tt]},
                                                     f2{bb[tt], bb[tt]});
#pragma unroll
              for (int q = 0; q < 2; ++q)
                if ((SUBS * mg + s0 + i) * 16 + lg * 4 + rp + q < MPIX)
                  sdst[((s0 + i) * 16 + rp + q) * SC + 16 * tt] = y[q];
            }
      } else {
#pragma unroll
        for (int i = 0; i < SPASS; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = (SUBS * mg + s0 + i) * 16 + lg * 4 + r;
            if (s0 + i < SUBS && m < MPIX) {
              const int cy = m / CTW, cx = m - cy * CTW;
              const bool inside = (unsigned)(cy0 + cy) < (unsigned)Hc && (unsigned)(cx0 + cx) < (unsigned)Wc;
#pragma unroll
              for (int tt = 0; tt < TT; ++tt) {
                const float v = fmaf(acc[i][tt][r], ba[tt], bb[tt]);
                // positions outside the conv output are MaxPool padding: 0 never beats a ReLU output
                sdst[((s0 + i) * 16 + r) * SC + 16 * tt] = inside ? v : 0.0f;
              }
            }
          }
        }
      }
#endif
    }
    STEM_T(2)
    __syncthreads();  // conv tile staged; every wave is done reading the patch
    STEM_T(3)
#if !(BNN_STEM_ABL & 8)
    if (seq_next < nseq) commit();  // next patch: registers -> LDS (fp16 hi/lo), overlaps the pooling
#endif
    STEM_T(4)

#if !(BNN_STEM_ABL & 4)
    // ---- 3x3 / stride-2 max pool: row maxima of the thread's 3 conv columns, then PTH column maxima
    const int px = px0 + pplx;
    const bool col_live = valid && pplx < PTW && px < Wp;
    const unsigned out_voff = col_live ? out_lane : kStemOOB;
    const unsigned out_tile = (unsigned)((n * COUT * Hp + py0) * Wp + px0) * 4u;  // wave-uniform
#pragma unroll
    for (int pj = 0; pj < PJ; ++pj) {
      const int pch = pch0 + NW * 8 * pj;
      float hm[CTH];
      {
        const float* sp = stage + (2 * (pplx < PTW ? pplx : 0)) * SC + pch;
#pragma unroll
        for (int r = 0; r < CTH; ++r)
          hm[r] = fmaxf(fmaxf(sp[(r * CTW) * SC], sp[(r * CTW + 1) * SC]), sp[(r * CTW + 2) * SC]);
      }
#pragma unroll
      for (int ply = 0; ply < PTH; ++ply) {
        const int py = py0 + ply;
        const bool live = col_live && py < Hp;  // py < Hp is workgroup-uniform
        const float v = fmaxf(fmaxf(fmaxf(hm[2 * ply], hm[2 * ply + 1]), hm[2 * ply + 2]), 0.0f);  // max-pool, then ReLU
        if (py < Hp)  // workgroup-uniform
          stem_st(r_out, out_voff, out_tile + (unsigned)((NW * 8 * pj * Hp + ply) * Wp) * 4u, v);
        if (P != nullptr && !(BNN_STEM_ABL & 16)) {  // lanes 8*plx .. 8*plx+7 hold the 8 channels of one byte of pixel (ply, plx)
          const unsigned long long mask = __ballot(live && is_pos(v));
          if (pchl == 0 && pplx < PTW)
            bits[((buf * PTH + ply) * PTW + pplx) * 8 + wave + NW * pj] = (uint8_t)(mask >> (8 * pplx));
        }
      }
    }
#endif
    prev_n = valid ? n : -1;
    prev_py0 = py0;
    prev_px0 = px0;
    buf ^= 1;
  }
  __syncthreads();
  flush_bits(buf ^ 1);
#ifdef BNN_STEM_TIMING
  __syncthreads();
  if (lane == 0 && M) {  // [workgroup][wave][6]: barrier0 wait, fetch-issue+matrix, epilogue, barrier1 wait, commit, pool
    for (int k = 0; k < 6; ++k) M[((size_t)blockIdx.x * 8 + wave) * 6 + k] = tph[k];
  }
#endif
}

template <bool HALF>
static int launch_stem_split_t(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                               int W, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  using namespace stem2;
  const int Hc = (H + 6 - KS) / 2 + 1, Wc = (W + 6 - KS) / 2 + 1;
  const int Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW;
  const long long ntiles = (long long)N * tiles_y * tiles_x;
  const int cus = current_device_cus();
  const int per_xcd = (int)((ntiles + 7) / 8);
  const long long want = (long long)cus * (8 / NW);  // 8 waves (two workgroups) per CU
  const unsigned grid = (unsigned)(ntiles < want ? ((ntiles + 7) / 8 * 8) : want);
  // > 64 KB of dynamic LDS needs the opt-in: per device and per kernel, so it is set on every launch (no mutable
  // global state in a re-entrant API; the call is a table write on the host)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_split_kernel<HALF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          LDS_BYTES) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  // byte sizes of the output streams (the C-ABI caps every tensor below 2^32 bytes)
  const unsigned out_bytes = (unsigned)((long long)N * COUT * Hp * Wp * 4);
  const unsigned plane_bytes = (unsigned)((long long)N * Hp * Wp * 8);
  hipLaunchKernelGGL(stem_split_kernel<HALF>, dim3(grid), dim3(NT), LDS_BYTES, stream, x, w, bn_a, bn_b, N, H,
                     W, Hc, Wc, Hp, Wp, tiles_y, tiles_x, per_xcd, out, P, M, out_bytes, plane_bytes);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_stem_split(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                      int W, int half, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  return half ? launch_stem_split_t<true>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream)
              : launch_stem_split_t<false>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream);
}

}  // namespace bnn
