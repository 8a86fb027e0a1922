// hblock.hip — the hierarchical block (bnn/models/layers/hierarchical_block.py:38-60) in ONE launch.
//
//     o1 = conv1(sign(act1(bn1(x))))     3x3, C_in -> C/2
//     o2 = conv2(sign(act2(bn2(o1))))    3x3, C/2  -> C/4
//     o3 = conv3(sign(act3(bn3(o2))))    3x3, C/4  -> C/4
//     y  = cat(o1, o2, o3) + shortcut(x)
//
// Launch by launch (bconv_sgpr_kernel x 3 + a packing pass in front) this block is 4 launches of 13-27 us for 16 us of
// integer-ALU work at batch 128 (profiles/r06_c5_kernel_roofline_before.md): the widths are small (O = 16 .. 256), the
// intermediate sign planes travel through HBM, C/2 and C/4 input channels below 64 are padded to 64-channel groups and
// every launch pays its own ramp and tail.  Here a workgroup owns a REGION — whole images, or a band of rows of one
// image with its halo — and runs the three convolutions back to back with every sign plane in LDS:
//
//   phase 0   the block's input planes (sign(act1(bn1(x))), P plane: activations out of a ReLU) HBM -> LDS, band + 3 halo rows
//   phase 1   conv1 on band + 2 halo rows: fp32 y[:, 0:C/2] = o1 + shortcut (band rows only);  sign(act2(bn2(o1))) -> LDS
//   phase 2   conv2 on band + 1 halo row:  fp32 y[:, C/2:3C/4];                                sign(act3(bn3(o2))) -> LDS
//   phase 3   conv3 on the band:           fp32 y[:, 3C/4:C]
//   every phase also leaves sign(act1'(bn1'(y))) of its channels — the NEXT block's input planes — in LDS;
//   phase 4   copies them out (coalesced 8-byte stores).
//
// The inner loop is the one of bconv_sgpr_kernel: lane = output pixel, the pixel's receptive field for one chunk of
// input channels in VGPRs (read from LDS), weights wave-uniform through the scalar cache into SGPRs, v_and_b32 +
// v_bcnt_u32_b32 (bconv_core.h: stream_weights).  Planes are dense: a cell holds ceil(C_in / 32) words (C_in = 16 / 32:
// ONE word per tap, not two), and a convolution of 16 output channels runs 16 channels, not a padded block of 32.
// Same integers and the same float operations, in the same order, as the launch-by-launch form: bit-identical y and planes.
//
// MODE 2 (the last block of a stage in front of `AvgPool2d(2, 2)` + a block with a shortcut convolution): y is only ever
// pooled and binarised — by the next block's bn1 -> ReLU and by its shortcut's BatchNorm (hierarchical_block.py:39, 30-36) —
// so the launch writes THOSE planes at half resolution and no fp32 tensor at all.  Lanes then walk the pixels window by
// window (4 lanes = one 2 x 2 window: the pool is three adds across a quad), in every convolution alike, so that the
// completion counters keep meaning "pixel group g of the previous convolution".
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "bconv_core.h"

namespace bnn {

struct HbPhase {
  int O;              // output channels
  int nchunk;         // chunks of CWC words of its input cell
  int halo;           // rows of its output domain beyond the band on each side (2, 1, 0)
  int c_off;          // first channel of its slice of the block's output
  int ppu;            // passes per unit (one load of the field)
  int npass, upg;     // passes (O / NACC), units per pixel group
  unsigned w_off;     // words: its weights in W, [pass][chunk][j][tap][CWC]
  unsigned a_off, pa_off, pb_off;  // floats: alpha / the next phase's BatchNorm in the constants
  unsigned lds_in;    // byte offset of its input plane, [chunk][cell][CWC]
  int rows_in;        // rows of that plane's slab per image: band rows + 2 * (halo + 1)
  int ncell_in;       // cells of the slab: G * rows_in * WP
  uint32_t m_upg;
  int s_upg;
};

struct HbGeo {
  int N, H, W, C;     // images, image size, block width
  int G, BR, nbi;     // images per region | rows per band, bands per image
  int WP;             // row pitch of the slabs in cells: W + 2
  int ng_in, cw_in;   // 64-channel groups / 32-bit words per pixel of the block's input planes
  HbPhase ph[3];
  unsigned lds_out;   // next block's planes: [group][cell][2 words], cell = (image, band row, column)
  int ncell_out;
  unsigned lds_done;  // completion counters: one per pixel group of conv1's, then of conv2's domain
  unsigned lds16;     // LDS bytes / 16
  unsigned na_off, nb_off;  // floats: the next block's bn1
  uint32_t m_hw, m_w;
  int s_hw, s_w;
  unsigned f32_bytes;  // bytes of the fp32 tensors (residual, out): the range of their descriptors
  int ncell_pool;      // MODE 2: cells of one pooled plane set of the region, G * (BR / 2) * (W / 2)
  uint32_t m_wc;       // phase 0: division by W (m_w divides by W / 2 in MODE 2)
  int s_wc;
};

// DS: the block's shortcut is a binary 1 x 1 convolution of ANOTHER binarisation of its input (the first block of a stage:
// hierarchical_block.py:30-36, BatchNorm -> sign -> conv1x1) and is computed here, per pass, from that tensor's two sign
// planes — no shortcut launch, no fp32 shortcut tensor written and read back.
struct HbDs {
  const uint64_t* P;   // [N, ceil(C_in / 64), H, W]: the shortcut input is positive / negative
  const uint64_t* M;
  const uint32_t* W;   // [C][cw_in] words, bit = the weight is +1 (hblock_ds_pack_weight_kernel)
  const float* A;      // [C]: alpha of the 1 x 1 convolution
};

namespace {

extern __shared__ __attribute__((aligned(16))) unsigned char hb_smem[];

__device__ __forceinline__ uint32_t hb_uniform(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

__device__ __forceinline__ uint32_t hb_ticket(uint32_t* ctr, int lane) {
  uint32_t t = 0;
  if (lane == 0) t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return hb_uniform(t);
}

template <int N>
__device__ __forceinline__ void hb_lds_words(const uint32_t* base, unsigned word_off, uint32_t* dst) {
  if constexpr (N == 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + word_off);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  } else if constexpr (N == 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(base + word_off);
    dst[0] = v.x; dst[1] = v.y;
  } else {
    dst[0] = base[word_off];
  }
}

#ifdef HB_TIMING  // variant builds only (tools/exp_hblock_timing.py): per-wave cycle stamps of the phases
__device__ unsigned long long bnn_hb_dbg[8 * 16 * 4096];
#define HB_NOW() __builtin_amdgcn_s_memtime()
#else
#define HB_NOW() 0ull
#endif

// N consecutive wave-uniform floats (32-byte aligned) as s_load_dwordx8 pieces.
template <int N>
__device__ __forceinline__ void hb_consts(const float* __restrict__ src, float (&dst)[N]) {
  static_assert(N % 8 == 0, "whole dwordx8 pieces");
  struct alignas(32) F8 { float v[8]; };
#pragma unroll
  for (int i = 0; i < N / 8; ++i) {
    const F8 t = *reinterpret_cast<const F8*>(__builtin_assume_aligned(src + 8 * i, 32));
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[8 * i + e] = t.v[e];
  }
}

// ((v of lane 0 + v of lane 1) + v of lane 2) + v of lane 3 of the lane's quad, in every lane: one DPP move and three adds
// with a DPP operand
template <int CTRL>
__device__ __forceinline__ float hb_quad(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
#else
  return v;
#endif
}
__device__ __forceinline__ float hb_quad_sum(float v) {
  float s = hb_quad<0x00>(v);
  s = hb_quad<0x55>(v) + s;
  s = hb_quad<0xAA>(v) + s;
  return hb_quad<0xFF>(v) + s;
}

// N consecutive wave-uniform 32-bit words (32-byte aligned), the same way.
template <int N>
__device__ __forceinline__ void hb_words(const uint32_t* __restrict__ src, uint32_t (&dst)[N]) {
  static_assert(N % 8 == 0, "whole dwordx8 pieces");
  struct alignas(32) U8 { uint32_t v[8]; };
#pragma unroll
  for (int i = 0; i < N / 8; ++i) {
    const U8 t = *reinterpret_cast<const U8*>(__builtin_assume_aligned(src + 8 * i, 32));
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[8 * i + e] = t.v[e];
  }
}

constexpr int HB_NONE = 0, HB_NEXT = 1, HB_POOL = 2;  // what a launch leaves for the next block
constexpr unsigned kOob = 0xFFFFFFF0u;  // a byte offset beyond every descriptor: loads return 0, stores are dropped
#ifndef HB_MINW  // waves per SIMD the kernel is register-allocated for (8: two 16-wave workgroups per CU)
#define HB_MINW 4
#endif

// The output domain of phase K in this region (wave-uniform): rows [ra, rb) of kk images, 64-pixel groups.
struct HbDom {
  int ra, nrows, npix, npg;
  int rlo, rhi;   // the rows that exist: [rlo, rhi) (POOL: [ra, ra + nrows) is that range widened to whole windows)
};
template <bool POOL>
__device__ __forceinline__ HbDom hb_domain(const HbGeo& g, int K, int kk, int y0, int rows) {
  HbDom d;
  d.rlo = max(0, y0 - g.ph[K].halo);
  d.rhi = min(g.H, y0 + rows + g.ph[K].halo);
  d.ra = POOL ? (d.rlo & ~1) : d.rlo;
  d.nrows = (POOL ? ((d.rhi + 1) & ~1) : d.rhi) - d.ra;  // (kk > 1: whole images, nrows == H)
  d.npix = kk * d.nrows * g.W;
  d.npg = (d.npix + 63) >> 6;
  return d;
}
// pixel j of a domain -> (image, row, column).  Linear: row-major.  POOL: window-major, 4 consecutive pixels = one
// 2 x 2 window (g.m_hw / g.m_w then divide by the WINDOWS of an image / of a row).
template <bool POOL>
__device__ __forceinline__ void hb_pixel(const HbGeo& g, const HbDom& d, int kk, int j, int& img, int& row, int& col) {
  if constexpr (POOL) {
    const int q = j >> 2, sub = j & 3, wpr = g.W >> 1;
    int wrem = q;
    img = 0;
    if (kk > 1) {
      img = (int)fast_div((uint32_t)q, g.m_hw, g.s_hw);
      wrem = q - imul<true>(img, (g.H >> 1) * wpr);
    }
    const int wy = (int)fast_div((uint32_t)wrem, g.m_w, g.s_w);
    row = d.ra + 2 * wy + (sub >> 1);
    col = 2 * (wrem - imul<true>(wy, wpr)) + (sub & 1);
  } else {
    int rem = j;
    img = 0;
    if (kk > 1) {
      img = (int)fast_div((uint32_t)j, g.m_hw, g.s_hw);
      rem = j - imul<true>(img, g.H * g.W);
    }
    const int rowl = (int)fast_div((uint32_t)rem, g.m_w, g.s_w);
    col = rem - imul<true>(rowl, g.W);
    row = d.ra + rowl;
  }
}

// Wait until the planes under the receptive fields of pixel group `pg` of phase K (>= 1) are complete: every pass of
// every pixel group of phase K - 1 that holds a row of the hull [first pixel's row - 1, last pixel's row + 1].
// The phases are not separated by barriers: a unit only ever waits for units with smaller tickets, which running
// waves hold (csrc/bconv_fly.hip uses the same argument).
template <bool POOL>
__device__ __forceinline__ void hb_wait_inputs(const HbGeo& g, int K, const HbDom& d, const HbDom& dp, int pg, int kk,
                                               const uint32_t* done_prev, int lane) {
  const int j0 = pg << 6, j1 = min(j0 + 63, d.npix - 1);
  int i0, i1, row0, row1, c0, c1;
  hb_pixel<POOL>(g, d, kk, j0, i0, row0, c0);
  hb_pixel<POOL>(g, d, kk, j1, i1, row1, c1);
  const int per_img = dp.nrows * g.W;
  // (POOL: a window row = 2 * W consecutive pixels of the previous domain, whose first row is even like this one's)
  const int above = max(row0 - 1, dp.rlo) - dp.ra, below = min(row1 + 1, dp.rhi - 1) - dp.ra;
  const int lo = (i0 * per_img + (POOL ? (above >> 1) * 2 : above) * g.W) >> 6;
  const int hi = (i1 * per_img + (POOL ? (below >> 1) * 2 + 1 : below) * g.W + g.W - 1) >> 6;
  const uint32_t want = (uint32_t)g.ph[K - 1].npass;
  for ([[maybe_unused]] unsigned idle = 0;; ++idle) {
    [[maybe_unused]] bool missing = false;
    for (int i = lo + lane; i <= hi; i += 64)
      missing |= __hip_atomic_load(&done_prev[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want;
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
    __builtin_amdgcn_s_sleep(4);
    if (idle > (1u << 24)) __builtin_trap();  // seconds of idling: a lost update — abort loudly rather than hang
#else
    break;
#endif
  }
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}

// One unit of one convolution of the block: passes p0 .. p0 + np - 1 of pixel group `pg`.
//   CWC / MULTI: words per (chunk, cell) of its input plane; several chunks
//   K: 0, 1, 2;  CWCN: CWC of the next phase (layout of the plane this one writes; unused for K == 2)
//   MODE: HB_NEXT — the next block's input planes are wanted (sign(act(bn1'(y)));  HB_POOL — those of AvgPool2d(2)(y) for
//         its bn1 (P) and its shortcut's BatchNorm (P and M), and no fp32 output.  Kp = [lane of the quad][a | b][C]: lane 0
//         (a1 / 4, b1), lane 1 (a2 / 4, b2), lane 2 (-a2 / 4, -b2) — each lane of a window tests ONE plane's bit,
//         fmaf(window sum, a, b) > 0: the division by 4 is exact in the scale, the negative plane is the positive one of
//         the negated affine
template <int CWC, bool MULTI, int K, int CWCN, int MODE, int DSW = 0>
__device__ __forceinline__ void hb_unit(const uint32_t* __restrict__ Wt, const float* __restrict__ Kc,
                                        const float* __restrict__ Kp, const float* __restrict__ res, float* __restrict__ out,
                                        const HbDs& ds, const HbGeo& g,
                                        unsigned char* smem, const HbDom& d, uint32_t* done, int pg, int p0, int np, int n0,
                                        int kk, int y0, int rows, int lane) {
  constexpr int NW = 9 * CWC;
  constexpr int NACC = CWC == 1 ? 16 : 8;  // channels per pass: weight runs of NACC * NW words are whole 64-byte lines
  constexpr bool LAST = K == 2;
  constexpr bool NEXT = MODE == HB_NEXT, POOL = MODE == HB_POOL;
  using f2 = __attribute__((ext_vector_type(2))) float;
  const HbPhase& ph = g.ph[K];
  const int hw = g.H * g.W;
  const uint32_t* pin = reinterpret_cast<const uint32_t*>(smem + ph.lds_in);
  const BufRsrc rres = make_rsrc_sized(res, g.f32_bytes), rout = make_rsrc_sized(out, g.f32_bytes);
  // the lane's pixel (lanes past the domain's last pixel copy it: same values to the same places)
  const int j = min((pg << 6) + lane, d.npix - 1);
  int img, row, col;
  hb_pixel<POOL>(g, d, kk, j, img, row, col);
  // POOL: a domain widened to whole windows has a row that does not exist (its lanes compute on a copy of the nearest
  // one and store nothing); lanes past the domain's end would pool four copies of one pixel: they store nothing either
  [[maybe_unused]] const bool exists = !POOL || (row >= d.rlo && row < d.rhi && (pg << 6) + lane < d.npix);
  if constexpr (POOL) row = min(max(row, d.rlo), d.rhi - 1);
  const bool interior = row >= y0 && row < y0 + rows;  // a band row: its fp32 values and next-block bits are this region's
  const unsigned lane_off =
      interior ? (unsigned)(imul<true>(imul<true>(n0 + img, g.C), hw) + imul<true>(row, g.W) + col) * 4u : kOob;
  // top-left tap of the receptive field in the phase's input slab (slab row 0 = image row y0 - halo - 1, column 0 = -1)
  const unsigned cell0 = (unsigned)(imul<true>(imul<true>(img, ph.rows_in) + (row - y0 + ph.halo), g.WP) + col);
  uint32_t pr[NW], mr[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) mr[i] = 0u;
  int nz = 0;
  if constexpr (!MULTI) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
      hb_lds_words<CWC>(pin, (cell0 + (unsigned)((t / 3) * g.WP + (t % 3))) * CWC, &pr[t * CWC]);
    nz = count_nonzero<NW, true>(pr, mr, 0);
  }
  // where this lane's bits go: the next phase's input plane / the next block's planes
  [[maybe_unused]] unsigned celln = 0u, cello = 0u;
  if constexpr (!LAST) {
    const HbPhase& pn = g.ph[K < 2 ? K + 1 : 2];
    celln = (unsigned)(imul<true>(imul<true>(img, pn.rows_in) + (row - y0 + pn.halo + 1), g.WP) + col + 1);
  }
  if constexpr (NEXT) cello = (unsigned)(imul<true>(imul<true>(img, g.BR) + (row - y0), g.W) + col);
  if constexpr (POOL)   // the window's cell in a pooled plane set
    cello = (unsigned)(imul<true>(imul<true>(img, g.BR >> 1) + ((row - y0) >> 1), g.W >> 1) + (col >> 1));
  // DSW > 0: the two sign planes of the shortcut's input at this lane's pixel (DSW words each) and how many are non-zero
  [[maybe_unused]] uint32_t dsp[DSW > 0 ? DSW : 1], dsm[DSW > 0 ? DSW : 1];
  [[maybe_unused]] int dsnz = 0;
  if constexpr (DSW > 0) {
    const size_t pix = (size_t)imul<true>(row, g.W) + (size_t)col;
#pragma unroll
    for (int q = 0; q < DSW / 2; ++q) {
      const size_t at = ((size_t)(n0 + img) * (DSW / 2) + q) * hw + pix;
      const uint64_t pv = interior ? ds.P[at] : 0ull, mv = interior ? ds.M[at] : 0ull;
      dsp[2 * q] = (uint32_t)pv; dsp[2 * q + 1] = (uint32_t)(pv >> 32);
      dsm[2 * q] = (uint32_t)mv; dsm[2 * q + 1] = (uint32_t)(mv >> 32);
    }
#pragma unroll
    for (int w = 0; w < DSW; ++w) dsnz += __builtin_popcount(dsp[w] | dsm[w]);
  }
#pragma unroll 1
  for (int ps = 0; ps < np; ++ps) {
    const int o0 = (p0 + ps) * NACC;
    // the pass's shortcut values: requested before the popcount loop, they land under it
    float resv[NACC];
    if constexpr (DSW > 0) {
      // ... or computed: dot = 2 * (agreeing non-zero inputs) - (non-zero inputs), value = fmaf(alpha, dot, 0) — the
      // integer and the one rounding of bnn_hip_bconv2d's plain epilogue (bconv_core.h)
      uint32_t wd[NACC * DSW];
      hb_words<NACC * DSW>(ds.W + (size_t)(ph.c_off + o0) * DSW, wd);
      float da[NACC];
      hb_consts<NACC>(ds.A + ph.c_off + o0, da);
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        int agree = 0;
#pragma unroll
        for (int w = 0; w < DSW; ++w) {
          const uint32_t wv = wd[i * DSW + w];
          agree += __builtin_popcount((dsp[w] & wv) | (dsm[w] & ~wv));
        }
        resv[i] = __builtin_fmaf(da[i], (float)(2 * agree - dsnz), 0.0f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NACC; ++i) resv[i] = buf_ld(rres, lane_off, (unsigned)(ph.c_off + o0 + i) * (unsigned)hw * 4u);
    }
    [[maybe_unused]] float qa[NACC], qb[NACC];
    if constexpr (POOL) {  // this lane's affine of the pass's channels (4 distinct addresses per wave)
      const float* kq = Kp + (size_t)((lane & 3) * 2 * g.C + ph.c_off + o0);
#pragma unroll
      for (int i = 0; i < NACC; i += 4) {
        const float4 a4 = *reinterpret_cast<const float4*>(kq + i), b4 = *reinterpret_cast<const float4*>(kq + g.C + i);
        qa[i] = a4.x; qa[i + 1] = a4.y; qa[i + 2] = a4.z; qa[i + 3] = a4.w;
        qb[i] = b4.x; qb[i + 1] = b4.y; qb[i + 2] = b4.z; qb[i + 3] = b4.w;
      }
    }
    int acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (int)kCountSeed;
    const uint32_t* wrun = Wt + ph.w_off + (size_t)(p0 + ps) * ph.nchunk * (NACC * NW);
    if constexpr (MULTI) {
      for (int ch = 0; ch < ph.nchunk; ++ch) {
        const unsigned cbase = (unsigned)(ch * ph.ncell_in) + cell0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          hb_lds_words<CWC>(pin, (cbase + (unsigned)((t / 3) * g.WP + (t % 3))) * CWC, &pr[t * CWC]);
        if (ps == 0) nz = count_nonzero<NW, true>(pr, mr, nz);
        stream_weights<NW, NACC, true, false, false>(wrun + (size_t)ch * (NACC * NW), pr, mr, acc);
      }
    } else {
      stream_weights<NW, NACC, true, true, false>(wrun, pr, mr, acc, (int)kCountSeed);
    }
    // epilogue: the float operations of bconv_core.h epilogue<EP_HB> (alpha, late residual, next BatchNorm in front of
    // the sign) and of pack_act.hip's bn_act_pack (the next block's bn1 on y), two channels per packed instruction
    const f2 dscale = {2.0f, 2.0f}, doff = {-(float)nz, -(float)nz};
    // the pass's per-channel constants: NACC consecutive floats each, fetched as whole 32-byte scalar loads (the layout
    // keeps every run 32-byte aligned: hblock_layout)
    float ka[NACC], kpa[NACC], kpb[NACC], kna[NACC], knb[NACC];
    hb_consts<NACC>(Kc + ph.a_off + o0, ka);
    if constexpr (!LAST) {
      hb_consts<NACC>(Kc + ph.pa_off + o0, kpa);
      hb_consts<NACC>(Kc + ph.pb_off + o0, kpb);
    }
    if constexpr (NEXT) {
      hb_consts<NACC>(Kc + g.na_off + ph.c_off + o0, kna);
      hb_consts<NACC>(Kc + g.nb_off + ph.c_off + o0, knb);
    }
    [[maybe_unused]] float pvi[NACC];
    [[maybe_unused]] float pvn[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i += 2) {
      const int co = ph.c_off + o0 + i;
      const f2 cnt = f2{__int_as_float(acc[i]), __int_as_float(acc[i + 1])} - f2{8388608.0f, 8388608.0f};
      const f2 dot = __builtin_elementwise_fma(cnt, dscale, doff);
      const f2 ov = __builtin_elementwise_fma(f2{ka[i], ka[i + 1]}, dot, f2{0.0f, 0.0f});
      if constexpr (!LAST) {
        const f2 v = __builtin_elementwise_fma(ov, f2{kpa[i], kpa[i + 1]}, f2{kpb[i], kpb[i + 1]});
        pvi[i] = v.x;
        pvi[i + 1] = v.y;
      }
      const f2 y = ov + f2{resv[i], resv[i + 1]};
      if constexpr (!POOL) {
        buf_st(rout, lane_off, (unsigned)co * (unsigned)hw * 4u, y.x);
        buf_st(rout, lane_off, (unsigned)(co + 1) * (unsigned)hw * 4u, y.y);
      } else {
        // AvgPool2d(2, 2) as ATen sums a window (pack_act.hip: avgpool2_bn_pack2): ((y00 + y01) + y10) + y11, every lane
        // of the quad with the whole sum; then this lane's BatchNorm branch of the next block
        const f2 v = __builtin_elementwise_fma(f2{hb_quad_sum(y.x), hb_quad_sum(y.y)}, f2{qa[i], qa[i + 1]}, f2{qb[i], qb[i + 1]});
        pvn[i] = v.x;
        pvn[i + 1] = v.y;
      }
      if constexpr (NEXT) {
        const f2 v = __builtin_elementwise_fma(y, f2{kna[i], kna[i + 1]}, f2{knb[i], knb[i + 1]});
        pvn[i] = v.x;
        pvn[i + 1] = v.y;
      }
    }
    if constexpr (NEXT) {
      uint32_t bits = 0u;
#pragma unroll
      for (int i = 0; i < NACC; ++i) bits = shift_in(bits, is_pos(pvn[i]));
      bits = __builtin_bitreverse32(bits) >> (32 - NACC);
      if (interior) {
        const int co0 = ph.c_off + o0;
        const unsigned widx = ((unsigned)((co0 >> 6) * g.ncell_out) + cello) * 2u + (unsigned)((co0 & 63) >> 5);
        unsigned char* dst = smem + g.lds_out + widx * 4u + (unsigned)((co0 & 31) >> 3);
        if constexpr (NACC == 16) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)bits;
        else *dst = (uint8_t)bits;
      }
    }
    if constexpr (POOL) {
      // three pieces per window — bn1: positive (its ReLU: no minus plane); shortcut: positive, negative — stored by
      // three lanes of the quad, one each
      uint32_t bits = 0u;
#pragma unroll
      for (int i = 0; i < NACC; ++i) bits = shift_in(bits, is_pos(pvn[i]));
      bits = __builtin_bitreverse32(bits) >> (32 - NACC);
      const int sub = lane & 3;
      if (interior && exists && sub < 3) {
        const int co0 = ph.c_off + o0;
        const unsigned widx = ((unsigned)((sub * (g.C >> 6) + (co0 >> 6)) * g.ncell_pool) + cello) * 2u + (unsigned)((co0 & 63) >> 5);
        unsigned char* dst = smem + g.lds_out + widx * 4u + (unsigned)((co0 & 31) >> 3);
        if constexpr (NACC == 16) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)bits;
        else *dst = (uint8_t)bits;
      }
    }
    if constexpr (!LAST) {
      uint32_t bits = 0u;
#pragma unroll
      for (int i = 0; i < NACC; ++i) bits = shift_in(bits, is_pos(pvi[i]));
      bits = __builtin_bitreverse32(bits) >> (32 - NACC);  // channel o0 + i in bit i
      const HbPhase& pn = g.ph[K < 2 ? K + 1 : 2];
      const int wq = o0 >> 5;
      const unsigned widx = ((unsigned)((wq / CWCN) * pn.ncell_in) + celln) * CWCN + (unsigned)(wq % CWCN);
      unsigned char* dst = smem + pn.lds_in + widx * 4u + (unsigned)((o0 & 31) >> 3);
      if (exists) {
        if constexpr (NACC == 16) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)bits;
        else *dst = (uint8_t)bits;
      }
      // the pass is complete for this pixel group: the byte stores above precede the counter update in this wave's
      // LDS instruction stream (release)
      if (lane == 0) __hip_atomic_fetch_add(&done[pg], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

template <int CWC, bool MULTI, int K, int CWCN, int MODE, int DSW = 0>
__device__ __forceinline__ void hb_phase(const uint32_t* __restrict__ Wt, const float* __restrict__ Kc,
                                         const float* __restrict__ Kp, const float* __restrict__ res,
                                         float* __restrict__ out, const HbDs& ds, const HbGeo& g, unsigned char* smem, int n0,
                                         int kk, int y0, int rows, int lane) {
  constexpr bool POOL = MODE == HB_POOL;
  const HbPhase& ph = g.ph[K];
  const HbDom d = hb_domain<POOL>(g, K, kk, y0, rows);
  uint32_t* ctl = reinterpret_cast<uint32_t*>(smem);
  uint32_t* done0 = reinterpret_cast<uint32_t*>(smem + g.lds_done);
  // completion counters: conv1's pixel groups, then conv2's
  uint32_t* done = K == 0 ? done0 : done0 + hb_domain<POOL>(g, 0, kk, y0, rows).npg;
  const uint32_t nunits = (uint32_t)(d.npg * ph.upg);
  for (;;) {
    const uint32_t u = hb_ticket(&ctl[K], lane);
    if (u >= nunits) break;
    const int pg = (int)fast_div(u, ph.m_upg, ph.s_upg), p0 = ((int)u - pg * ph.upg) * ph.ppu;
    if constexpr (K > 0) {
      const HbDom dp = hb_domain<POOL>(g, K - 1, kk, y0, rows);
      hb_wait_inputs<POOL>(g, K, d, dp, pg, kk, K == 1 ? done0 : done, lane);
    }
    hb_unit<CWC, MULTI, K, CWCN, MODE, DSW>(Wt, Kc, Kp, res, out, ds, g, smem, d, done, pg, p0, min(ph.ppu, ph.npass - p0),
                                            n0, kk, y0, rows, lane);
  }
}

template <int CWC1, bool M1, int CWC2, bool M2, int CWC3, bool M3, int MODE, int DSW = 0>
__global__ __launch_bounds__(1024, HB_MINW) void hblock_kernel(const uint64_t* __restrict__ inP, const uint32_t* __restrict__ Wt,
                                                      const float* __restrict__ Kc, const float* __restrict__ res,
                                                      float* __restrict__ out, uint64_t* __restrict__ outP,
                                                      const HbGeo g, const float* __restrict__ Kp,
                                                      uint64_t* __restrict__ outP2, uint64_t* __restrict__ outM2,
                                                      const HbDs ds) {
  constexpr bool NEXT = MODE == HB_NEXT;
  unsigned char* smem = hb_smem;
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  [[maybe_unused]] const unsigned long long t_entry = HB_NOW();
  const int bi = g.nbi > 1 ? (int)blockIdx.x / g.nbi : (int)blockIdx.x;
  const int bj = (int)blockIdx.x - bi * g.nbi;
  const int n0 = bi * g.G, kk = min(g.G, g.N - n0);
  const int y0 = bj * g.BR, rows = min(g.BR, g.H - y0);
  const int hw = g.H * g.W;
  {  // zero: tickets, completion counters, padding cells, halo rows outside the image
    uint4* z = reinterpret_cast<uint4*>(smem);
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (unsigned i = tid; i < g.lds16; i += nthr) z[i] = zero;
  }
  __syncthreads();
  [[maybe_unused]] const unsigned long long t_zero = HB_NOW();
  {  // phase 0: the block's input planes, band + 3 halo rows (rows of the image only)
    const HbPhase& p1 = g.ph[0];
    const int ra = max(0, y0 - 3), rb = min(g.H, y0 + rows + 3);
    const int per_img = (rb - ra) * g.W;
    uint32_t* pl = reinterpret_cast<uint32_t*>(smem + p1.lds_in);
    // (image, 64-channel group) pairs in the outer loops, wave-uniform; one division per element, by W, as a multiply —
    // the flat form's three divisions by run-time values were ~80 VALU instructions per element)
    for (int img = 0; img < kk; ++img)
      for (int gq = 0; gq < g.ng_in; ++gq) {
        const uint64_t* src = inP + ((size_t)(n0 + img) * g.ng_in + gq) * hw + (size_t)ra * g.W;
        const int w0 = 2 * gq;
        const unsigned cbase = (unsigned)((img * p1.rows_in + (ra - y0 + 3)) * g.WP + 1);
        for (int r2 = tid; r2 < per_img; r2 += nthr) {
          const int rowl = (int)fast_div((uint32_t)r2, g.m_wc, g.s_wc), col = r2 - rowl * g.W;
          const uint64_t v = src[r2];
          const unsigned cell = cbase + (unsigned)(rowl * g.WP + col);
          if constexpr (CWC1 == 1) {
            pl[cell] = (uint32_t)v;
          } else {
            const unsigned widx = ((unsigned)((w0 / CWC1) * p1.ncell_in) + cell) * CWC1 + (unsigned)(w0 % CWC1);
            *reinterpret_cast<uint2*>(pl + widx) = uint2{(uint32_t)v, (uint32_t)(v >> 32)};
          }
        }
      }
  }
  __syncthreads();
  // The three convolutions, each with its own ticket counter, NOT separated by barriers: a wave that finds conv1's
  // tickets handed out goes on to conv2's units, whose inputs — pixel groups of conv1's plane — are tracked by completion
  // counters (hb_wait_inputs).  A unit only waits for units of the previous convolution, all of which are in the hands
  // of running waves by then.
  [[maybe_unused]] const unsigned long long t_p0 = HB_NOW();
  hb_phase<CWC1, M1, 0, CWC2, MODE, DSW>(Wt, Kc, Kp, res, out, ds, g, smem, n0, kk, y0, rows, lane);
  [[maybe_unused]] const unsigned long long t_c1 = HB_NOW();
  hb_phase<CWC2, M2, 1, CWC3, MODE, DSW>(Wt, Kc, Kp, res, out, ds, g, smem, n0, kk, y0, rows, lane);
  [[maybe_unused]] const unsigned long long t_c2 = HB_NOW();
  hb_phase<CWC3, M3, 2, 1, MODE, DSW>(Wt, Kc, Kp, res, out, ds, g, smem, n0, kk, y0, rows, lane);
  [[maybe_unused]] const unsigned long long t_c3 = HB_NOW();
  if constexpr (NEXT) {
    __syncthreads();
    const int ngo = g.C >> 6;
    const int per_img = rows * g.W;
    const uint2* po = reinterpret_cast<const uint2*>(smem + g.lds_out);
    for (int img = 0; img < kk; ++img)
      for (int gq = 0; gq < ngo; ++gq) {
        const uint2* src = po + (unsigned)(gq * g.ncell_out + img * (g.BR * g.W));
        uint64_t* dst = outP + ((size_t)(n0 + img) * ngo + gq) * hw + (size_t)y0 * g.W;
        for (int r2 = tid; r2 < per_img; r2 += nthr) {
          const uint2 v = src[r2];
          dst[r2] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        }
      }
  }
  if constexpr (MODE == HB_POOL) {   // the three pooled plane sets of the band's windows
    __syncthreads();
    const int ngo = g.C >> 6, wpr = g.W >> 1;
    const int per_img = (rows >> 1) * wpr;
    const uint2* po = reinterpret_cast<const uint2*>(smem + g.lds_out);
    for (int st = 0; st < 3; ++st)
      for (int img = 0; img < kk; ++img)
        for (int gq = 0; gq < ngo; ++gq) {
          const uint2* src = po + (unsigned)((st * ngo + gq) * g.ncell_pool + img * ((g.BR >> 1) * wpr));
          uint64_t* dst = (st == 0 ? outP : (st == 1 ? outP2 : outM2)) + ((size_t)(n0 + img) * ngo + gq) * (hw >> 2) +
                          (size_t)(y0 >> 1) * wpr;
          for (int r2 = tid; r2 < per_img; r2 += nthr) {
            const uint2 v = src[r2];
            dst[r2] = (uint64_t)v.x | ((uint64_t)v.y << 32);
          }
        }
  }
#ifdef HB_TIMING
  if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long* d = bnn_hb_dbg + ((size_t)blockIdx.x * 16 + (tid >> 6)) * 8;
    d[0] = t_entry; d[1] = t_zero; d[2] = t_p0; d[3] = t_c1; d[4] = t_c2; d[5] = t_c3; d[6] = HB_NOW(); d[7] = 0;
  }
#endif
}

// standard packed weights (bnn_hip_pack_weight_f32: wbits[ob][chunk][j][tap][cwc], 64-channel granularity) -> the dense
// layout of one phase, [pass][chunk][j < NACC][tap][CWC]
__global__ __launch_bounds__(256) void hblock_pack_weight_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                                 int O, int cw, int cwc, int nacc, int cw_s, int cwc_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O * 9 * cw) return;
  const int cwi = i % cw, tap = (i / cw) % 9, o = i / (9 * cw);
  const int ob = o >> 5, j = o & 31, nchunk_s = cw_s / cwc_s;
  const uint32_t v = src[((size_t)(ob * nchunk_s + cwi / cwc_s) * 32 + j) * 9 * cwc_s + tap * cwc_s + cwi % cwc_s];
  const int nchunk = cw / cwc;
  dst[((size_t)((o / nacc) * nchunk + cwi / cwc) * nacc + o % nacc) * 9 * cwc + tap * cwc + cwi % cwc] = v;
}

constexpr int kHbLdsBudget = 160 * 1024;

struct HbShape {
  int cin[3], O[3], cw[3], cwc[3], nchunk[3], nacc[3];
};

bool hb_shape(int C_in, int planes, HbShape& s) {
  if (C_in <= 0 || C_in > (1 << 20) || planes < 64 || planes % 64 != 0 || planes > 4096) return false;
  s.cin[0] = C_in; s.cin[1] = planes / 2; s.cin[2] = planes / 4;
  s.O[0] = planes / 2; s.O[1] = planes / 4; s.O[2] = planes / 4;
  for (int k = 0; k < 3; ++k) {
    s.cw[k] = (s.cin[k] + 31) / 32;
    if (s.cw[k] <= 4) {
      if (s.cw[k] == 3) return false;
      s.cwc[k] = s.cw[k];
    } else {
      if (s.cw[k] % 4) return false;
      s.cwc[k] = 4;
    }
    s.nchunk[k] = s.cw[k] / s.cwc[k];
    s.nacc[k] = s.cwc[k] == 1 ? 16 : 8;
    if (s.O[k] % s.nacc[k]) return false;
  }
  // the block's input planes arrive as whole 64-channel groups; a one-word cell takes the low half of group 0
  if (s.cw[0] == 1 ? C_in > 32 : (s.cw[0] % 2 != 0)) return false;
  return true;
}

}  // namespace

int hblock_layout(int C_in, int planes, bnn_hip_hblock_layout* L) {
  HbShape s;
  if (!hb_shape(C_in, planes, s)) return BNN_HIP_ERR_UNSUPPORTED;
  long long w = 0;
  for (int k = 0; k < 3; ++k) {
    L->w_off[k] = w;
    w += (long long)s.O[k] * 9 * s.cw[k];
  }
  L->weight_words = w;
  long long f = 0;
  for (int k = 0; k < 3; ++k) { L->alpha_off[k] = f; f += s.O[k]; }
  for (int k = 0; k < 2; ++k) { L->pack_a_off[k] = f; f += s.O[k]; L->pack_b_off[k] = f; f += s.O[k]; }
  L->next_a_off = f; f += planes;
  L->next_b_off = f; f += planes;
  L->const_floats = f;
  return BNN_HIP_OK;
}

int launch_hblock_pack_weights(int C_in, int planes, const uint32_t* const w[3], uint32_t* dst, hipStream_t s) {
  HbShape sh;
  bnn_hip_hblock_layout L;
  if (!hb_shape(C_in, planes, sh) || hblock_layout(C_in, planes, &L) != BNN_HIP_OK) return BNN_HIP_ERR_UNSUPPORTED;
  for (int k = 0; k < 3; ++k) {
    const int cw_s = 2 * ((sh.cin[k] + 63) / 64), cwc_s = choose_cwc(cw_s, 3, 3);
    const int n = sh.O[k] * 9 * sh.cw[k];
    hipLaunchKernelGGL(hblock_pack_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w[k], dst + L.w_off[k],
                       sh.O[k], sh.cw[k], sh.cwc[k], sh.nacc[k], cw_s, cwc_s);
  }
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

namespace {

// LDS bytes of a region of G images x BR band rows, and the offsets of its pieces.
long long hb_lds(const HbShape& s, int W, int planes, int G, int BR, int mode, HbGeo* g) {
  long long off = 16;  // tickets
  {  // completion counters of conv1 / conv2 (64-pixel groups of their domains, at most band + halo rows; HB_POOL: + 2 rows
     // of a domain widened to whole windows)
    long long n = 0;
    for (int k = 0; k < 2; ++k) n += ((long long)G * (BR + 2 * (2 - k) + (mode == HB_POOL ? 2 : 0)) * W + 63) / 64;
    if (g) g->lds_done = (unsigned)off;
    off += (n * 4 + 15) / 16 * 16;
  }
  for (int k = 0; k < 3; ++k) {
    const int rows_in = BR + 2 * (3 - k);
    const long long ncell = (long long)G * rows_in * (W + 2);
    if (g) {
      g->ph[k].lds_in = (unsigned)off;
      g->ph[k].rows_in = rows_in;
      g->ph[k].ncell_in = (int)ncell;
    }
    off += (ncell * s.cw[k] * 4 + 15) / 16 * 16;
  }
  const long long ncell_out = (long long)G * BR * W;
  if (g) {
    g->lds_out = (unsigned)off;
    g->ncell_out = (int)ncell_out;
  }
  if (mode == HB_NEXT) off += ((long long)(planes / 64) * ncell_out * 8 + 15) / 16 * 16;
  if (mode == HB_POOL) {
    const long long ncell_pool = (long long)G * (BR / 2) * (W / 2);
    if (g) g->ncell_pool = (int)ncell_pool;
    off += (3LL * (planes / 64) * ncell_pool * 8 + 15) / 16 * 16;
  }
  return off;
}

template <class K>
int hb_launch(K kernel, const HbGeo& g, int nblocks, int waves, const uint64_t* inP, const uint32_t* W, const float* Kc,
              const float* res, float* out, uint64_t* outP, hipStream_t s, const float* Kp = nullptr,
              uint64_t* outP2 = nullptr, uint64_t* outM2 = nullptr, HbDs ds = HbDs{nullptr, nullptr, nullptr, nullptr}) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          kMaxDynamicLds) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nblocks), dim3((unsigned)waves * kWave), (size_t)g.lds16 * 16, s, inP, W, Kc,
                     res, out, outP, g, Kp, outP2, outM2, ds);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace

// The region plan: rows per band / images per region / waves.  Whole images when there are enough of them to fill the
// chip (or other work shares it: BNN_HIP_FLAG_THROUGHPUT); else bands of rows — each band recomputes 2 halo rows of
// conv1 and 1 of conv2 on each side, so the split is taken only where the idle compute units cost more.
int hblock_plan(const bnn_hip_hblock_desc* d, int* G_out, int* BR_out, int* waves_out, int mode = HB_NEXT) {
  HbShape s;
  if (!hb_shape(d->C_in, d->planes, s)) return BNN_HIP_ERR_UNSUPPORTED;
  const bool pool = mode == HB_POOL;   // bands of whole windows: an even number of rows
  if (pool && (d->H % 2 || d->W % 2)) return BNN_HIP_ERR_UNSUPPORTED;
  const int ncu = current_device_cus();
  static const bool split_when_shared = [] { const char* e = std::getenv("BNN_HBLOCK_SHARED_SPLIT"); return e && e[0] == '1'; }();
  const bool shared = (d->flags & BNN_HIP_FLAG_THROUGHPUT) != 0 && !split_when_shared;
  int G = 1, BR = d->H;
  if (d->rows_per_band > 0 || d->images_per_band > 0) {
    G = d->images_per_band > 0 ? std::min(d->images_per_band, d->N) : 1;
    BR = d->rows_per_band > 0 ? std::min(d->rows_per_band, d->H) : d->H;
    if (G > 1 && BR != d->H) return BNN_HIP_ERR_INVALID_ARG;
    if (pool && BR % 2) return BNN_HIP_ERR_INVALID_ARG;
  } else {
    const double w1 = (double)s.cin[0] * s.O[0], w2 = (double)s.cin[1] * s.O[1], w3 = (double)s.cin[2] * s.O[2];
    const int slots = shared ? std::max(1, ncu / 2) : ncu;
    double best = 0;
    int best_nbi = 0;
    for (int nbi = 1; nbi <= 4 && nbi <= d->H; ++nbi) {
      int br = (d->H + nbi - 1) / nbi;
      if (nbi > 1 && br < 4) break;
      if (pool) br += br & 1;
      const int nb = (d->H + br - 1) / br;   // (what nbi bands of br rows really are: 14 rows as 8 + 6)
      if (hb_lds(s, d->W, d->planes, 1, br, mode, nullptr) > kHbLdsBudget) continue;
      const long long rounds = ((long long)d->N * nb + slots - 1) / slots;
      // work in whole 64-pixel groups: what the lanes of a wave execute, used or not
      auto groups = [&](int r) { return (double)(((long long)std::min(d->H, r) * d->W + 63) / 64); };
      const double work = w1 * groups(br + 4) + w2 * groups(br + 2) + w3 * groups(br);
      const double cost = rounds * work;
      if (best_nbi == 0 || cost < best * 0.97) { best = cost; best_nbi = nbi; }
    }
    if (best_nbi == 0) return BNN_HIP_ERR_UNSUPPORTED;
    BR = (d->H + best_nbi - 1) / best_nbi;
    if (pool) BR += BR & 1;
    if (best_nbi == 1) {  // small images, many of them: several per region (fewer idle lanes in the last pixel group)
      while (G * 2 <= d->N && (long long)d->H * d->W * G * 2 <= 64 * 64 && d->N / (G * 2) >= 2 * ncu &&
             hb_lds(s, d->W, d->planes, G * 2, BR, mode, nullptr) <= kHbLdsBudget / 2)
        G *= 2;
    }
  }
  if (hb_lds(s, d->W, d->planes, G, BR, mode, nullptr) > kHbLdsBudget) return BNN_HIP_ERR_UNSUPPORTED;
  int waves = d->waves > 0 ? d->waves : 16;
  if (waves > 16) return BNN_HIP_ERR_INVALID_ARG;
  *G_out = G;
  *BR_out = BR;
  *waves_out = waves;
  return BNN_HIP_OK;
}

namespace {
template <int C1, bool M1, int C2, bool M2, int C3, bool M3, bool POOLED, int DSW>
int hb_dispatch(int mode, const HbGeo& g, int nblocks, int waves, const uint64_t* inP, const uint32_t* W, const float* Kc,
                const float* Kp, const float* res, float* out, uint64_t* outP, uint64_t* outP2, uint64_t* outM2,
                hipStream_t stream, const HbDs* ds) {
  if (ds != nullptr) {   // the shortcut convolution inside the launch: the widths that OPEN a stage, a next block behind them
    if constexpr (DSW > 0) {
      if (mode == HB_NEXT && g.cw_in == DSW)
        return hb_launch(hblock_kernel<C1, M1, C2, M2, C3, M3, HB_NEXT, DSW>, g, nblocks, waves, inP, W, Kc, res, out, outP,
                         stream, nullptr, nullptr, nullptr, *ds);
    }
    return BNN_HIP_ERR_UNSUPPORTED;
  }
  if constexpr (POOLED) {
    if (mode == HB_POOL)
      return hb_launch(hblock_kernel<C1, M1, C2, M2, C3, M3, HB_POOL>, g, nblocks, waves, inP, W, Kc, res, out, outP, stream,
                       Kp, outP2, outM2);
  }
  if (mode == HB_NEXT)
    return hb_launch(hblock_kernel<C1, M1, C2, M2, C3, M3, HB_NEXT>, g, nblocks, waves, inP, W, Kc, res, out, outP, stream);
  if (mode == HB_NONE)
    return hb_launch(hblock_kernel<C1, M1, C2, M2, C3, M3, HB_NONE>, g, nblocks, waves, inP, W, Kc, res, out, outP, stream);
  return BNN_HIP_ERR_UNSUPPORTED;
}

int hb_run(const bnn_hip_hblock_desc* d, int mode, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* Kp,
           const float* res, float* out, uint64_t* outP, uint64_t* outP2, uint64_t* outM2, hipStream_t stream,
           const HbDs* ds = nullptr) {
  HbShape s;
  bnn_hip_hblock_layout L;
  if (!hb_shape(d->C_in, d->planes, s) || hblock_layout(d->C_in, d->planes, &L) != BNN_HIP_OK) return BNN_HIP_ERR_UNSUPPORTED;
  int G, BR, waves;
  const int st = hblock_plan(d, &G, &BR, &waves, mode);
  if (st != BNN_HIP_OK) return st;
  HbGeo g;
  std::memset(&g, 0, sizeof(g));
  g.N = d->N; g.H = d->H; g.W = d->W; g.C = d->planes;
  g.G = G; g.BR = BR; g.nbi = (d->H + BR - 1) / BR;
  g.WP = d->W + 2;
  g.ng_in = (d->C_in + 63) / 64;
  g.cw_in = s.cw[0];
  g.lds16 = (unsigned)((hb_lds(s, d->W, d->planes, G, BR, mode, &g) + 15) / 16);
  int c_off = 0;
  for (int k = 0; k < 3; ++k) {
    HbPhase& p = g.ph[k];
    p.O = s.O[k];
    p.nchunk = s.nchunk[k];
    p.halo = 2 - k;
    p.c_off = c_off;
    c_off += s.O[k];
    p.npass = s.O[k] / s.nacc[k];
    // passes per unit: as many as keep every wave of the workgroup busy at least twice per phase
    const int rows_dom = std::min(d->H, BR + 2 * p.halo);
    const int npg = (G * rows_dom * d->W + 63) / 64;
    static const int upw = [] { const char* e = std::getenv("BNN_HBLOCK_UNITS_PER_WAVE"); return e ? std::atoi(e) : 2; }();
    int ppu = 32 / s.nacc[k];
    while (ppu > 1 && (ppu > p.npass || npg * ((p.npass + ppu - 1) / ppu) < upw * waves)) ppu >>= 1;
    p.ppu = ppu;
    p.upg = (p.npass + ppu - 1) / ppu;
    div_magic((uint32_t)p.upg, p.m_upg, p.s_upg);
    p.w_off = (unsigned)L.w_off[k];
    p.a_off = (unsigned)L.alpha_off[k];
    p.pa_off = k < 2 ? (unsigned)L.pack_a_off[k] : 0u;
    p.pb_off = k < 2 ? (unsigned)L.pack_b_off[k] : 0u;
  }
  g.na_off = (unsigned)L.next_a_off;
  g.nb_off = (unsigned)L.next_b_off;
  div_magic((uint32_t)d->W, g.m_wc, g.s_wc);
  if (mode == HB_POOL) {   // (hb_pixel: windows of an image / of a row)
    div_magic((uint32_t)(d->H * d->W / 4), g.m_hw, g.s_hw);
    div_magic((uint32_t)(d->W / 2), g.m_w, g.s_w);
  } else {
    div_magic((uint32_t)(d->H * d->W), g.m_hw, g.s_hw);
    div_magic((uint32_t)d->W, g.m_w, g.s_w);
  }
  g.f32_bytes = (unsigned)((long long)d->N * d->planes * d->H * d->W * 4);
  const int nblocks = ((d->N + G - 1) / G) * g.nbi;

#define HB_PICK(C1, M1_, C2, M2_, C3, M3_, POOLED, DSW)                                                                  \
  if (s.cwc[0] == C1 && (s.nchunk[0] > 1) == M1_ && s.cwc[1] == C2 && (s.nchunk[1] > 1) == M2_ && s.cwc[2] == C3 &&    \
      (s.nchunk[2] > 1) == M3_)                                                                                         \
    return hb_dispatch<C1, M1_, C2, M2_, C3, M3_, POOLED, DSW>(mode, g, nblocks, waves, inP, W, Kc, Kp, res, out, outP, \
                                                               outP2, outM2, stream, ds);
  // (the pooled form: the widths that END a stage of the [64, 128, 256, 512] networks)
  // (... and the form with the shortcut convolution inside: the widths that OPEN one, with the words of its input plane)
  HB_PICK(2, false, 1, false, 1, false, true, 0)    // 64 -> 64:   64 -> 32 -> 16 -> 16
  HB_PICK(2, false, 2, false, 1, false, false, 2)   // 64 -> 128:  64 -> 64 -> 32 -> 32
  HB_PICK(4, false, 2, false, 1, false, true, 0)    // 128 -> 128
  HB_PICK(4, false, 4, false, 2, false, false, 4)   // 128 -> 256
  HB_PICK(4, true, 4, false, 2, false, true, 0)     // 256 -> 256
  HB_PICK(4, true, 4, true, 4, false, false, 0)     // 256 -> 512 and 512 -> 512
#undef HB_PICK
  return BNN_HIP_ERR_UNSUPPORTED;
}
}  // namespace

int launch_hblock(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc,
                  const float* res, float* out, uint64_t* outP, hipStream_t stream) {
  return hb_run(d, outP != nullptr ? HB_NEXT : HB_NONE, inP, W, Kc, nullptr, res, out, outP, nullptr, nullptr, stream);
}

// The block + AvgPool2d(2, 2) + the two binarisations of the next stage's first block (MODE 2 at the top of the file).
int launch_hblock_pool(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc,
                       const float* Kp, const float* res, uint64_t* outP1, uint64_t* outP2, uint64_t* outM2,
                       hipStream_t stream) {
  return hb_run(d, HB_POOL, inP, W, Kc, Kp, res, nullptr, outP1, outP2, outM2, stream);
}

// The first block of a stage with its shortcut convolution inside (HbDs at the top of the file).
int launch_hblock_ds(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc,
                     const uint64_t* dsP, const uint64_t* dsM, const uint32_t* dsW, const float* dsA, float* out,
                     uint64_t* outP, hipStream_t stream) {
  const HbDs ds{dsP, dsM, dsW, dsA};
  return hb_run(d, HB_NEXT, inP, W, Kc, nullptr, nullptr, out, outP, nullptr, nullptr, stream, &ds);
}

bool hblock_ds_supported(const bnn_hip_hblock_desc* d) {
  HbShape s;
  if (!hb_shape(d->C_in, d->planes, s) || d->planes != 2 * d->C_in || !hblock_supported(d)) return false;
  const bool m0 = s.nchunk[0] > 1;
  return (s.cwc[0] == 2 && !m0 && s.cwc[1] == 2 && s.cwc[2] == 1) || (s.cwc[0] == 4 && !m0 && s.cwc[1] == 4 && s.cwc[2] == 2);
}

// standard packed weights of the 1 x 1 shortcut convolution (wbits[ob][chunk][j][cwc], 64-channel granularity) -> [o][cw]
__global__ __launch_bounds__(256) void hblock_ds_pack_weight_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                                    int O, int cw, int cwc_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O * cw) return;
  const int cwi = i % cw, o = i / cw;
  const int ob = o >> 5, j = o & 31, nchunk_s = cw / cwc_s;
  dst[i] = src[((size_t)(ob * nchunk_s + cwi / cwc_s) * 32 + j) * cwc_s + cwi % cwc_s];
}

int launch_hblock_ds_pack_weights(int C_in, int planes, const uint32_t* w, uint32_t* dst, hipStream_t s) {
  if (C_in <= 0 || C_in % 64 != 0 || C_in > 4096 || planes <= 0 || planes > 4096) return BNN_HIP_ERR_UNSUPPORTED;
  const int cw = C_in / 32, cwc_s = choose_cwc(cw, 1, 1), n = planes * cw;
  hipLaunchKernelGGL(hblock_ds_pack_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, dst, planes, cw, cwc_s);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

bool hblock_pool_supported(const bnn_hip_hblock_desc* d) {
  HbShape s;
  if (!hb_shape(d->C_in, d->planes, s) || d->C_in != d->planes) return false;
  int G, BR, waves;
  if (hblock_plan(d, &G, &BR, &waves, HB_POOL) != BNN_HIP_OK) return false;
  const bool m0 = s.nchunk[0] > 1;
  return (s.cwc[0] == 2 && !m0 && s.cwc[1] == 1 && s.cwc[2] == 1) || (s.cwc[0] == 4 && !m0 && s.cwc[1] == 2 && s.cwc[2] == 1) ||
         (s.cwc[0] == 4 && m0 && s.cwc[1] == 4 && s.nchunk[1] == 1 && s.cwc[2] == 2);
}

bool hblock_supported(const bnn_hip_hblock_desc* d) {
  HbShape s;
  if (!hb_shape(d->C_in, d->planes, s)) return false;
  int G, BR, waves;
  if (hblock_plan(d, &G, &BR, &waves) != BNN_HIP_OK) return false;
  const bool m[3] = {s.nchunk[0] > 1, s.nchunk[1] > 1, s.nchunk[2] > 1};
  const int c[3] = {s.cwc[0], s.cwc[1], s.cwc[2]};
  auto is = [&](int c1, bool m1, int c2, bool m2, int c3, bool m3) {
    return c[0] == c1 && m[0] == m1 && c[1] == c2 && m[1] == m2 && c[2] == c3 && m[2] == m3;
  };
  return is(2, false, 1, false, 1, false) || is(2, false, 2, false, 1, false) || is(4, false, 2, false, 1, false) ||
         is(4, false, 4, false, 2, false) || is(4, true, 4, false, 2, false) || is(4, true, 4, true, 4, false);
}

}  // namespace bnn

#ifdef HB_TIMING
extern "C" int bnn_hip_debug_hblock_timing(unsigned long long* host_dst, size_t n_words) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(bnn::bnn_hb_dbg), n_words * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif
