// bconv_fly.hip — the binary convolution LAYER in one launch: fp32 (or fp16) NCHW in -> fp32 NCHW out, with the
// activations binarised ON THE FLY into LDS.  No packed copy of the input ever reaches HBM.
//
// Replaces bnn/layers/conv.py:90-97 (Conv2d.forward) end to end:
//     sign(x)                         bnn/ops.py:63-66,151-152   -> pack_item(): fp32 -> two bit planes in LDS
//     conv2d(., sign(W) * alpha)      bnn/layers/conv.py:93      -> XNOR / popcount on the LDS tile (stream_weights)
//     + bias, post-process            bnn/layers/conv.py:94-96   -> epilogue<EP_PLAIN>
// Same integers and the same float operations as pack_act + bconv_sgpr_kernel (bit-identical outputs).
//
// Work decomposition
//   * workgroup = ONE BAND of the output: `kimg` whole images, or `BR` output rows of one image.  Its input window
//     (band rows + halo, zero padding materialised as zero cells) lives in LDS as sign planes:
//         cell(slab, row, col) = 2 x cw32 words;  layout [plane][chunk][cell][cwc words]  (a lane's receptive field
//         is `taps` ds_read_b128, consecutive lanes = consecutive cells: conflict-free)
//   * the waves of the workgroup are identical workers driven by two ticket counters in LDS — no barrier after the
//     initial zero fill:
//       - PACK items   (64 consecutive input pixels x one 32-channel word): lane = pixel, 32 coalesced dword loads
//         (buffer loads: per-channel offset in an SGPR), sign bits by v_alignbit_b32, non-zero bits by
//         v_cmp_class_f32 + v_addc_co_u32 (NaN / +-0 -> neither plane, denormals keep their sign), two ds_write_b32,
//         then ready[pixel group] += 1 (release)
//       - CONV units   (64 consecutive output pixels x OBW 32-channel blocks): wait until the ready counters of the
//         input pixel groups under the unit's receptive fields are complete (acquire), load the field from LDS into
//         registers, then exactly the main loop of bconv_sgpr_kernel: weights through the scalar cache into SGPRs,
//         v_bitop3_b32 + v_bcnt_u32_b32, fmaf epilogue, coalesced NCHW stores
//     A wave takes a unit ticket, first packs until the pack tickets are `ahead` pixel groups in front of its unit,
//     and — should its inputs still be missing — keeps packing while it waits: whoever waits works, so the scheme
//     cannot deadlock, and the HBM latency of one wave's packing hides under the popcount loops of the other waves
//     of its SIMD (4 per SIMD).  The input is read from HBM exactly once per band (halo rows of a row-split band
//     twice).
#include <hip/hip_fp16.h>

#include <algorithm>

#include "bconv_core.h"

namespace bnn {

struct FlyGeo {
  int C, HW;             // input channels, H * W
  int in_half;           // input element type: 0 = fp32, 1 = fp16
  unsigned x_bytes;      // range of the input descriptor
  int nwords;            // 32-channel words per pixel that hold real channels: ceil(C / 32)
  int cwc;               // words per (chunk, cell): the weight layout's chunk width
  int kimg, BR, nbi;     // band: whole images per band | output rows per band, bands per image
  int HPS, WP, ncell;    // LDS slab: rows per image, row pitch (cells), cells per band
  int prr;               // kimg > 1: real input pixels per image slab
  int nobu;              // work units per pixel group: ceil(ceil(O / 32) / OBW)
  int ahead;             // pixel groups the packing is kept in front of a unit
  unsigned off_ready, off_P, off_M, lds16;  // byte offsets into the dynamic LDS; total size in 16-byte pieces
  uint32_t m_nwords, m_W, m_prr, m_nobu;
  int s_nwords, s_W, s_prr, s_nobu;
};

// One band (wave-uniform).
struct Band {
  int n0, kk;           // first image, images
  int oy0, rows_o;      // first output row, output rows (per image)
  int npix, q0;         // output pixels of the band; global index of the first one
  int iy_base;          // input row of LDS slab row 0 (may be negative: padding)
  int iy_lo, rows_real; // real input rows [iy_lo, iy_lo + rows_real) per image
  int prr, in_pix;      // real input pixels per image / of the band
  int npg, nipg;        // 64-pixel groups of output / input pixels
};

__device__ __forceinline__ Band make_band(const Geo& g, const FlyGeo& f, int b) {
  Band B;
  const int bi = f.nbi > 1 ? b / f.nbi : b;
  const int bj = b - bi * f.nbi;
  B.n0 = bi * f.kimg;
  B.kk = min(f.kimg, g.N - B.n0);
  B.oy0 = bj * f.BR;
  B.rows_o = min(f.BR, g.Ho - B.oy0);
  B.npix = B.kk * B.rows_o * g.Wo;
  B.q0 = B.n0 * g.Ho * g.Wo + B.oy0 * g.Wo;
  B.iy_base = B.oy0 * g.sh - g.ph;
  B.iy_lo = max(0, B.iy_base);
  const int iy_end = min(g.H, B.iy_base + (B.rows_o - 1) * g.sh + (g.KH - 1) * g.dh + 1);
  B.rows_real = max(0, iy_end - B.iy_lo);
  B.prr = B.rows_real * g.Wd;
  B.in_pix = B.kk * B.prr;
  B.npg = (B.npix + 63) >> 6;
  B.nipg = (B.in_pix + 63) >> 6;
  return B;
}

// Input pixel groups [lo, hi] that hold the receptive fields of output pixel group `pg` (hull over its 64 pixels;
// hi < lo: nothing real under them).  Wave-uniform arithmetic only.
__device__ __forceinline__ void need_range(const Geo& g, const FlyGeo& f, const Band& B, int pg, int& lo, int& hi) {
  const int j0 = pg << 6, j1 = min(j0 + 64, B.npix) - 1;
  int s0 = 0, s1 = 0, r0 = j0, r1 = j1;
  if (f.kimg > 1) {  // whole images: rows_o * Wo == Ho * Wo
    s0 = (int)fast_div((uint32_t)j0, g.m_hw, g.s_hw);
    s1 = (int)fast_div((uint32_t)j1, g.m_hw, g.s_hw);
    r0 = j0 - s0 * (g.Ho * g.Wo);
    r1 = j1 - s1 * (g.Ho * g.Wo);
  }
  const int oyl0 = (int)fast_div((uint32_t)r0, g.m_wo, g.s_wo), oyl1 = (int)fast_div((uint32_t)r1, g.m_wo, g.s_wo);
  // first real row under the first pixel / last real row under the last pixel, as indices into the real rows
  const int rlo = min(max(B.iy_base + oyl0 * g.sh, B.iy_lo) - B.iy_lo, B.rows_real);
  const int rhi = min(B.iy_base + oyl1 * g.sh + (g.KH - 1) * g.dh, B.iy_lo + B.rows_real - 1) - B.iy_lo;
  const int plo = s0 * B.prr + rlo * g.Wd;
  const int phi = s1 * B.prr + (max(rhi, -1) + 1) * g.Wd - 1;
  lo = plo >> 6;
  hi = phi < 0 ? -1 : (phi >> 6);
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

// Next ticket of an LDS counter, the same value in every lane.
__device__ __forceinline__ uint32_t take_ticket(uint32_t* ctr, int lane) {
  uint32_t t = 0;
  if (lane == 0) t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return uniform(t);
}

constexpr int kClassNonzero = kClassPos | kClassNeg;  // finite non-zero or infinite: sign(x) != 0

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t buf_ld_u32s(BufRsrc r, unsigned lane_boff, unsigned chan_boff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_boff, (int)chan_boff, 0);
}
__device__ __forceinline__ uint32_t buf_ld_u16s(BufRsrc r, unsigned lane_boff, unsigned chan_boff) {
  return (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)lane_boff, (int)chan_boff, 0);
}
// One fp32 element into the two running words of its pixel:
//     s = 2 * s + (x >> 31)          sign bit                     v_alignbit_b32 on the pair {s, x}
//     z = 2 * z + (sign(x) != 0)     finite non-zero or infinite  v_cmp_class_f32 -> vcc, v_addc_co_u32 z, z, z, vcc
// Hand-written: (1) hipcc 7.2 folds the class test to `fcmp one` and then selects v_cmp_NEQ_f32 (true for NaN) when
// the result feeds a ballot — NaN would count as non-zero; (2) the v_alignbit between the compare and the add-with-
// carry is the wait state gfx950 wants between a VALU write of an SGPR pair and its VALU read.
__device__ __forceinline__ void shift_in_f32(uint32_t& s, uint32_t& z, uint32_t x, uint32_t class_mask) {
  asm("v_cmp_class_f32 vcc, %2, %3\n\t"
      "v_alignbit_b32 %1, %1, %2, 31\n\t"
      "v_addc_co_u32 %0, vcc, %0, %0, vcc"
      : "+v"(z), "+v"(s)
      : "v"(x), "s"(class_mask)
      : "vcc");
}
#else
__device__ __forceinline__ uint32_t buf_ld_u32s(BufRsrc, unsigned, unsigned) { return 0u; }
__device__ __forceinline__ uint32_t buf_ld_u16s(BufRsrc, unsigned, unsigned) { return 0u; }
__device__ __forceinline__ void shift_in_f32(uint32_t& s, uint32_t& z, uint32_t x, uint32_t) {
  s = (s << 1) | (x >> 31);
  z = (z << 1) | (((x & 0x7FFFFFFFu) != 0u && (x & 0x7FFFFFFFu) <= 0x7F800000u) ? 1u : 0u);
}
#endif

// sign() of one 32-channel word of 64 consecutive input pixels (lane = pixel): P / M bits of channels c0 .. c0+31
// (channel c0 + b in bit b; channels >= C: 0).  All loads of the word are in flight at once.
//   fp32: S = sign bits, Z = "sign(x) != 0" bits (shift_in_f32: three VALU instructions per element);
//         P = Z & ~S, M = Z & S.  -0.0 and NaN have Z = 0, denormals Z = 1: the planes of
//         pack_act_kernel bit for bit.
//   fp16: the class test of the exactly widened value (what pack_act_kernel<__half> does).
template <bool HALF>
__device__ __forceinline__ void pack_word(BufRsrc rx, unsigned voff, int c0, int C, unsigned chan_bytes,
                                          uint32_t& P, uint32_t& M) {
  // a word at the channel tail re-reads channel C-1 for its missing channels (valid memory, no branches) and
  // clears their bits afterwards
  uint32_t v[32];
#pragma unroll
  for (int b = 0; b < 32; ++b) {
    const unsigned so = (unsigned)min(c0 + b, C - 1) * chan_bytes;
    v[b] = HALF ? buf_ld_u16s(rx, voff, so) : buf_ld_u32s(rx, voff, so);
  }
  const int nch = C - c0;
  const uint32_t keep = nch >= 32 ? 0xFFFFFFFFu : ((1u << nch) - 1u);
  if constexpr (HALF) {
    uint32_t p = 0u, m = 0u;
#pragma unroll
    for (int b = 31; b >= 0; --b) {
      const float u = __half2float(__ushort_as_half((unsigned short)v[b]));
      p = shift_in(p, is_pos(u));
      m = shift_in(m, is_neg(u));
    }
    P = p & keep;
    M = m & keep;
  } else {
    uint32_t s = 0u, z = 0u;
#pragma unroll
    for (int b = 31; b >= 0; --b) shift_in_f32(s, z, v[b], (uint32_t)kClassNonzero);
    z &= keep;
    P = z & ~s;
    M = z & s;
  }
}

// Everything a wave needs to take part in the band's dataflow.
struct FlyCtx {
  uint32_t* ctl;        // [0] unit tickets, [1] pack tickets
  uint32_t* ready;      // per input pixel group: words packed so far
  uint32_t* ldsP;
  uint32_t* ldsM;
  int lane;
  int nitems;
};

// Pack one item if any is left; false when all items have been handed out.
__device__ __forceinline__ bool pack_one(const Geo& g, const FlyGeo& f, const Band& B, const FlyCtx& c,
                                         const void* __restrict__ x) {
  if (__hip_atomic_load(&c.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (uint32_t)c.nitems) return false;
  const uint32_t t = take_ticket(&c.ctl[1], c.lane);
  if (t >= (uint32_t)c.nitems) return false;
  const int ipg = (int)fast_div(t, f.m_nwords, f.s_nwords);
  const int w = (int)t - ipg * f.nwords;
  const int p = (ipg << 6) + c.lane;
  const bool valid = p < B.in_pix;
  const int pc = valid ? p : B.in_pix - 1;
  int slab = 0, r = pc;
  if (f.kimg > 1) {
    slab = (int)fast_div((uint32_t)pc, f.m_prr, f.s_prr);
    r = pc - slab * B.prr;
  }
  const int rowl = (int)fast_div((uint32_t)r, f.m_W, f.s_W);
  const int ix = r - rowl * g.Wd;
  const unsigned esz = f.in_half ? 2u : 4u;
  const unsigned elem = (unsigned)(B.n0 + slab) * (unsigned)f.C * (unsigned)f.HW + (unsigned)(B.iy_lo * g.Wd + r);
  const unsigned voff = valid ? elem * esz : 0xFFFFFFF0u;  // beyond the descriptor: the hardware returns 0
  const int cell = (slab * f.HPS + (B.iy_lo - B.iy_base) + rowl) * f.WP + ix + g.pw;
  const BufRsrc rx = make_rsrc_sized(x, f.x_bytes);
  const unsigned chan_bytes = (unsigned)f.HW * esz;
  uint32_t Pw, Mw;
  if (f.in_half) pack_word<true>(rx, voff, w * 32, f.C, chan_bytes, Pw, Mw);
  else pack_word<false>(rx, voff, w * 32, f.C, chan_bytes, Pw, Mw);
  const int wch = w / f.cwc, wi = w - wch * f.cwc;
  const unsigned a = (unsigned)(wch * f.ncell + cell) * (unsigned)f.cwc + (unsigned)wi;
  if (valid) {
    c.ldsP[a] = Pw;
    c.ldsM[a] = Mw;
  }
  // the cell writes of every lane precede the counter update in this wave's LDS instruction stream
  if (c.lane == 0) __hip_atomic_fetch_add(&c.ready[ipg], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  return true;
}

// True when every input pixel group in [lo, hi] is completely packed.
__device__ __forceinline__ bool range_ready(const FlyGeo& f, const FlyCtx& c, int lo, int hi) {
  bool missing = false;
  for (int i = lo + c.lane; i <= hi; i += 64)
    missing |= __hip_atomic_load(&c.ready[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)f.nwords;
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_amdgcn_ballot_w64(missing) != 0ull) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
#else
  return !missing;
#endif
}

// The dataflow skeleton: zero fill, then units until the tickets run out.  `conv(pg, obu)` computes one unit.
template <class ConvUnit>
__device__ __forceinline__ void fly_run(const Geo& g, const FlyGeo& f, const void* __restrict__ x,
                                        unsigned char* smem, ConvUnit&& conv) {
  const int tid = threadIdx.x, lane = tid & 63;
  {  // zero padding cells, counters, tickets: the whole allocation
    uint4* z = reinterpret_cast<uint4*>(smem);
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (unsigned i = tid; i < f.lds16; i += blockDim.x) z[i] = zero;
  }
  __syncthreads();
  const Band B = make_band(g, f, blockIdx.x);
  FlyCtx c;
  c.ctl = reinterpret_cast<uint32_t*>(smem);
  c.ready = reinterpret_cast<uint32_t*>(smem + f.off_ready);
  c.ldsP = reinterpret_cast<uint32_t*>(smem + f.off_P);
  c.ldsM = reinterpret_cast<uint32_t*>(smem + f.off_M);
  c.lane = lane;
  c.nitems = B.nipg * f.nwords;
  const int nunits = B.npg * f.nobu;
  for (;;) {
    const uint32_t u = take_ticket(&c.ctl[0], lane);
    if (u >= (uint32_t)nunits) break;
    const int pg = (int)fast_div(u, f.m_nobu, f.s_nobu);
    const int obu = (int)u - pg * f.nobu;
    // keep the packing `ahead` pixel groups in front of this unit, and wait for the unit's own inputs; a wave that
    // has to wait packs meanwhile (one call site: the packing code exists once per kernel)
    int lo, hi;
    need_range(g, f, B, min(pg + f.ahead, B.npg - 1), lo, hi);
    const uint32_t want = (uint32_t)min((hi + 1) * f.nwords, c.nitems);
    need_range(g, f, B, pg, lo, hi);
    for (unsigned idle = 0;;) {
      if (__hip_atomic_load(&c.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want &&
          range_ready(f, c, lo, hi))
        break;
      if (!pack_one(g, f, B, c, x)) {
        // every item has been handed out: the missing ones are in flight in other waves (a few microseconds).
        // Watchdog: seconds of idling can only mean a lost update — abort the launch loudly rather than hang.
        __builtin_amdgcn_s_sleep(8);
        if (++idle > (1u << 24)) __builtin_trap();
      }
    }
    conv(B, c, pg, obu, lane);
  }
}

// ---------------------------------------------------------------------------------
// Tiled unit: the main loop of bconv_sgpr_kernel on a field read from LDS.
//   MULTI: walk the chunks (C > 32 * CWC, and every 1x1 layer: PASSES = 1, 32 accumulators)
//   WZ:    zero weights (second scalar stream with the non-zero mask), 4 passes of 8 channels
//   OBW:   32-channel blocks per unit (one field load, single-chunk layers only)
// ---------------------------------------------------------------------------------
template <int KH, int KW, int CWC, bool MULTI, bool WZ, int OBW>
struct TiledUnit {
  static constexpr int T = KH * KW, NW = T * CWC;
  static constexpr int PASSES = WZ ? 4 : (MULTI ? 1 : (T > 1 ? 4 : 1));
  static constexpr int NACC = kOCB / PASSES;
  static_assert(OBW == 1 || !MULTI, "several blocks per unit: single-chunk layers only");

  template <int N>
  __device__ static __forceinline__ void lds_words(const uint32_t* base, unsigned word_off, uint32_t* dst) {
    if constexpr (N >= 4) {
#pragma unroll
      for (int i = 0; i < N / 4; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + word_off + 4 * i);
        dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
      }
    } else {
      const uint2 v = *reinterpret_cast<const uint2*>(base + word_off);
      dst[0] = v.x; dst[1] = v.y;
    }
  }

  __device__ static __forceinline__ void load_chunk(const FlyGeo& f, const FlyCtx& c, unsigned cell0, int ch,
                                                    uint32_t (&pr)[NW], uint32_t (&mr)[NW]) {
    const unsigned base = (unsigned)ch * (unsigned)f.ncell + cell0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const unsigned wo = (base + (unsigned)((t / KW) * f.WP + (t % KW))) * CWC;
      lds_words<CWC>(c.ldsP, wo, &pr[t * CWC]);
      lds_words<CWC>(c.ldsM, wo, &mr[t * CWC]);
    }
  }

  __device__ static __forceinline__ void run(const Geo& g, const FlyGeo& f, const EpiArgs& epi,
                                             const uint32_t* __restrict__ W, const uint32_t* __restrict__ Z,
                                             const Band& B, const FlyCtx& c, int pg, int obu, int lane) {
    const int jl = min((pg << 6) + lane, B.npix - 1);  // lanes past the band's last pixel copy it (same stores)
    const Pix px = decode_pixel<true>(g, B.q0 + jl);
    const unsigned cell0 = (unsigned)(((px.n - B.n0) * f.HPS + (px.oy - B.oy0) * g.sh) * f.WP + px.ox * g.sw);
    uint32_t pr[NW], mr[NW];
    int nz = 0;
    if constexpr (!MULTI) {
      load_chunk(f, c, cell0, 0, pr, mr);
      if constexpr (!WZ) nz = count_nonzero<NW>(pr, mr, 0);
    }
#pragma unroll 1
    for (int obi = 0; obi < OBW; ++obi) {
      const int ob = obu * OBW + obi;
      if (ob * kOCB >= g.O) break;
      const uint32_t* wblk = W + (size_t)ob * g.nchunk * (kOCB * NW);
      const uint32_t* zblk = WZ ? Z + (size_t)ob * g.nchunk * (kOCB * NW) : nullptr;
      const bool fullb = (ob + 1) * kOCB <= g.O;
      if constexpr (MULTI) nz = 0;
#pragma unroll 1
      for (int ps = 0; ps < PASSES; ++ps) {
        int acc[NACC];
        [[maybe_unused]] int nzacc[NACC];
        float resv[NACC];
        constexpr bool SEEDED = !WZ && NACC % 2 == 0;
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
          acc[j] = SEEDED ? (int)kCountSeed : 0;
          resv[j] = 0.0f;
          if constexpr (WZ) nzacc[j] = 0;
        }
        if constexpr (MULTI) {
          for (int ch = 0; ch < g.nchunk; ++ch) {
            load_chunk(f, c, cell0, ch, pr, mr);
            if (!WZ && ps == 0) nz = count_nonzero<NW>(pr, mr, nz);
            const size_t woff = ((size_t)ch * kOCB + ps * NACC) * NW;
            if constexpr (WZ) stream_weights_wz<NW, NACC>(wblk + woff, zblk + woff, pr, mr, acc, nzacc);
            else stream_weights<NW, NACC>(wblk + woff, pr, mr, acc);
          }
        } else {
          const size_t woff = (size_t)ps * (NACC * NW);
          if constexpr (WZ) stream_weights_wz<NW, NACC>(wblk + woff, zblk + woff, pr, mr, acc, nzacc);
          else stream_weights<NW, NACC, false, true>(wblk + woff, pr, mr, acc, SEEDED ? (int)kCountSeed : 0);
        }
        const int o0 = ob * kOCB + ps * NACC;
        uint32_t pbits = 0u, mbits = 0u;  // (no packed output here)
        if (fullb) {
          if constexpr (SEEDED) {
            epilogue<NACC, EP_PLAIN, true, true>(g, px, o0, acc, resv, epi, pbits, mbits, 0, -2.0f, (float)nz);
          } else {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = WZ ? nzacc[j] - 2 * acc[j] : nz - 2 * acc[j];
            epilogue<NACC, EP_PLAIN, true>(g, px, o0, acc, resv, epi, pbits, mbits);
          }
        } else {
#pragma unroll
          for (int j = 0; j < NACC; ++j) {
            if constexpr (SEEDED) acc[j] -= (int)kCountSeed;
            acc[j] = WZ ? nzacc[j] - 2 * acc[j] : nz - 2 * acc[j];
          }
          epilogue<NACC, EP_PLAIN>(g, px, o0, acc, resv, epi, pbits, mbits);
        }
      }
    }
  }
};

extern __shared__ __attribute__((aligned(16))) unsigned char fly_smem[];

#define BNN_FLY_PARAMS                                                                                   \
  const void *__restrict__ x, const uint32_t *__restrict__ W, const uint32_t *__restrict__ Z,            \
      const float *__restrict__ alpha, const float *__restrict__ bias, const float *__restrict__ scale,  \
      void *__restrict__ out, const Geo g, const FlyGeo f

template <int KH, int KW, int CWC, bool MULTI, bool WZ, int OBW>
__global__ __launch_bounds__(1024) void bconv_fly_kernel(BNN_FLY_PARAMS) {
  EpiArgs epi{alpha, bias, scale, nullptr, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, nullptr, nullptr};
  fly_run(g, f, x, fly_smem, [&](const Band& B, const FlyCtx& c, int pg, int obu, int lane) {
    TiledUnit<KH, KW, CWC, MULTI, WZ, OBW>::run(g, f, epi, W, Z, B, c, pg, obu, lane);
  });
}

// ---------------------------------------------------------------------------------
// Generic unit: any KH / KW / stride / padding / dilation (the arithmetic of bconv_generic_kernel on the LDS tile).
// ---------------------------------------------------------------------------------
constexpr int kFlyOG = 8;  // output channels per pass of the generic unit

template <bool WZ>
__global__ __launch_bounds__(1024) void bconv_fly_generic_kernel(BNN_FLY_PARAMS) {
  EpiArgs epi{alpha, bias, scale, nullptr, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, nullptr, nullptr};
  fly_run(g, f, x, fly_smem, [&](const Band& B, const FlyCtx& c, int pg, int ob, int lane) {
    const int jl = min((pg << 6) + lane, B.npix - 1);
    const Pix px = decode_pixel<true>(g, B.q0 + jl);
    const unsigned cell0 = (unsigned)(((px.n - B.n0) * f.HPS + (px.oy - B.oy0) * g.sh) * f.WP + px.ox * g.sw);
    const int taps = g.KH * g.KW;
    const int per_o = taps * g.cwc;
    int dotv[kOCB];
#pragma unroll
    for (int j = 0; j < kOCB; ++j) dotv[j] = 0;
#pragma unroll
    for (int pass = 0; pass < kOCB / kFlyOG; ++pass) {
      const int j0 = pass * kFlyOG;
      int acc[kFlyOG], nzw[kFlyOG];
#pragma unroll
      for (int k = 0; k < kFlyOG; ++k) { acc[k] = 0; nzw[k] = 0; }
      int nz = 0;
      if (ob * kOCB + j0 < g.O) {
        for (int t = 0; t < taps; ++t) {
          const int ky = t / g.KW, kx = t - ky * g.KW;
          const unsigned cell = cell0 + (unsigned)(ky * g.dh * f.WP + kx * g.dw);
          for (int cw = 0; cw < g.cw32; ++cw) {
            const int ch = cw / g.cwc, ci = cw - ch * g.cwc;
            const unsigned a = ((unsigned)ch * (unsigned)f.ncell + cell) * (unsigned)g.cwc + (unsigned)ci;
            const uint32_t pw = c.ldsP[a], mw = c.ldsM[a];
            if (!WZ) nz += __builtin_popcount(pw | mw);
            const size_t wbase = ((size_t)(ob * g.nchunk + ch) * kOCB + j0) * per_o + t * g.cwc + ci;
#pragma unroll
            for (int k = 0; k < kFlyOG; ++k) {
              const uint32_t w = W[wbase + (size_t)k * per_o];
              uint32_t d = disagree(w, mw, pw);
              if (WZ) {
                const uint32_t z = Z[wbase + (size_t)k * per_o];
                d &= z;
                nzw[k] += __builtin_popcount((pw | mw) & z);
              }
              acc[k] += __builtin_popcount(d);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kFlyOG; ++k) dotv[j0 + k] = (WZ ? nzw[k] : nz) - 2 * acc[k];
    }
    uint32_t pbits = 0u, mbits = 0u;
    float resv[kOCB];
#pragma unroll
    for (int j = 0; j < kOCB; ++j) resv[j] = 0.0f;
    epilogue<kOCB, EP_PLAIN>(g, px, ob * kOCB, dotv, resv, epi, pbits, mbits);
  });
}

// ---------------------------------------------------------------------------------
// host side: band plan + dispatch
// ---------------------------------------------------------------------------------
namespace {

constexpr int kLdsBudget = 160 * 1024;   // per CU (MI355X_MICROARCH.md); one workgroup may take all of it
constexpr int kLdsHalf = 78 * 1024;      // two workgroups per CU

int slab_rows(const ConvP& p, int BR) { return (BR - 1) * p.sh + (p.KH - 1) * p.dh + 1; }

// real input pixels of the largest band -> entries of the ready array
long long band_in_pixels(const ConvP& p, int kimg, int BR) {
  const int rows = std::min(p.H, slab_rows(p, BR));
  return (long long)kimg * rows * p.Wd;
}

long long lds_bytes_for(const ConvP& p, int kimg, int BR, unsigned* off_ready, unsigned* off_P, unsigned* off_M) {
  const long long ncell = (long long)kimg * slab_rows(p, BR) * (p.Wd + 2 * p.pw);
  const long long nipg = (band_in_pixels(p, kimg, BR) + 63) / 64 + 1;
  const long long ready = 16;
  const long long P = (ready + 4 * nipg + 15) / 16 * 16;
  const long long plane = ncell * p.cw32 * 4;
  if (off_ready) *off_ready = (unsigned)ready;
  if (off_P) *off_P = (unsigned)P;
  if (off_M) *off_M = (unsigned)(P + plane);
  return P + 2 * plane;
}

}  // namespace

bool fly_supported(const ConvP& p) {
  if (!small_indices(p)) return false;
  // one output row of one image with its halo must fit
  return lds_bytes_for(p, 1, 1, nullptr, nullptr, nullptr) <= kLdsBudget;
}

int fly_default_plan(const ConvP& p, int /*flags*/, bnn_hip_fly_plan* plan) {
  if (!fly_supported(p)) return BNN_HIP_ERR_UNSUPPORTED;
  const int nob = (p.O + kOCB - 1) / kOCB;
  const bool single3 = p.KH == 3 && p.KW == 3 && p.dh == 1 && p.dw == 1 && p.nchunk == 1;
  int obw = 1;
  if (single3 && nob >= 2) obw = 2;
  const long long img = lds_bytes_for(p, 1, p.Ho, nullptr, nullptr, nullptr);
  int kimg = 1, BR = p.Ho;
  if (img <= kLdsBudget) {
    // whole images: as many as keep two workgroups per CU resident, without starving the chip of bands
    // (>= 2 per CU wanted) and without bands of more than ~128 pixel groups
    const long long pix = (long long)p.Ho * p.Wo;
    while (true) {
      const int k2 = kimg * 2;
      if (k2 > p.N) break;
      if (lds_bytes_for(p, k2, p.Ho, nullptr, nullptr, nullptr) > kLdsHalf) break;
      if ((p.N + k2 - 1) / k2 < 512) break;
      if (pix * k2 > 128 * 64) break;
      kimg = k2;
    }
  } else {
    // rows of one image: the largest band that still leaves two workgroups per CU; else the largest that fits
    BR = 1;
    for (int r = p.Ho; r >= 1; --r)
      if (lds_bytes_for(p, 1, r, nullptr, nullptr, nullptr) <= kLdsHalf) { BR = r; break; }
    if (BR == 1 && lds_bytes_for(p, 1, 1, nullptr, nullptr, nullptr) > kLdsHalf) {
      for (int r = p.Ho; r >= 1; --r)
        if (lds_bytes_for(p, 1, r, nullptr, nullptr, nullptr) <= kLdsBudget) { BR = r; break; }
    }
    // equal bands
    const int nbi = (p.Ho + BR - 1) / BR;
    BR = (p.Ho + nbi - 1) / nbi;
  }
  const long long lds = lds_bytes_for(p, kimg, BR, nullptr, nullptr, nullptr);
  const long long units = (((long long)kimg * BR * p.Wo + 63) / 64) * ((nob + obw - 1) / obw);
  int waves = lds > kLdsHalf ? 16 : 8;
  while (waves > 1 && waves > units) waves >>= 1;
  plan->images_per_band = kimg;
  plan->rows_per_band = BR;
  plan->waves = waves;
  plan->blocks_per_unit = obw;
  plan->lds_bytes = (int32_t)lds;
  plan->n_bands = ((p.N + kimg - 1) / kimg) * ((p.Ho + BR - 1) / BR);
  return BNN_HIP_OK;
}

namespace {

template <class K>
int launch_k(K kernel, const ConvP& p, const void* x, const Geo& g, const FlyGeo& f, int nbands, int waves,
             hipStream_t s) {
  const size_t lds = (size_t)f.lds16 * 16;
  // per device and per kernel: set on every launch (cheap, and correct in a process that drives several GPUs)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nbands), dim3((unsigned)waves * kWave), lds, s, x, p.W, p.Z, p.alpha,
                     p.bias, p.scale, p.out, g, f);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

template <int KH, int KW, int CWC, bool MULTI>
int launch_tiled(const ConvP& p, const void* x, const Geo& g, const FlyGeo& f, int nbands, int waves, int obw,
                 bool wz, hipStream_t s) {
  if (wz) return launch_k(bconv_fly_kernel<KH, KW, CWC, MULTI, true, 1>, p, x, g, f, nbands, waves, s);
  if constexpr (!MULTI) {
    if (obw == 4) return launch_k(bconv_fly_kernel<KH, KW, CWC, false, false, 4>, p, x, g, f, nbands, waves, s);
    if (obw == 2) return launch_k(bconv_fly_kernel<KH, KW, CWC, false, false, 2>, p, x, g, f, nbands, waves, s);
  }
  return launch_k(bconv_fly_kernel<KH, KW, CWC, MULTI, false, 1>, p, x, g, f, nbands, waves, s);
}

}  // namespace

// `plan` == nullptr: the default plan.  A caller's plan is validated (LDS budget, ranges) and its derived fields
// are recomputed; only images_per_band / rows_per_band / waves / blocks_per_unit are taken from it.
int launch_bconv_fly(const ConvP& p, const void* x, int x_half, int flags, const bnn_hip_fly_plan* user,
                     hipStream_t s) {
  bnn_hip_fly_plan plan;
  int st = fly_default_plan(p, flags, &plan);
  if (st != BNN_HIP_OK) return st;
  const bool wz = (flags & BNN_HIP_FLAG_WEIGHT_ZEROS) != 0;
  const bool generic = (flags & BNN_HIP_FLAG_FORCE_GENERIC) || p.dh != 1 || p.dw != 1 ||
                       !((p.KH == 3 && p.KW == 3) || (p.KH == 1 && p.KW == 1));
  const bool single3 = !generic && p.KH == 3 && p.nchunk == 1;
  if (user) {
    plan.images_per_band = user->images_per_band;
    plan.rows_per_band = user->rows_per_band;
    plan.waves = user->waves;
    plan.blocks_per_unit = user->blocks_per_unit;
    if (plan.images_per_band < 1 || plan.rows_per_band < 1 || plan.rows_per_band > p.Ho) return BNN_HIP_ERR_INVALID_ARG;
    if (plan.images_per_band > 1 && plan.rows_per_band != p.Ho) return BNN_HIP_ERR_INVALID_ARG;
    if (plan.waves < 1 || plan.waves > 16) return BNN_HIP_ERR_INVALID_ARG;
    if (plan.blocks_per_unit != 1 && plan.blocks_per_unit != 2 && plan.blocks_per_unit != 4) return BNN_HIP_ERR_INVALID_ARG;
    plan.images_per_band = std::min(plan.images_per_band, p.N);
  }
  if (generic || wz || !single3) plan.blocks_per_unit = 1;
  const int kimg = plan.images_per_band, BR = plan.rows_per_band;
  unsigned off_ready, off_P, off_M;
  const long long lds = lds_bytes_for(p, kimg, BR, &off_ready, &off_P, &off_M);
  if (lds > kLdsBudget) return user ? BNN_HIP_ERR_INVALID_ARG : BNN_HIP_ERR_UNSUPPORTED;

  const Geo g = make_geo(p);
  FlyGeo f;
  const int C = p.C;
  f.C = C;
  f.HW = p.H * p.Wd;
  f.in_half = x_half ? 1 : 0;
  f.x_bytes = (unsigned)((long long)p.N * C * p.H * p.Wd * (x_half ? 2 : 4));
  f.nwords = (C + 31) / 32;
  f.cwc = p.cwc;
  f.kimg = kimg;
  f.BR = BR;
  f.nbi = (p.Ho + BR - 1) / BR;
  f.HPS = slab_rows(p, BR);
  f.WP = p.Wd + 2 * p.pw;
  f.ncell = kimg * f.HPS * f.WP;
  {  // whole-image bands: real rows per image
    const int iy_end = std::min(p.H, -p.ph + (p.Ho - 1) * p.sh + (p.KH - 1) * p.dh + 1);
    f.prr = std::max(0, iy_end) * p.Wd;
  }
  const int nob = (p.O + kOCB - 1) / kOCB;
  f.nobu = (nob + plan.blocks_per_unit - 1) / plan.blocks_per_unit;
  f.ahead = (plan.waves + f.nobu - 1) / f.nobu + 1;
  f.off_ready = off_ready;
  f.off_P = off_P;
  f.off_M = off_M;
  f.lds16 = (unsigned)((lds + 15) / 16);
  div_magic((uint32_t)f.nwords, f.m_nwords, f.s_nwords);
  div_magic((uint32_t)p.Wd, f.m_W, f.s_W);
  div_magic((uint32_t)std::max(1, f.prr), f.m_prr, f.s_prr);
  div_magic((uint32_t)f.nobu, f.m_nobu, f.s_nobu);
  const int nbands = ((p.N + kimg - 1) / kimg) * f.nbi;
  const int waves = plan.waves;

  if (generic) {
    if (wz) return launch_k(bconv_fly_generic_kernel<true>, p, x, g, f, nbands, waves, s);
    return launch_k(bconv_fly_generic_kernel<false>, p, x, g, f, nbands, waves, s);
  }
  const int obw = plan.blocks_per_unit;
#define BNN_FLY_PICK(KH_, KW_, C_, M_) \
  if (p.KH == KH_ && p.KW == KW_ && p.cwc == C_ && (p.nchunk > 1 || KH_ == 1) == M_) \
    return launch_tiled<KH_, KW_, C_, M_>(p, x, g, f, nbands, waves, obw, wz, s);
  BNN_FLY_PICK(3, 3, 4, false) BNN_FLY_PICK(3, 3, 4, true) BNN_FLY_PICK(3, 3, 2, false) BNN_FLY_PICK(3, 3, 2, true)
  BNN_FLY_PICK(1, 1, 16, true) BNN_FLY_PICK(1, 1, 8, true) BNN_FLY_PICK(1, 1, 4, true) BNN_FLY_PICK(1, 1, 2, true)
#undef BNN_FLY_PICK
  if (wz) return launch_k(bconv_fly_generic_kernel<true>, p, x, g, f, nbands, waves, s);
  return launch_k(bconv_fly_generic_kernel<false>, p, x, g, f, nbands, waves, s);
}

}  // namespace bnn
