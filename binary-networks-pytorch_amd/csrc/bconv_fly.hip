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
//         is `taps` ds_read_b128, consecutive lanes = consecutive cells: conflict-free), plus one counter per cell:
//         its non-zero channels (the window's count is `taps` loads and adds instead of 2 VALU per field word)
//   * two ticket counters in LDS drive the waves — no barrier after the initial zero fill:
//       - PACK items   (64 consecutive input pixels x up to four 32-channel words): lane = pixel, 32 coalesced dword loads
//         (buffer loads: per-channel offset in an SGPR, non-temporal), sign bits by v_alignbit_b32, non-zero bits by
//         v_cmp_class_f32 + v_addc_co_u32 (NaN / +-0 -> neither plane, denormals keep their sign), two ds_write_b32,
//         then ready[pixel group] += 1 (release)
//       - CONV units   (64 consecutive output pixels x a run of 8-channel passes): wait until the ready counters of
//         the input pixel groups under the unit's receptive fields are complete (acquire), load the field from LDS
//         into registers, then exactly the main loop of bconv_sgpr_kernel: weights through the scalar cache into
//         SGPRs, v_bitop3_b32 + v_bcnt_u32_b32, fmaf epilogue, coalesced NCHW stores
//   * PRODUCER waves (the first `nprod` of the workgroup) take pack tickets front to back as fast as HBM delivers, then
//     join the others; the other waves take unit tickets and only pack while they have to WAIT for their inputs
//     (start-up): whoever waits works, so the scheme cannot deadlock.  Producers never store and consumers (almost)
//     never load: gfx950 counts loads and stores in ONE counter (vmcnt), so a wave that mixes them waits for its
//     stores' acknowledgements whenever it needs a loaded value (measured: packing between the popcount passes of a
//     wave — "asynchronous", loads landing under the next pass — was 15 % SLOWER than dedicated producers).
//   * wave-uniform state is NOT kept across a unit: the popcount loop owns 64 of the ~100 SGPRs for the weight
//     stream, and everything else alive across it was spilled to VGPR lanes and read back with v_readlane — ~200
//     VALU instructions per unit.  Each iteration re-reads what it needs from the kernel-argument segment instead
//     (s_load, no VALU): fresh_geo().
//   The input is read from HBM exactly once per band (halo rows of a row-split band twice).
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstddef>

#include "bconv_core.h"

namespace bnn {

struct FlyGeo {
  int C, HW;             // input channels, H * W
  int in_half;           // input element type: 0 = fp32, 1 = fp16
  unsigned x_bytes;      // range of the input descriptor
  int nwords;            // 32-channel words per pixel that hold real channels: ceil(C / 32)
  int wpi, gpi;          // PACK items: words per item (min(4, cwc)), items per input pixel group
  int cwc;               // words per (chunk, cell): the weight layout's chunk width
  int kimg, BR, nbi;     // band: whole images per band | output rows per band, bands per image
  int HPS, WP, ncell;    // LDS slab: rows per image, row pitch (cells), cells per band
  int prr;               // kimg > 1: real input pixels per image slab
  // work units, in PASSES of the kernel (a pass = 32 / ppb output channels of one 32-channel block):
  int ppb;               // passes per 32-channel block (the kernel's PASSES)
  int npp;               // passes per pixel group: ceil(O / 32) * ppb
  int cpu;               // passes per COARSE unit (blocks_per_unit * ppb): one load of the field
  int nobu;              // coarse units per pixel group: ceil(npp / cpu)
  int fine_head, fine_tail;  // pixel groups at the start / end of a band whose units are single passes: the first
                             // rows get all waves at once (they are all that is packed yet), the last ones even out
                             // the waves' finishing times
  int ahead;             // no producer waves: pixel groups the packing is kept in front of a unit
  int nprod;             // producer waves per workgroup
  unsigned off_ready, off_P, off_M, off_cnt, lds16;  // byte offsets into the dynamic LDS; total size / 16
  uint32_t m_gpi, m_W, m_prr, m_nobu, m_cwc, m_npp;
  int s_gpi, s_W, s_prr, s_nobu, s_cwc, s_npp;
};

// The kernels' arguments: seven pointers (kept as __restrict__ kernel parameters: the compiler must know that the
// weights are not written by the launch to stream them through the scalar cache), then the two geometry blocks —
// which the kernels never touch through their formal parameters: fresh_geo() re-reads them.
struct FlyPtrs {
  const void* x;        // activations: fp32 / fp16 NCHW
  const uint32_t* W;    // packed weights (bnn_hip_pack_weight_f32)
  const uint32_t* Z;    // their non-zero mask (BNN_HIP_FLAG_WEIGHT_ZEROS)
  const float* alpha;
  const float* bias;
  const float* scale;
  void* out;
};
struct FlyArgs {   // = the layout of the kernel-argument segment
  FlyPtrs ptr;
  Geo g;
  FlyGeo f;
};
struct FlyGeos {
  Geo g;
  FlyGeo f;
};
#define BNN_FLY_PARAMS                                                                                   \
  const void *__restrict__ x, const uint32_t *__restrict__ W, const uint32_t *__restrict__ Z,            \
      const float *__restrict__ alpha, const float *__restrict__ bias, const float *__restrict__ scale,  \
      void *__restrict__ out, const Geo, const FlyGeo
#define BNN_FLY_PTRS FlyPtrs{x, W, Z, alpha, bias, scale, out}

// The geometry, re-read from the kernel-argument segment (constant address space: uniform loads from it are s_load
// instructions; only the fields a caller uses survive).  The empty asm makes the pointer a NEW value for the
// optimiser at every call, so that the loads are neither hoisted out of the unit loop nor kept alive across the
// popcount loop.
__device__ __forceinline__ FlyGeos fresh_geo() {
  FlyGeos a;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const uint32_t __attribute__((address_space(4))) * KWords;
  KWords p = (KWords)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  static_assert(sizeof(FlyGeos) % 4 == 0 && offsetof(FlyArgs, g) % 4 == 0, "whole dwords");
  static_assert(offsetof(FlyArgs, g) == sizeof(FlyPtrs) && offsetof(FlyArgs, f) == sizeof(FlyPtrs) + sizeof(Geo) &&
                    alignof(Geo) == 4 && alignof(FlyGeo) == 4 && offsetof(FlyGeos, f) == sizeof(Geo),
                "the geometry blocks follow the seven pointers without padding, as the kernel parameters do");
  uint32_t* d = reinterpret_cast<uint32_t*>(&a);
#pragma unroll
  for (unsigned i = 0; i < sizeof(FlyGeos) / 4; ++i) d[i] = p[offsetof(FlyArgs, g) / 4 + i];
#else
  a = FlyGeos{};
#endif
  return a;
}

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
constexpr uint32_t kReadyNeg = 0x10000u;  // ready[] = items packed (low half) + kReadyNeg per item with a negative value

#ifndef BNN_FLY_LOAD_AUX  // cache policy of the activation loads: 2 = nt on gfx950 (the tensor is read exactly once;
#define BNN_FLY_LOAD_AUX 2  // measured on config 2: 254 us with nt, 269 us with the default policy)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t buf_ld_u32s(BufRsrc r, unsigned lane_boff, unsigned chan_boff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_boff, (int)chan_boff, BNN_FLY_LOAD_AUX);
}
__device__ __forceinline__ uint32_t buf_ld_u16s(BufRsrc r, unsigned lane_boff, unsigned chan_boff) {
  return (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)lane_boff, (int)chan_boff,
                                                                        BNN_FLY_LOAD_AUX);
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

// What a wave needs to take part in the band's dataflow (LDS pointers; rebuilt per unit).
struct FlyCtx {
  uint32_t* ctl;        // [0] unit tickets, [1] pack tickets
  uint32_t* ready;      // per input pixel group: words packed so far
  uint32_t* ldsP;
  uint32_t* ldsM;
  uint32_t* cnt;        // per cell: channels with sign(x) != 0
  int lane;
  int nitems;
};

__device__ __forceinline__ FlyCtx make_ctx(const FlyGeo& f, const Band& B, unsigned char* smem, int lane) {
  FlyCtx c;
  c.ctl = reinterpret_cast<uint32_t*>(smem);
  c.ready = reinterpret_cast<uint32_t*>(smem + f.off_ready);
  c.ldsP = reinterpret_cast<uint32_t*>(smem + f.off_P);
  c.ldsM = reinterpret_cast<uint32_t*>(smem + f.off_M);
  c.cnt = reinterpret_cast<uint32_t*>(smem + f.off_cnt);
  c.lane = lane;
  c.nitems = B.nipg * f.gpi;
#ifdef BNN_FLY_EXP_SKIP_PACK
  c.nitems = 0;
#endif
  return c;
}

// sign() of the 32 channels c0 .. c0+31 of 64 consecutive input pixels (lane = pixel), all 32 loads in flight at once:
// P / M bits with channel c0 + b in bit b.
//   fp32: s = sign bits, z = "sign(x) != 0" bits (shift_in_f32: three VALU instructions per element), P = z & ~s,
//         M = z & s.  -0.0 and NaN have z = 0, denormals z = 1: the planes of pack_act_kernel bit for bit.
//   fp16: the class test of the exactly widened value (what pack_act_kernel<__half> does).
// A word at the channel tail re-reads channel C-1 for its missing channels (valid memory, no branches); the caller
// clears their bits.
template <bool HALF>
__device__ __forceinline__ void pack_word(BufRsrc rx, unsigned voff, int c0, int C, unsigned chan_bytes,
                                          uint32_t& P, uint32_t& M) {
  uint32_t v[32];
#pragma unroll
  for (int b = 0; b < 32; ++b) {
    const unsigned so = (unsigned)min(c0 + b, C - 1) * chan_bytes;
    v[b] = HALF ? buf_ld_u16s(rx, voff, so) : buf_ld_u32s(rx, voff, so);
  }
  uint32_t s = 0u, z = 0u;
#pragma unroll
  for (int b = 31; b >= 0; --b) {
    if constexpr (HALF) {
      const float u = __half2float(__ushort_as_half((unsigned short)v[b]));
      s = shift_in(s, is_pos(u));
      z = shift_in(z, is_neg(u));
    } else {
      shift_in_f32(s, z, v[b], (uint32_t)kClassNonzero);
    }
  }
  P = HALF ? s : (z & ~s);
  M = HALF ? z : (z & s);
}

// Pack the next item if one is left (synchronously); false when all items have been handed out.
// An item = 64 consecutive input pixels x a GROUP of `wpi` consecutive 32-channel words (ticket, pixel decode and
// addresses once per group; measured with one word per item: 250 VALU instructions per word, 96 of them the bits).
__device__ __forceinline__ bool pack_item(const void* __restrict__ x, const Geo& g, const FlyGeo& f, const Band& B,
                                          const FlyCtx& c) {
  if (__hip_atomic_load(&c.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (uint32_t)c.nitems) return false;
  const uint32_t t = take_ticket(&c.ctl[1], c.lane);
  if (t >= (uint32_t)c.nitems) return false;
  const int ipg = (int)fast_div(t, f.m_gpi, f.s_gpi);
  const int w0 = ((int)t - ipg * f.gpi) * f.wpi;  // first word of the group
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
#ifdef BNN_FLY_EXP_SAMEIMG  // experiment: every band reads image 0 (the activations come out of L2 / MALL)
  const unsigned elem = (unsigned)(B.iy_lo * g.Wd + r);
#else
  const unsigned elem = (unsigned)(B.n0 + slab) * (unsigned)f.C * (unsigned)f.HW + (unsigned)(B.iy_lo * g.Wd + r);
#endif
  const unsigned voff = valid ? elem * esz : 0xFFFFFFF0u;  // beyond the descriptor: the hardware returns 0
  const unsigned cell = (unsigned)((slab * f.HPS + (B.iy_lo - B.iy_base) + rowl) * f.WP + ix + g.pw);
  // the group lies inside one chunk (wpi divides cwc): its words are consecutive in the cell
  const int wch = (int)fast_div((uint32_t)w0, f.m_cwc, f.s_cwc);
  const unsigned a0 = ((unsigned)(wch * f.ncell) + cell) * (unsigned)f.cwc + (unsigned)(w0 - wch * f.cwc);
  const BufRsrc rx = make_rsrc_sized(x, f.x_bytes);
  uint32_t nonzero = 0u;
  [[maybe_unused]] uint32_t anyneg = 0u;
#pragma unroll 1
  for (int k = 0; k < f.wpi; ++k) {
    const int nch = f.C - (w0 + k) * 32;
    if (nch <= 0) break;  // words past the last channel stay zero
    uint32_t Pw, Mw;
    if (f.in_half) pack_word<true>(rx, voff, (w0 + k) * 32, f.C, (unsigned)f.HW * 2u, Pw, Mw);
    else pack_word<false>(rx, voff, (w0 + k) * 32, f.C, (unsigned)f.HW * 4u, Pw, Mw);
    const uint32_t keep = nch >= 32 ? 0xFFFFFFFFu : ((1u << nch) - 1u);
    Pw &= keep;
    Mw &= keep;
    if (valid) {
      c.ldsP[a0 + (unsigned)k] = Pw;
      c.ldsM[a0 + (unsigned)k] = Mw;
    }
    nonzero += (uint32_t)__builtin_popcount(Pw | Mw);
    anyneg |= Mw;
  }
  if (valid)
    __hip_atomic_fetch_add(&c.cnt[cell], nonzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // the cell writes of every lane precede the counter update in this wave's LDS instruction stream.  The counter's
  // upper half records whether the item saw a NEGATIVE value: units whose inputs have none (the output of a ReLU —
  // most binary layers of the reference's nets) run the P-plane-only loop (kReadyNeg).
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t inc = 1u + (__builtin_amdgcn_ballot_w64(valid && anyneg != 0u) != 0ull ? kReadyNeg : 0u);
#else
  const uint32_t inc = 1u;
#endif
  if (c.lane == 0) __hip_atomic_fetch_add(&c.ready[ipg], inc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  return true;
}

// True when every input pixel group in [lo, hi] is completely packed; `anyneg`: some of them hold a negative value.
__device__ __forceinline__ bool range_ready(const FlyGeo& f, const FlyCtx& c, int lo, int hi, bool& anyneg) {
  bool missing = false, neg = false;
  for (int i = lo + c.lane; i <= hi; i += 64) {
    const uint32_t v = __hip_atomic_load(&c.ready[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    missing |= (v & (kReadyNeg - 1u)) < (uint32_t)f.gpi;
    neg |= v >= kReadyNeg;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_amdgcn_ballot_w64(missing) != 0ull) return false;
  anyneg = __builtin_amdgcn_ballot_w64(neg) != 0ull;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
#else
  anyneg = neg;
  return !missing;
#endif
}

#ifdef BNN_FLY_TIMING  // variant builds only (tools/exp_fly_timing.py): per-wave cycle stamps of the dataflow
__device__ unsigned long long bnn_fly_dbg[8 * 16 * 4096];
#define FLY_NOW() __builtin_amdgcn_s_memtime()
#else
#define FLY_NOW() 0ull
#endif

#ifndef BNN_FLY_WAIT_AHEAD  // measured on config 2 (round 5, two alternating runs each): no limit 230.0 / 228.5 us,
#define BNN_FLY_WAIT_AHEAD 0  // 8: 228.6 / 230.3, 2: 227.3 / 228.1, 0: 225.3 / 224.2 — the mean wave starts its first
#endif                        // convolution 26 us after entry instead of 36 (per-wave stamps, -DBNN_FLY_TIMING)
constexpr int kWaitAhead = BNN_FLY_WAIT_AHEAD;

// The dataflow skeleton: zero fill, producers, then units until the tickets run out.
// `conv(g, f, B, c, pg, p0, np, lane, nonneg)` computes one unit: passes p0 .. p0+np-1 of pixel group pg; `nonneg`:
// no negative value under the unit's receptive fields (M planes all zero).
template <class ConvUnit>
__device__ __forceinline__ void fly_run(const void* __restrict__ x, unsigned char* smem, ConvUnit&& conv) {
  const int tid = threadIdx.x, lane = tid & 63;
  [[maybe_unused]] const unsigned long long t_entry = FLY_NOW();
  [[maybe_unused]] unsigned long long t_first = 0, t_wait = 0, t_conv = 0, t_pack = 0, n_units = 0, n_items = 0;
  {  // zero padding cells, counters, tickets: the whole allocation
    const unsigned n16 = fresh_geo().f.lds16;
    uint4* z = reinterpret_cast<uint4*>(smem);
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (unsigned i = tid; i < n16; i += blockDim.x) z[i] = zero;
  }
  __syncthreads();
#ifdef BNN_FLY_EXP_SKIP_PACK  // experiment: the convolution alone (on an all-zero tile)
  {
    const FlyGeos A = fresh_geo();
    const Band B = make_band(A.g, A.f, blockIdx.x);
    uint32_t* ready = reinterpret_cast<uint32_t*>(smem + A.f.off_ready);
    for (int i = tid; i < B.nipg; i += blockDim.x) ready[i] = (uint32_t)A.f.gpi + kReadyNeg;
    __syncthreads();
  }
#endif
  [[maybe_unused]] const unsigned long long t_zero = FLY_NOW();
  {  // PRODUCER waves: the band front to back, as fast as HBM delivers; then they join the others
    const FlyGeos A = fresh_geo();
    if ((tid >> 6) < A.f.nprod) {
      const Band B = make_band(A.g, A.f, blockIdx.x);
      const FlyCtx c = make_ctx(A.f, B, smem, lane);
      while (pack_item(x, A.g, A.f, B, c)) {
#ifdef BNN_FLY_TIMING
        ++n_items;
#endif
      }
#ifdef BNN_FLY_TIMING
      t_pack = FLY_NOW() - t_zero;
#endif
    }
  }
  for (;;) {
    const FlyGeos A = fresh_geo();
    const Geo& g = A.g;
    const FlyGeo& f = A.f;
    const Band B = make_band(g, f, blockIdx.x);
    const FlyCtx c = make_ctx(f, B, smem, lane);
    // units: single passes over the first `fine_head` and the last `fine_tail` pixel groups, coarse units between
    const int pg_head = min(f.fine_head, B.npg), pg_tail = min(f.fine_tail, B.npg - pg_head);
    const int u_head = pg_head * f.npp, u_mid = (B.npg - pg_head - pg_tail) * f.nobu;
    const int nunits = u_head + u_mid + pg_tail * f.npp;
    const uint32_t u = take_ticket(&c.ctl[0], lane);
    if (u >= (uint32_t)nunits) break;
    int pg, p0, np;  // pixel group, first pass, passes
    if (u < (uint32_t)u_head || u >= (uint32_t)(u_head + u_mid)) {
      const uint32_t v = u < (uint32_t)u_head ? u : u - (uint32_t)(u_head + u_mid);
      const int q = (int)fast_div(v, f.m_npp, f.s_npp);
      pg = (u < (uint32_t)u_head ? 0 : B.npg - pg_tail) + q;
      p0 = (int)v - q * f.npp;
      np = 1;
    } else {
      const uint32_t v = u - (uint32_t)u_head;
      const int q = (int)fast_div(v, f.m_nobu, f.s_nobu);
      pg = pg_head + q;
      p0 = ((int)v - q * f.nobu) * f.cpu;
      np = min(f.cpu, f.npp - p0);
    }
    int lo, hi;
    uint32_t want = 0u;  // without producer waves: pack tickets that should have been handed out by now
    if (f.nprod == 0) {
      need_range(g, f, B, min(pg + f.ahead, B.npg - 1), lo, hi);
      want = (uint32_t)min((hi + 1) * f.gpi, c.nitems);
    }
    need_range(g, f, B, pg, lo, hi);  // this unit's own inputs
    [[maybe_unused]] const unsigned long long t_u0 = FLY_NOW();
    bool anyneg = true;
    for (unsigned idle = 0;;) {
      if (__hip_atomic_load(&c.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want &&
          range_ready(f, c, lo, hi, anyneg))
        break;
      // whoever waits works — but (with producer waves, which hand out every item sooner or later) only on items near
      // what this unit needs: at the start of a band all 16 waves wait, and if each of them takes the next item the
      // first three rows arrive together with the first sixteen items (a third of the band: ~25 us of HBM time in
      // which no convolution runs).  BNN_FLY_WAIT_AHEAD: pixel groups beyond the unit's own inputs a waiting consumer
      // may still pack (< 0: no limit)
      const bool near = f.nprod == 0 || kWaitAhead < 0 ||
                        __hip_atomic_load(&c.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <
                            (uint32_t)((hi + 1 + kWaitAhead) * f.gpi);
      if (near && pack_item(x, g, f, B, c)) {
#ifdef BNN_FLY_TIMING
        ++n_items;
#endif
      } else {
        // every item has been handed out: the missing ones are in flight in other waves (a few microseconds).
        // Watchdog: seconds of idling can only mean a lost update — abort the launch loudly rather than hang.
        __builtin_amdgcn_s_sleep(8);
        if (++idle > (1u << 24)) __builtin_trap();
      }
    }
#ifdef BNN_FLY_TIMING
    const unsigned long long t_c0 = FLY_NOW();
    if (!t_first) t_first = t_c0;
    t_wait += t_c0 - t_u0;
#endif
    conv(g, f, B, c, pg, p0, np, lane, !anyneg);
#ifdef BNN_FLY_TIMING
    t_conv += FLY_NOW() - t_c0;
    ++n_units;
#endif
  }
#ifdef BNN_FLY_TIMING
  if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long* d = bnn_fly_dbg + ((size_t)blockIdx.x * 16 + (tid >> 6)) * 8;
    d[0] = t_entry; d[1] = t_zero; d[2] = t_first; d[3] = FLY_NOW(); d[4] = t_wait; d[5] = t_conv;
    d[6] = t_pack; d[7] = (n_units << 32) | n_items;
  }
#endif
}

// ---------------------------------------------------------------------------------
// Tiled unit: the main loop of bconv_sgpr_kernel on a field read from LDS.
//   MULTI: walk the chunks (C > 32 * CWC, and every 1x1 layer: PASSES = 1, 32 accumulators)
//   WZ:    zero weights (second scalar stream with the non-zero mask), 4 passes of 8 channels
// ---------------------------------------------------------------------------------
#ifndef BNN_FLY_ILP  // words per group of the popcount loop (stream_weights<..., ILP>)
#define BNN_FLY_ILP 2
#endif
constexpr int kIlp = BNN_FLY_ILP;
#ifndef BNN_FLY_NONNEG  // 1: units without a negative input take the P-plane-only loop (v_and + v_bcnt).  Measured on
#define BNN_FLY_NONNEG 0  // config 2 with a ReLU input: 234.4 us against 227.8 us without it — both loops in one kernel
#endif                    // (16 waves at different places of two 9 KB loop bodies) cost more than the faster pair gains,
                          // although the P-only PACKED kernel is the faster one (195 vs 201 us).  Off; tested when on.
constexpr bool kFlyNonneg = BNN_FLY_NONNEG != 0;

template <int KH, int KW, int CWC, bool MULTI, bool WZ>
struct TiledUnit {
  static constexpr int T = KH * KW, NW = T * CWC;
  static constexpr int PASSES = WZ ? 4 : (MULTI ? 1 : (T > 1 ? 4 : 1));
  static constexpr int NACC = kOCB / PASSES;

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

  template <bool NN>
  __device__ static __forceinline__ void load_chunk(const FlyCtx& c, unsigned cell0, unsigned chunk_cells, int WP,
                                                    uint32_t (&pr)[NW], uint32_t (&mr)[NW]) {
    const unsigned base = chunk_cells + cell0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const unsigned wo = (base + (unsigned)((t / KW) * WP + (t % KW))) * CWC;
      lds_words<CWC>(c.ldsP, wo, &pr[t * CWC]);
      if constexpr (!NN) lds_words<CWC>(c.ldsM, wo, &mr[t * CWC]);
    }
  }

  // One unit: passes p0 .. p0+np-1 of pixel group `pg` (pass p = channels (p % PASSES) * NACC .. of block p / PASSES).
  // NN: no negative value under the unit's windows (the M plane reads all zero): the P plane alone stays in
  // registers and the loop counts AGREEMENTS, popcount(w & p), with v_and_b32 + v_bcnt_u32_b32 — the pair issues
  // ~10 % faster than v_bitop3_b32 + v_bcnt_u32_b32 (43.5 vs 39.3 T lane-op/s); dot = 2 * agreements - non-zeros.
  template <bool NN>
  __device__ static __forceinline__ void unit(const FlyPtrs& P, const Geo& g, const FlyGeo& f, const Band& B,
                                              const FlyCtx& c, int pg, int p0, int np, int lane) {
    const EpiArgs epi{P.alpha, P.bias, P.scale, nullptr, nullptr, nullptr, nullptr, P.out,
                      nullptr, nullptr, nullptr, nullptr, nullptr};
    const int jl = min((pg << 6) + lane, B.npix - 1);  // lanes past the band's last pixel copy it (same stores)
    const Pix px = decode_pixel<true>(g, B.q0 + jl);
    const int WP = f.WP, ncell = f.ncell;
    const unsigned cell0 = (unsigned)(((px.n - B.n0) * f.HPS + (px.oy - B.oy0) * g.sh) * WP + px.ox * g.sw);
    uint32_t pr[NW], mr[NW];
    if constexpr (NN) {
#pragma unroll
      for (int i = 0; i < NW; ++i) mr[i] = 0u;
    }
    int nz = 0;  // non-zero inputs under the window: the per-cell counters of the taps
    if constexpr (!WZ) {
#pragma unroll
      for (int t = 0; t < T; ++t) nz += (int)c.cnt[cell0 + (unsigned)((t / KW) * WP + (t % KW))];
    }
    if constexpr (!MULTI) load_chunk<NN>(c, cell0, 0u, WP, pr, mr);
#pragma unroll 1
    for (int p = p0; p < p0 + np; ++p) {
      const int ob = PASSES == 1 ? p : p / PASSES, ps = PASSES == 1 ? 0 : p - ob * PASSES;
      const uint32_t* wblk = P.W + (size_t)ob * g.nchunk * (kOCB * NW);
      const uint32_t* zblk = WZ ? P.Z + (size_t)ob * g.nchunk * (kOCB * NW) : nullptr;
      const bool fullb = (ob + 1) * kOCB <= g.O;
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
          load_chunk<NN>(c, cell0, (unsigned)(ch * ncell), WP, pr, mr);
          const size_t woff = ((size_t)ch * kOCB + ps * NACC) * NW;
          if constexpr (WZ) stream_weights_wz<NW, NACC>(wblk + woff, zblk + woff, pr, mr, acc, nzacc);
          else stream_weights<NW, NACC, NN, false, false, kIlp>(wblk + woff, pr, mr, acc);
        }
      } else {
        const size_t woff = (size_t)ps * (NACC * NW);
        if constexpr (WZ) stream_weights_wz<NW, NACC>(wblk + woff, zblk + woff, pr, mr, acc, nzacc);
        else stream_weights<NW, NACC, NN, true, false, kIlp>(wblk + woff, pr, mr, acc,
                                                             SEEDED ? (int)kCountSeed : 0);
      }
      const int o0 = ob * kOCB + ps * NACC;
      uint32_t pbits = 0u, mbits = 0u;  // (no packed output here)
      auto to_dot = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = WZ ? nzacc[j] - 2 * acc[j] : NN ? 2 * acc[j] - nz : nz - 2 * acc[j];
      };
      if (fullb) {
        if constexpr (SEEDED) {
          epilogue<NACC, EP_PLAIN, true, true>(g, px, o0, acc, resv, epi, pbits, mbits, 0, NN ? 2.0f : -2.0f,
                                               NN ? -(float)nz : (float)nz);
        } else {
          to_dot();
          epilogue<NACC, EP_PLAIN, true>(g, px, o0, acc, resv, epi, pbits, mbits);
        }
      } else {
        if constexpr (SEEDED) {
#pragma unroll
          for (int j = 0; j < NACC; ++j) acc[j] -= (int)kCountSeed;
        }
        to_dot();
        epilogue<NACC, EP_PLAIN>(g, px, o0, acc, resv, epi, pbits, mbits);
      }
    }
  }

  __device__ static __forceinline__ void run(const FlyPtrs& P, const Geo& g, const FlyGeo& f, const Band& B,
                                             const FlyCtx& c, int pg, int p0, int np, int lane, bool nonneg) {
#ifdef BNN_FLY_EXP_SKIP_CONV  // experiment: the packing pipeline alone
    return;
#endif
    if constexpr (WZ || !kFlyNonneg) {
      unit<false>(P, g, f, B, c, pg, p0, np, lane);
    } else {
      if (nonneg) unit<true>(P, g, f, B, c, pg, p0, np, lane);
      else unit<false>(P, g, f, B, c, pg, p0, np, lane);
    }
  }
};

extern __shared__ __attribute__((aligned(16))) unsigned char fly_smem[];

template <int KH, int KW, int CWC, bool MULTI, bool WZ>
__global__ __launch_bounds__(1024) void bconv_fly_kernel(BNN_FLY_PARAMS) {
  const FlyPtrs P = BNN_FLY_PTRS;
  fly_run(x, fly_smem,
          [&](const Geo& g, const FlyGeo& f, const Band& B, const FlyCtx& c, int pg, int p0, int np, int lane,
              bool nonneg) { TiledUnit<KH, KW, CWC, MULTI, WZ>::run(P, g, f, B, c, pg, p0, np, lane, nonneg); });
}

// ---------------------------------------------------------------------------------
// Generic unit: any KH / KW / stride / padding / dilation (the arithmetic of bconv_generic_kernel on the LDS tile).
// One pass = one 32-channel block (ppb = 1).
// ---------------------------------------------------------------------------------
constexpr int kFlyOG = 8;  // output channels per inner pass of the generic unit

template <bool WZ>
__global__ __launch_bounds__(1024) void bconv_fly_generic_kernel(BNN_FLY_PARAMS) {
  fly_run(x, fly_smem, [&](const Geo& g, const FlyGeo& f, const Band& B, const FlyCtx& c, int pg, int p0, int np,
                           int lane, bool) {
    const EpiArgs epi{alpha, bias, scale, nullptr, nullptr, nullptr, nullptr, out,
                      nullptr, nullptr, nullptr, nullptr, nullptr};
    const int jl = min((pg << 6) + lane, B.npix - 1);
    const Pix px = decode_pixel<true>(g, B.q0 + jl);
    const unsigned cell0 = (unsigned)(((px.n - B.n0) * f.HPS + (px.oy - B.oy0) * g.sh) * f.WP + px.ox * g.sw);
    const int taps = g.KH * g.KW;
    const int per_o = taps * g.cwc;
    int nz = 0;
    if (!WZ) {
      for (int t = 0; t < taps; ++t) {
        const int ky = t / g.KW, kx = t - ky * g.KW;
        nz += (int)c.cnt[cell0 + (unsigned)(ky * g.dh * f.WP + kx * g.dw)];
      }
    }
#pragma unroll 1
    for (int ob = p0; ob < p0 + np; ++ob) {
      int dotv[kOCB];
#pragma unroll
      for (int j = 0; j < kOCB; ++j) dotv[j] = 0;
#pragma unroll
      for (int pass = 0; pass < kOCB / kFlyOG; ++pass) {
        const int j0 = pass * kFlyOG;
        int acc[kFlyOG], nzw[kFlyOG];
#pragma unroll
        for (int k = 0; k < kFlyOG; ++k) { acc[k] = 0; nzw[k] = 0; }
        if (ob * kOCB + j0 < g.O) {
          for (int t = 0; t < taps; ++t) {
            const int ky = t / g.KW, kx = t - ky * g.KW;
            const unsigned cell = cell0 + (unsigned)(ky * g.dh * f.WP + kx * g.dw);
            for (int cw = 0; cw < g.cw32; ++cw) {
              const int ch = cw / g.cwc, ci = cw - ch * g.cwc;
              const unsigned a = ((unsigned)ch * (unsigned)f.ncell + cell) * (unsigned)g.cwc + (unsigned)ci;
              const uint32_t pw = c.ldsP[a], mw = c.ldsM[a];
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
    }
  });
}

// ---------------------------------------------------------------------------------
// host side: band plan + dispatch
// ---------------------------------------------------------------------------------
namespace {

constexpr int kLdsBudget = 160 * 1024;   // per CU (MI355X_MICROARCH.md); one workgroup may take all of it
constexpr int kLdsHalf = 78 * 1024;      // two workgroups per CU
#ifndef BNN_FLY_FINE_HEAD  // default pixel groups of single-pass units at the start / end of a band
#define BNN_FLY_FINE_HEAD 1
#endif
#ifndef BNN_FLY_FINE_TAIL
#define BNN_FLY_FINE_TAIL 2
#endif
constexpr int kFineHead = BNN_FLY_FINE_HEAD, kFineTail = BNN_FLY_FINE_TAIL;

int slab_rows(const ConvP& p, int BR) { return (BR - 1) * p.sh + (p.KH - 1) * p.dh + 1; }

// real input pixels of the largest band -> entries of the ready array
long long band_in_pixels(const ConvP& p, int kimg, int BR) {
  const int rows = std::min(p.H, slab_rows(p, BR));
  return (long long)kimg * rows * p.Wd;
}

long long lds_bytes_for(const ConvP& p, int kimg, int BR, unsigned* off_ready = nullptr, unsigned* off_P = nullptr,
                        unsigned* off_M = nullptr, unsigned* off_cnt = nullptr) {
  const long long ncell = (long long)kimg * slab_rows(p, BR) * (p.Wd + 2 * p.pw);
  const long long nipg = (band_in_pixels(p, kimg, BR) + 63) / 64 + 1;
  const long long ready = 16;
  const long long P = (ready + 4 * nipg + 15) / 16 * 16;
  const long long plane = ncell * p.cw32 * 4;
  if (off_ready) *off_ready = (unsigned)ready;
  if (off_P) *off_P = (unsigned)P;
  if (off_M) *off_M = (unsigned)(P + plane);
  if (off_cnt) *off_cnt = (unsigned)(P + 2 * plane);
  return (P + 2 * plane + 4 * ncell + 15) / 16 * 16;
}

}  // namespace

bool fly_supported(const ConvP& p) {
  if (!small_indices(p)) return false;
  // one output row of one image with its halo must fit
  return lds_bytes_for(p, 1, 1) <= kLdsBudget;
}

int fly_default_plan(const ConvP& p, int /*flags*/, bnn_hip_fly_plan* plan) {
  if (!fly_supported(p)) return BNN_HIP_ERR_UNSUPPORTED;
  const int nob = (p.O + kOCB - 1) / kOCB;
  const bool single3 = p.KH == 3 && p.KW == 3 && p.dh == 1 && p.dw == 1 && p.nchunk == 1;
  int obw = 1;
  if (single3 && nob >= 2) obw = 2;
  const long long img = lds_bytes_for(p, 1, p.Ho);
  int kimg = 1, BR = p.Ho;
  if (img <= kLdsBudget) {
    // whole images: as many as keep two workgroups per CU resident, without starving the chip of bands
    // (>= 2 per CU wanted) and without bands of more than ~128 pixel groups
    const long long pix = (long long)p.Ho * p.Wo;
    while (true) {
      const int k2 = kimg * 2;
      if (k2 > p.N) break;
      if (lds_bytes_for(p, k2, p.Ho) > kLdsHalf) break;
      if ((p.N + k2 - 1) / k2 < 512) break;
      if (pix * k2 > 128 * 64) break;
      kimg = k2;
    }
  } else {
    // rows of one image: the largest band that still leaves two workgroups per CU; else the largest that fits
    BR = 1;
    for (int r = p.Ho; r >= 1; --r)
      if (lds_bytes_for(p, 1, r) <= kLdsHalf) { BR = r; break; }
    if (BR == 1 && lds_bytes_for(p, 1, 1) > kLdsHalf) {
      for (int r = p.Ho; r >= 1; --r)
        if (lds_bytes_for(p, 1, r) <= kLdsBudget) { BR = r; break; }
    }
    // equal bands
    const int nbi = (p.Ho + BR - 1) / BR;
    BR = (p.Ho + nbi - 1) / nbi;
  }
  const long long lds = lds_bytes_for(p, kimg, BR);
  const long long units = (((long long)kimg * BR * p.Wo + 63) / 64) * ((nob + obw - 1) / obw);
  // one 16-wave workgroup per CU (4 waves per SIMD at <= 128 VGPRs) beat two 8-wave workgroups on every ResNet-18
  // shape (64 ch 56x56: 87 vs 100 us; 512 ch 7x7: 77 vs 85 us), also where the band is small enough for two
  int waves = 16;
  while (waves > 1 && waves > units) waves >>= 1;
  plan->images_per_band = kimg;
  plan->rows_per_band = BR;
  plan->waves = waves;
  plan->blocks_per_unit = obw;
  plan->pack_ahead = -1;
  plan->fine_head = -1;
  plan->fine_tail = -1;
  plan->producers = -1;
  plan->lds_bytes = (int32_t)lds;
  plan->n_bands = ((p.N + kimg - 1) / kimg) * ((p.Ho + BR - 1) / BR);
  return BNN_HIP_OK;
}

namespace {

template <class K>
int launch_k(K kernel, const ConvP& p, const void* x, const Geo& g, const FlyGeo& f, int nbands, int waves,
             hipStream_t s) {
  const size_t lds = (size_t)f.lds16 * 16;
  // per device and per kernel: set on every launch (cheap, and correct in a process that drives several GPUs) — to
  // ONE constant, the CU's whole LDS, never to this launch's own size: two threads launching different layer shapes
  // must not be able to lower each other's limit between the set and the launch
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          kMaxDynamicLds) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kernel, dim3((unsigned)nbands), dim3((unsigned)waves * kWave), lds, s, x, p.W, p.Z, p.alpha,
                     p.bias, p.scale, p.out, g, f);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

template <int KH, int KW, int CWC, bool MULTI>
int launch_tiled(const ConvP& p, const void* x, const Geo& g, const FlyGeo& f, int nbands, int waves, bool wz,
                 hipStream_t s) {
  if (wz) return launch_k(bconv_fly_kernel<KH, KW, CWC, MULTI, true>, p, x, g, f, nbands, waves, s);
  return launch_k(bconv_fly_kernel<KH, KW, CWC, MULTI, false>, p, x, g, f, nbands, waves, s);
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
    plan.pack_ahead = user->pack_ahead;
    plan.fine_head = user->fine_head;
    plan.fine_tail = user->fine_tail;
    plan.producers = user->producers;
    if (plan.images_per_band < 1 || plan.rows_per_band < 1 || plan.rows_per_band > p.Ho) return BNN_HIP_ERR_INVALID_ARG;
    if (plan.images_per_band > 1 && plan.rows_per_band != p.Ho) return BNN_HIP_ERR_INVALID_ARG;
    if (plan.waves < 1 || plan.waves > 16) return BNN_HIP_ERR_INVALID_ARG;
    if (plan.blocks_per_unit != 1 && plan.blocks_per_unit != 2 && plan.blocks_per_unit != 4) return BNN_HIP_ERR_INVALID_ARG;
    plan.images_per_band = std::min(plan.images_per_band, p.N);
  }
  if (!user && !single3) plan.blocks_per_unit = 1;
  const int kimg = plan.images_per_band, BR = plan.rows_per_band;
  unsigned off_ready, off_P, off_M, off_cnt;
  const long long lds = lds_bytes_for(p, kimg, BR, &off_ready, &off_P, &off_M, &off_cnt);
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
  f.wpi = std::min(4, p.cwc);
  f.gpi = (f.nwords + f.wpi - 1) / f.wpi;
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
  const bool multi = p.nchunk > 1 || (p.KH == 1 && p.KW == 1);
  f.ppb = generic ? 1 : (wz ? 4 : (multi ? 1 : 4));  // TiledUnit::PASSES
  f.npp = nob * f.ppb;
  f.cpu = plan.blocks_per_unit * f.ppb;
  f.nobu = (f.npp + f.cpu - 1) / f.cpu;
  // single passes at both ends of a band (only where a coarse unit is more than one pass)
  f.fine_head = f.cpu > 1 ? (plan.fine_head >= 0 ? plan.fine_head : kFineHead) : 0;
  f.fine_tail = f.cpu > 1 ? (plan.fine_tail >= 0 ? plan.fine_tail : kFineTail) : 0;
  f.ahead = plan.pack_ahead >= 0 ? plan.pack_ahead : (plan.waves + f.nobu - 1) / f.nobu + 1;
  f.nprod = std::min(plan.waves - 1, plan.producers >= 0 ? plan.producers : (plan.waves + 7) / 8);
  f.off_ready = off_ready;
  f.off_P = off_P;
  f.off_M = off_M;
  f.off_cnt = off_cnt;
  f.lds16 = (unsigned)((lds + 15) / 16);
  div_magic((uint32_t)f.gpi, f.m_gpi, f.s_gpi);
  div_magic((uint32_t)p.Wd, f.m_W, f.s_W);
  div_magic((uint32_t)std::max(1, f.prr), f.m_prr, f.s_prr);
  div_magic((uint32_t)f.nobu, f.m_nobu, f.s_nobu);
  div_magic((uint32_t)f.cwc, f.m_cwc, f.s_cwc);
  div_magic((uint32_t)f.npp, f.m_npp, f.s_npp);
  const int nbands = ((p.N + kimg - 1) / kimg) * f.nbi;
  const int waves = plan.waves;

  if (generic) {
    if (wz) return launch_k(bconv_fly_generic_kernel<true>, p, x, g, f, nbands, waves, s);
    return launch_k(bconv_fly_generic_kernel<false>, p, x, g, f, nbands, waves, s);
  }
#define BNN_FLY_PICK(KH_, KW_, C_, M_) \
  if (p.KH == KH_ && p.KW == KW_ && p.cwc == C_ && multi == M_) \
    return launch_tiled<KH_, KW_, C_, M_>(p, x, g, f, nbands, waves, wz, s);
  BNN_FLY_PICK(3, 3, 4, false) BNN_FLY_PICK(3, 3, 4, true) BNN_FLY_PICK(3, 3, 2, false) BNN_FLY_PICK(3, 3, 2, true)
  BNN_FLY_PICK(1, 1, 16, true) BNN_FLY_PICK(1, 1, 8, true) BNN_FLY_PICK(1, 1, 4, true) BNN_FLY_PICK(1, 1, 2, true)
#undef BNN_FLY_PICK
  if (wz) return launch_k(bconv_fly_generic_kernel<true>, p, x, g, f, nbands, waves, s);
  return launch_k(bconv_fly_generic_kernel<false>, p, x, g, f, nbands, waves, s);
}

}  // namespace bnn

#ifdef BNN_FLY_TIMING
extern "C" int bnn_hip_debug_fly_timing(unsigned long long* host_dst, size_t n_words) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(bnn::bnn_fly_dbg), n_words * sizeof(unsigned long long)) ==
                 hipSuccess ? 0 : -3;
}
#endif
