// bconv_core.h — device-side building blocks shared by the binary-convolution kernels of bconv.hip (packed
// activations from HBM) and bconv_fly.hip (fp32 / fp16 activations binarised on the fly into LDS):
// geometry, index arithmetic, the scalar weight stream (stream_weights*), the epilogue.
// Arithmetic replaced: bnn/layers/conv.py:90-97 (see bconv.hip).
#pragma once
#include <type_traits>
#include <utility>

#include "bnn_dev.h"

namespace bnn {

// Geometry + epilogue switches.  Pointers travel as separate __restrict__ kernel arguments so
// that the compiler keeps wave-uniform reads (weights, per-channel constants) on the scalar path.
struct Geo {
  int N, H, Wd, Ho, Wo, O;
  int KH, KW, sh, sw, ph, pw, dh, dw;
  int cw32, cwc, nchunk;
  int npix;      // N*Ho*Wo
  int flags;     // EF_*
  int cw32_out;  // words per pixel per plane of the packed OUTPUT (EF_PACK)
  int c_off, c_tot;  // fp32 output / residual are [N,c_tot,Ho,Wo]; this conv owns channels c_off..c_off+O
  int tiles;     // 64-pixel tiles of the output
  int tiles_per_xcd;  // ceil(tiles / 8)
  // index arithmetic of the tiled kernels: q / (Ho*Wo) and r / Wo as multiply-high + shift (s < 0: divisor 1)
  uint32_t m_hw, m_wo, m_tpx;
  int s_hw, s_wo, s_tpx;
  unsigned in_bytes;  // bytes of one packed input plane tensor (P or M): the range of the field loads' descriptor
  int ds_cw;     // folded shortcut convolution (ShortcutArgs): 32-bit words per pixel of its input plane; 0 = none
  int ds_h, ds_w;  // > 0: the shortcut plane is given UN-POOLED at ds_h x ds_w (= the block's input) and the kernel ORs
                   // the 2 x 2 window of an output pixel itself (AvgPool2d(2, ceil) of non-negative values -> sign)
};

enum : int {
  EF_RAW = 1,      // store the int32 dot, nothing else
  EF_BIAS = 2,
  EF_SCALE = 4,    // BasicScaleBinarizer (ops.py:200-202)
  EF_BN = 8,       // folded eval-mode BatchNorm: y = fmaf(y, bn_a, bn_b)
  EF_RES = 16,     // y += residual (fp32 NCHW)
  EF_RELU = 32,
  EF_PRELU = 64,
  EF_OUTF = 128,   // write y as fp32 NCHW
  EF_PACK = 256,   // write sign(y) as bit planes for the next binary layer
  // pre-activation dataflows (BNN_HIP_EPI_*, include/bnn_hip.h)
  EF_RES_LATE = 512,    // residual added after the activation
  EF_PACK_PRE = 1024,   // binarise the value before a late residual
  EF_PACK_AFF = 2048,   // next layer's BatchNorm applied to the value that is binarised
  EF_PACK_RELU = 4096,  // planes of sign(relu(p)): M = 0
};

struct EpiArgs {
  const float* alpha;
  const float* bias;
  const float* scale;
  const float* bn_a;
  const float* bn_b;
  const float* prelu;
  const float* res;
  void* out;
  uint32_t* outP;
  uint32_t* outM;
  const float* pack_a;
  const float* pack_b;
  const int32_t* thr;  // EP_MIDT: per channel {T, flip word of its 32-channel block, A_nn, A_tp}: bit = (dot >= T) ^ flip (bnn_hip.h)
};

#ifndef BNN_TILED_MIN_WAVES  // waves per SIMD the tiled kernels are register-allocated for
#define BNN_TILED_MIN_WAVES 1
#endif

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).
// Used where every index must be a constant (register arrays, SGPR blocks) — `#pragma unroll`
// is only a hint and hipcc falls back to runtime-indexed code when it declines.
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// popcount(x) + acc in ONE instruction.  Written as asm because hipcc's reassociation otherwise
// splits long accumulation chains into v_bcnt(x, 0) + v_add3 trees (+25 % VALU in the hot loop).
__device__ __forceinline__ int popc_acc(uint32_t x, int acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  int r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
#else
  return acc + __builtin_popcount(x);
#endif
}
// Same with a wave-uniform addend (an SGPR or an inline constant: the start value of a chain costs no VGPR).
__device__ __forceinline__ int popc_acc_s(uint32_t x, int acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  int r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(acc));
  return r;
#else
  return acc + __builtin_popcount(x);
#endif
}

// WB consecutive weight words, loaded with one s_load_dwordx16.
template <int WB>
struct alignas(WB * 4) WBlock {
  uint32_t v[WB];
};

template <int LV>
__device__ __forceinline__ void load_words(const uint32_t* __restrict__ src, size_t word_off,
                                           uint32_t* dst) {
  using V = typename WordVec<LV>::type;
  const V v = *reinterpret_cast<const V*>(
      __builtin_assume_aligned(src + word_off, LV * sizeof(uint32_t)));
  const uint32_t* e = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
  for (int i = 0; i < LV; ++i) dst[i] = e[i];
}

// Global access as UNIFORM base + 32-bit per-lane BYTE offset: the form gfx950 encodes as
// `global_load/store v_data, v_off, s[base:base+1]` — no 64-bit vector address per access (two VALU ops and two
// VGPRs each when the offset is an element index the compiler cannot prove small).  The C-ABI keeps every tensor a
// launch touches below 2^32 bytes (capi.hip), so the byte offsets cannot wrap.  (hipcc takes this form for the
// field loads; for per-channel bases it re-associates to (base + lane) + channel, a 64-bit vector address again —
// pinning the base with an "s"-constrained asm turns the access into a FLAT one, which is worse.)
template <class T>
__device__ __forceinline__ T ld_off(const T* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ void st_off(T* __restrict__ base, unsigned byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// Buffer addressing for the fp32 streams of the straight-line epilogue: base = a 128-bit descriptor in SGPRs,
// per-channel byte offset = the instruction's SGPR `soffset`, per-lane byte offset = one VGPR (`offen`) —
// `buffer_load_dword v, v_off, s[desc:desc+3], s_chan offen`.  With plain pointers hipcc re-associates
// (base + lane) + channel into a 64-bit VECTOR address per access (two VALU ops and a VGPR pair each).
// The descriptor is built from kernel arguments only (wave-uniform); 0x00020000 = raw 32-bit data format.
#if defined(__HIP_DEVICE_COMPILE__)
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ BufRsrc make_rsrc_sized(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t buf_ld_u32(BufRsrc r, unsigned boff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)boff, 0, 0);
}
// wave-uniform address (scalar offset only): every lane receives the same dword
__device__ __forceinline__ uint32_t buf_ld_u32s(BufRsrc r, unsigned sboff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, 0, (int)sboff, 0);
}
__device__ __forceinline__ float buf_ld(BufRsrc r, unsigned lane_boff, unsigned chan_boff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_boff, (int)chan_boff, 0));
}
__device__ __forceinline__ void buf_st(BufRsrc r, unsigned lane_boff, unsigned chan_boff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)lane_boff, (int)chan_boff, 0);
}
#else  // host pass of hipcc only parses these
struct BufRsrc {};
__device__ __forceinline__ BufRsrc make_rsrc(const void*) { return {}; }
__device__ __forceinline__ BufRsrc make_rsrc_sized(const void*, unsigned) { return {}; }
__device__ __forceinline__ uint32_t buf_ld_u32(BufRsrc, unsigned) { return 0u; }
__device__ __forceinline__ uint32_t buf_ld_u32s(BufRsrc, unsigned) { return 0u; }
__device__ __forceinline__ float buf_ld(BufRsrc, unsigned, unsigned) { return 0.0f; }
__device__ __forceinline__ void buf_st(BufRsrc, unsigned, unsigned, float) {}
#endif

// word = 2 * word + bit: ONE v_addc_co_u32 whose carry-in is the lane mask of `bit` (hipcc builds the same value from
// v_cndmask + v_or3 + a shift: two instructions per bit).
__device__ __forceinline__ uint32_t shift_in(uint32_t word, bool bit) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(bit);
  unsigned long long carry_out;
  asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(word), "=s"(carry_out) : "s"(mask));
  return word;
#else
  return word + word + (bit ? 1u : 0u);
#endif
}

// Output pixel of this lane.
struct Pix {
  int q, n, r, oy, ox;
  bool live;
  // per-lane parts of the tensor offsets (elements / uint64 words), computed once:
  unsigned out_base;  // n*c_tot*Ho*Wo + r              fp32 output and residual, channel 0
  unsigned pk_base;   // n*(cw32_out/2)*Ho*Wo + r       packed output, group 0
  unsigned in_base;   // n*(cw32/2)*H*W                 packed input, group 0, pixel 0
};

// FAST (the tiled kernels; launch_bconv() checks the ranges): 32-bit integer multiplies and divides are the slow
// instructions of the vector ALU (v_mul_lo/hi_u32 issue at quarter rate; a division by a run-time value is ~20
// instructions, four of them such multiplies).  With every factor below 2^24 the products are v_mul_u32_u24 (full
// rate) and the two divisions one v_mul_hi_u32 + shift each (Geo::m_*, s_*): ~60 issue slots per wave less.
// (unsigned: a v_mul_u32_u24 ignores its operands' high bits, so nested products need no re-extension; the signed
// form would put a shift pair between two multiplies)
template <bool FAST>
__device__ __forceinline__ int imul(int a, int b) {
  if constexpr (FAST) {
    int r = (int)__umul24((unsigned)a, (unsigned)b);
#if defined(__HIP_DEVICE_COMPILE__)
    // opaque: a consumer that only needs the low 24 bits (another u24 multiply) would otherwise strip this one's
    // operand masks, and what is left is selected as the quarter-rate v_mul_lo_u32 again
    asm("" : "+v"(r));
#endif
    return r;
  } else {
    return a * b;
  }
}
__device__ __forceinline__ uint32_t fast_div(uint32_t x, uint32_t m, int sh) {
  return sh < 0 ? x : (__umulhi(x, m) >> sh);
}

template <bool FAST = false>
__device__ __forceinline__ Pix decode_pixel(const Geo& g, int q) {
  Pix p;
  p.live = q < g.npix;
  p.q = p.live ? q : g.npix - 1;
  const int hw = g.Ho * g.Wo;
  if constexpr (FAST) {
    p.n = (int)fast_div((uint32_t)p.q, g.m_hw, g.s_hw);
    p.r = p.q - imul<true>(p.n, hw);
    p.oy = (int)fast_div((uint32_t)p.r, g.m_wo, g.s_wo);
    p.ox = p.r - imul<true>(p.oy, g.Wo);
  } else {
    p.n = p.q / hw;
    p.r = p.q - p.n * hw;
    p.oy = p.r / g.Wo;
    p.ox = p.r - p.oy * g.Wo;
  }
  p.out_base = (unsigned)(imul<FAST>(imul<FAST>(p.n, g.c_tot), hw) + p.r);
  p.pk_base = (unsigned)(imul<FAST>(imul<FAST>(p.n, g.cw32_out >> 1), hw) + p.r);
  p.in_base = (unsigned)imul<FAST>(imul<FAST>(p.n, g.cw32 >> 1), g.H * g.Wd);
  return p;
}

// Receptive field of one chunk: T taps x CWC words x 2 planes into registers.  Taps that fall
// into the zero padding yield P = M = 0 (padding is applied after sign(): conv.py:91-92).
// NN ("non-negative"): the caller guarantees the M plane is all zero (activations out of a ReLU are
// {0,+1}); only P is loaded and `mr` stays dead, which halves the field's registers and loads.
// BUFLD: field loads through a sized buffer descriptor (single-chunk kernels; the multi-chunk ones have no SGPRs to spare
// beside the 64 of the weight stream: the descriptor pushed 100+ v_readlane/v_writelane into their chunk loop).
template <int KH, int KW, int CWC, bool NN = false, bool FAST = false, bool BUFLD = false>
__device__ __forceinline__ void load_field(const Geo& g, const Pix& px, int ch,
                                           const uint32_t* __restrict__ P,
                                           const uint32_t* __restrict__ M,
                                           uint32_t (&pr)[KH * KW * CWC],
                                           uint32_t (&mr)[KH * KW * CWC]) {
  // Planes are stored channel-group planar, [n][group of 64 channels][y][x] uint64: the 64
  // lanes of a wave (consecutive pixels) read 512 contiguous bytes per load whatever C is.
  constexpr int GC = CWC / 2;  // 64-channel groups per chunk
  const int plane = g.H * g.Wd;
  const unsigned img = px.in_base + (unsigned)(ch * GC * plane);  // uint64 words; host keeps planes < 2^29 words
  const int iy0 = imul<FAST>(px.oy, g.sh) - g.ph, ix0 = imul<FAST>(px.ox, g.sw) - g.pw;
  const int row0 = imul<FAST>(px.oy, g.sh * g.Wd) - g.ph * g.Wd;  // iy0 * W with non-negative factors only
#pragma unroll
  for (int t = 0; t < KH * KW; ++t) {
    const int iy = iy0 + t / KW;
    const int ix = ix0 + t % KW;
    const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.Wd;
    if constexpr (BUFLD) {
      // buffer loads: a tap in the zero padding gets an offset beyond the descriptor's range and the hardware returns
      // 0 for it — no select per loaded word behind the load (2 x CWC v_cndmask per tap).  (Two 4-byte loads per
      // uint64: hipcc 7.2 mis-lowers the 8-byte buffer-load builtin; the backend merges the pair where it can.)
      const BufRsrc rP = make_rsrc_sized(P, g.in_bytes), rM = make_rsrc_sized(M, g.in_bytes);
#pragma unroll
      for (int gi = 0; gi < GC; ++gi) {
        const unsigned boff = ok ? (img + (unsigned)(gi * plane + row0 + (t / KW) * g.Wd + ix)) * 8u : 0xFFFFFFF8u;
        pr[t * CWC + gi * 2] = buf_ld_u32(rP, boff);
        pr[t * CWC + gi * 2 + 1] = buf_ld_u32(rP, boff + 4u);
        mr[t * CWC + gi * 2] = NN ? 0u : buf_ld_u32(rM, boff);
        mr[t * CWC + gi * 2 + 1] = NN ? 0u : buf_ld_u32(rM, boff + 4u);
      }
      continue;
    }
    const int pix = ok ? row0 + (t / KW) * g.Wd + ix : 0;
#pragma unroll
    for (int gi = 0; gi < GC; ++gi) {
      const unsigned boff = (img + (unsigned)(gi * plane + pix)) * 8u;
      const uint2 pv = ld_off(reinterpret_cast<const uint2*>(P), boff);
      uint2 mv = {0u, 0u};
      if constexpr (!NN) mv = ld_off(reinterpret_cast<const uint2*>(M), boff);
      pr[t * CWC + gi * 2] = ok ? pv.x : 0u;
      pr[t * CWC + gi * 2 + 1] = ok ? pv.y : 0u;
      mr[t * CWC + gi * 2] = (!NN && ok) ? mv.x : 0u;
      mr[t * CWC + gi * 2 + 1] = (!NN && ok) ? mv.y : 0u;
    }
  }
}

template <int NW, bool NN = false>
__device__ __forceinline__ int count_nonzero(const uint32_t (&pr)[NW], const uint32_t (&mr)[NW],
                                             int nz) {
  int a = nz, b = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t v = NN ? pr[i] : (pr[i] | mr[i]);
    if (i & 1) b = popc_acc(v, b);
    else a = popc_acc(v, a);
  }
  return a + b;
}

// Epilogue of one wave: 64 pixels x up to 32 output channels of block `ob`.
//   y = fmaf(alpha, dot, bias) [* scale] ; [y = fmaf(y, bn_a, bn_b)] ; [y += res] ; [relu|prelu]
// then fp32 NCHW store and/or sign(y) re-packed for the next binary layer.
//
// Addressing: the per-lane part of every NCHW address (n*O*hw + r) is ONE 32-bit offset computed
// once; the per-channel part (o*hw) is wave-uniform and stays in SGPRs, so each store/load is a
// `global_* v_off, v_data, s[base]` with no per-channel vector address arithmetic.
//
// Epilogue profiles: the switches are wave-uniform run-time flags, but the two combinations a
// residual block issues on every conv are also compiled with the flags as constants, which
// removes ~1/3 of the epilogue's instructions (and all its scalar branches).
enum : int {
  EP_PLAIN = 0,    // alpha [, bias] [, post-scale] -> fp32 : the drop-in Conv2d.forward
  EP_RUNTIME = 1,  // anything, decided at run time
  EP_MID = 2,      // BN + ReLU -> packed only            (conv1 of a BasicBlock)
  EP_OUT = 3,      // BN + residual + ReLU -> fp32 + packed (conv2 of a BasicBlock)
  EP_DS = 4,       // BN -> fp32                            (1x1 conv of a shortcut branch)
  EP_LAST = 5,     // BN + residual + ReLU -> fp32 only     (conv2 of the LAST BasicBlock: the head reads fp32)
  EP_HB = 6,       // HBlock stages 1, 2 (hierarchical_block.py:38-60): raw conv + late residual -> fp32 slice of the
                   // concatenation; planes of sign(relu(bn_next(conv))) for the next stage
  EP_HB3 = 7,      // HBlock stage 3: raw conv + late residual -> fp32 slice
  EP_OUTP = 9,     // BN + residual + ReLU -> packed only  (conv2 of a BasicBlock whose fp32 output nobody reads: the
                   // next block takes its shortcut from the sign planes — AvgPool -> binary 1x1 — like its convs)
  EP_MIDT = 8,     // EP_MID with the BN + ReLU + sign folded into an integer compare of the dot (thresholds
                   // derived on the device from the same float operations: bnn_hip_sign_thresholds_f32)
};
constexpr int kFlagsMid = EF_BN | EF_RELU | EF_PACK;
constexpr int kFlagsOut = EF_BN | EF_RES | EF_RELU | EF_OUTF | EF_PACK;
constexpr int kFlagsOutP = EF_BN | EF_RES | EF_RELU | EF_PACK;
constexpr int kFlagsDs = EF_BN | EF_OUTF;
constexpr int kFlagsLast = EF_BN | EF_RES | EF_RELU | EF_OUTF;
constexpr int kFlagsHb = EF_RES | EF_RES_LATE | EF_PACK_PRE | EF_OUTF | EF_PACK | EF_PACK_AFF | EF_PACK_RELU;
constexpr int kFlagsHb3 = EF_RES | EF_RES_LATE | EF_PACK_PRE | EF_OUTF;
__device__ __forceinline__ constexpr int ep_flags(int ep, int runtime) {
  return (ep == EP_MID || ep == EP_MIDT) ? kFlagsMid
         : ep == EP_OUT ? kFlagsOut
         : ep == EP_OUTP ? kFlagsOutP
         : ep == EP_DS ? kFlagsDs
         : ep == EP_LAST ? kFlagsLast
         : ep == EP_HB ? kFlagsHb
         : ep == EP_HB3 ? kFlagsHb3
                        : runtime;
}

// FULL: all NACC channels exist (o0 + NACC <= O; wave-uniform, chosen by the caller) AND no per-lane guard: lanes past
// the last pixel were clamped to it by decode_pixel(), compute the same values and store them to the same addresses.
// The point is the instruction stream, not the handful of skipped compares: with a guard per access the compiler puts
// every residual load and every store into its own exec-mask region and, being conservative at the joins, an
// `s_waitcnt vmcnt(0)` in front of EVERY store — a wave then pays a full memory round trip per channel (measured on
// the conv2-type kernels: the two fp32 streams were purely additive to the ALU time, 74 + 29 + 29 = 132 us).
// Straight-line code lets the loads and stores of a pass queue up behind each other.
// RAWF (straight-line epilogue of the compile-time profiles): `dot` is not the dot product but the BIT PATTERN of the float
// 2^23 + count — the popcount chains of stream_weights() start from 0x4B000000, and an integer added to that pattern is
// added to the float's value.  The dot product then is fma(value - 2^23, dscale, doff) with dscale = +-2 and doff =
// -+(non-zero inputs of the lane): two packed instructions per channel PAIR, all operands small integers, so exact —
// instead of add-shift, subtract and v_cvt_f32_i32 per channel.
constexpr uint32_t kCountSeed = 0x4B000000u;  // 2^23 as fp32
template <int NACC, int EP, bool FULL = false, bool RAWF = false>
__device__ __forceinline__ void epilogue(const Geo& g, const Pix& px, int o0,
                                         const int (&dot)[NACC], const float (&resv)[NACC],
                                         const EpiArgs& e, uint32_t& pbits, uint32_t& mbits,
                                         [[maybe_unused]] int negnz = 0, [[maybe_unused]] float dscale = 0.0f,
                                         [[maybe_unused]] float doff = 0.0f) {
  static_assert(!RAWF || (FULL && NACC % 2 == 0 && EP != EP_RUNTIME && EP != EP_MIDT), "see above");
  // EP_MIDT: `dot` holds the raw popcount, `dscale` +-2 and `negnz` the lane's -+(non-zero inputs); other profiles: the dot product
  constexpr bool FUSED = EP != EP_PLAIN;
  using f2 = __attribute__((ext_vector_type(2))) float;
  [[maybe_unused]] auto dot_pair = [&](int j) -> f2 {
    if constexpr (RAWF) {
      const f2 c = f2{__int_as_float(dot[j]), __int_as_float(dot[j + 1])} - f2{8388608.0f, 8388608.0f};
      return __builtin_elementwise_fma(c, f2{dscale, dscale}, f2{doff, doff});
    } else {
      return f2{(float)dot[j], (float)dot[j + 1]};
    }
  };
  const int hw = g.Ho * g.Wo;
  const unsigned lane_off = px.out_base * 4u;  // BYTES; host keeps N*c_tot*hw < 2^30
  const int f = ep_flags(EP, g.flags);
  const bool full = FULL || o0 + NACC <= g.O;
  const bool live = FULL || px.live;
  if (!RAWF && (f & EF_RAW)) {  // (launch_sgpr() sends raw output to the run-time profile)
    if (live) {
      int32_t* o32 = static_cast<int32_t*>(e.out);
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        if (full || o0 + j < g.O) st_off(o32 + (size_t)(o0 + j + g.c_off) * hw, lane_off, dot[j]);
    }
    return;
  }
  float* outf = static_cast<float*>(e.out);
  if (!FUSED) {  // alpha, optional bias / post-scale, fp32 store: the drop-in Conv2d.forward
    const bool hb = (f & EF_BIAS) != 0, hs = (f & EF_SCALE) != 0;
    if constexpr (FULL && NACC % 2 == 0) {  // two channels per v_pk_fma_f32 (see the fused profiles below)
#pragma unroll
      for (int j = 0; j < NACC; j += 2) {
        const int o = o0 + j;
        f2 y = __builtin_elementwise_fma(f2{e.alpha[o], e.alpha[o + 1]}, dot_pair(j),
                                         hb ? f2{e.bias[o], e.bias[o + 1]} : f2{0.0f, 0.0f});
        if (hs) y *= f2{e.scale[o], e.scale[o + 1]};
#ifdef BNN_EXP_NOSTORE  // experiment (tools/fly_variants.sh): compute everything, store (almost) nothing
        if (y.x == 12345.678f)
#endif
        {
        buf_st(make_rsrc(e.out), lane_off, (unsigned)(o + g.c_off) * (unsigned)hw * 4u, y.x);
        buf_st(make_rsrc(e.out), lane_off, (unsigned)(o + 1 + g.c_off) * (unsigned)hw * 4u, y.y);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      const int o = o0 + j;
      if (full || o < g.O) {
        float y = fmaf(e.alpha[o], (float)dot[j], hb ? e.bias[o] : 0.0f);
        if (hs) y *= e.scale[o];
        if (live) st_off(outf + (size_t)(o + g.c_off) * hw, lane_off, y);
      }
    }
    return;
  }
  const int bit0 = o0 & 31;  // position of channel o0 inside its 32-channel output word
  if constexpr (EP == EP_MIDT) {
    // sign(relu(bn(alpha * dot))) == 1  <=>  (dot >= T[o]) ^ flip[o]  (csrc/thresholds.hip); M stays 0.  `dot` holds the
    // agreement (NN) / disagreement count of ONE popcount chain: dot = +-2*count + negnz is a single v_lshl_add_u32, the
    // compare takes T as its scalar operand, the flips of the block are one XOR at store time (store_packed*).
    const int km = (int)dscale;  // +2 / -2
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      const int o = o0 + j;
      if (full || o < g.O) {
        const bool bit = dot[j] * km + negnz >= e.thr[4 * o];
        if constexpr (FULL) pbits = shift_in(pbits, bit);
        else pbits |= (bit ? 1u : 0u) << (bit0 + j);
      }
    }
    return;
  }
  [[maybe_unused]] float pvs[NACC];  // FULL: the values to binarise; their bits are shifted in after the arithmetic
  if constexpr (FULL && NACC % 2 == 0) {
    // Straight-line epilogue, two channels per instruction: v_pk_fma_f32 / v_pk_add_f32 take the wave-uniform
    // constants of channels (o, o+1) as ONE aligned SGPR pair (gfx950 VALU instructions read one scalar operand, so
    // the scalar form needs a v_mov per second constant), and the ReLU is one v_max_i32 on the bit pattern
    // (negative floats are negative integers; +NaN stays NaN like torch.relu, -0.0 becomes +0.0 which compares equal).
    // Same fp32 operations in the same order per channel as the scalar loop below: bit-identical results.
    const f2 zero2 = {0.0f, 0.0f};
    const bool no_clamp = !(f & (EF_OUTF | EF_PRELU | EF_PACK_AFF | EF_RES_LATE));
#pragma unroll
    for (int j = 0; j < NACC; j += 2) {
      const int o = o0 + j;
      f2 y = __builtin_elementwise_fma(f2{e.alpha[o], e.alpha[o + 1]}, dot_pair(j),
                                       (f & EF_BIAS) ? f2{e.bias[o], e.bias[o + 1]} : zero2);
      if (f & EF_SCALE) y *= f2{e.scale[o], e.scale[o + 1]};
      // (v_pk_fma_f32 reads ONE scalar operand pair: bn_b costs two v_mov per channel pair.  Fetching it as a wave-
      // uniform buffer load instead — no vector-ALU work — was measured in round 5 and is SLOWER in the single-chunk
      // kernels, 72.5 -> 76.6 us on layer2.0.conv2: the compiler sinks the loads behind the `fullb` branch to their use,
      // and every pass then waits out a memory round trip; neutral in the multi-chunk kernels.  CHANGELOG.md.)
      if (f & EF_BN) y = __builtin_elementwise_fma(y, f2{e.bn_a[o], e.bn_a[o + 1]}, f2{e.bn_b[o], e.bn_b[o + 1]});
      if ((f & EF_RES) && !(f & EF_RES_LATE)) y += f2{resv[j], resv[j + 1]};
      if ((f & EF_RELU) && !no_clamp) {
        const float yx = y.x, yy = y.y;  // (bit_cast straight on a vector element reads element 0 with hipcc 7.2)
        y = f2{__int_as_float(max(__float_as_int(yx), 0)), __int_as_float(max(__float_as_int(yy), 0))};
      }
      if (f & EF_PRELU) {
        y.x = (y.x >= 0.0f) ? y.x : e.prelu[o] * y.x;
        y.y = (y.y >= 0.0f) ? y.y : e.prelu[o + 1] * y.y;
      }
      f2 pv = y;
      if ((f & EF_RES) && (f & EF_RES_LATE)) y += f2{resv[j], resv[j + 1]};
      if (!(f & EF_PACK_PRE)) pv = y;
      if (f & EF_OUTF) {
        buf_st(make_rsrc(e.out), lane_off, (unsigned)(o + g.c_off) * (unsigned)hw * 4u, y.x);
        buf_st(make_rsrc(e.out), lane_off, (unsigned)(o + 1 + g.c_off) * (unsigned)hw * 4u, y.y);
      }
      if (f & EF_PACK) {
        if (f & EF_PACK_AFF)
          pv = __builtin_elementwise_fma(pv, f2{e.pack_a[o], e.pack_a[o + 1]}, f2{e.pack_b[o], e.pack_b[o + 1]});
        pvs[j] = pv.x;
        pvs[j + 1] = pv.y;
      }
    }
  } else {
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int o = o0 + j;
    if (full || o < g.O) {  // wave-uniform
      float y = fmaf(e.alpha[o], (float)dot[j], (f & EF_BIAS) ? e.bias[o] : 0.0f);
      if (f & EF_SCALE) y *= e.scale[o];
      if (f & EF_BN) y = fmaf(y, e.bn_a[o], e.bn_b[o]);
      if ((f & EF_RES) && !(f & EF_RES_LATE)) y += resv[j];
      // packed-only output of a ReLU: sign(relu(y)) has P = (y > 0), M = 0 — the clamp itself is dead work
      const bool relu_dead = !(f & (EF_OUTF | EF_PRELU | EF_PACK_AFF | EF_RES_LATE));
      if ((f & EF_RELU) && !relu_dead) y = (y < 0.0f) ? 0.0f : y;  // keeps NaN, like torch.relu
      if (f & EF_PRELU) y = (y >= 0.0f) ? y : e.prelu[o] * y;
      float pv = y;  // the value the next binary layer binarises
      if ((f & EF_RES) && (f & EF_RES_LATE)) y += resv[j];
      if (!(f & EF_PACK_PRE)) pv = y;
#if BNN_NT_STORE
      if ((f & EF_OUTF) && live) __builtin_nontemporal_store(y, &(outf + (size_t)(o + g.c_off) * hw)[lane_off]);
#else
      if constexpr (FULL) {
        if (f & EF_OUTF) buf_st(make_rsrc(e.out), lane_off, (unsigned)(o + g.c_off) * (unsigned)hw * 4u, y);
      } else {
        if ((f & EF_OUTF) && live) st_off(outf + (size_t)(o + g.c_off) * hw, lane_off, y);
      }
#endif
      if (f & EF_PACK) {
        if (f & EF_PACK_AFF) pv = fmaf(pv, e.pack_a[o], e.pack_b[o]);
        // straight out of a ReLU nothing is negative: the M plane of this block stays 0
        const bool no_neg = (f & EF_PACK_RELU) ||
                            ((f & EF_RELU) && !(f & EF_PACK_AFF) && (!(f & EF_RES_LATE) || (f & EF_PACK_PRE)));
        if constexpr (FULL) {
          pvs[j] = pv;
        } else {
          pbits |= (is_pos(pv) ? 1u : 0u) << (bit0 + j);
          if (!no_neg) mbits |= (is_neg(pv) ? 1u : 0u) << (bit0 + j);
        }
      }
    }
  }
  }
  if constexpr (FULL) {
    if (f & EF_PACK) {
      // shift-in: word = 2*word + bit is ONE v_addc_co_u32 with the compare result as carry-in (instead of
      // v_cndmask + v_or3 + shift); the channels arrive in ascending order, so the finished word is bit-reversed
      // (store_packed* undo that with one v_bfrev_b32 per block).  Kept apart from the arithmetic above so that hipcc
      // still pairs the BatchNorm fmas of two channels into v_pk_fma_f32.
      const bool no_neg = (f & EF_PACK_RELU) ||
                          ((f & EF_RELU) && !(f & EF_PACK_AFF) && (!(f & EF_RES_LATE) || (f & EF_PACK_PRE)));
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        pbits = shift_in(pbits, is_pos(pvs[j]));
        if (!no_neg) mbits = shift_in(mbits, is_neg(pvs[j]));
      }
    }
  }
}

// EP_MIDT, single-chunk kernels, full blocks: the sign test of a channel in TWO vector instructions (instead of
// v_lshl_add + v_cmp + v_addc).  With nz = 2q + p the lane's non-zero inputs, T = 2a + t and `cnt` the channel's popcount:
//   agreements   (NN):  2 cnt - nz >= T  <=>  cnt - q >= ceil(T / 2) + (p & T even)
//   disagreements    :  nz - 2 cnt >= T  <=>  cnt - q <  floor(-T / 2) + 1 + (p & T odd)
// Both are "cnt' < A + carry" with cnt' = cnt - q + kMidtBias (the popcount chain STARTS from kMidtBias - q: the VGPR addend
// of its first v_bcnt, free), A per channel from the thresholds table (thr[4o + 2] / thr[4o + 3], csrc/thresholds.hip)
// and the carry mask chosen from T's parity on the scalar unit (one bit test on the block's parity word):
//   v_subbrev_co_u32 tmp, vcc, A, cnt', vcc     vcc <- borrow = (cnt' < A + carry), unsigned: hence the bias
//   v_addc_co_u32    word, vcc, word, word, vcc  shifts the borrow in
// A gfx9 vector instruction reads ONE scalar operand and the carry-in is one, so A has to be a vector register: it is
// fetched as a wave-uniform BUFFER load (every lane the same address: one cache line, no vector-ALU work — a v_mov from
// the scalar value would be the third instruction again), eight per pass, issued before the pass's popcount loop.
// (The parity coupling cannot be removed: 2 cnt - nz has the parity of the LANE's nz, so the integer bound on cnt
// differs by one between lanes of even and odd nz exactly when T has the other parity.)
// The borrow is the bit itself for the disagreement form and its complement for the agreement form (the complement
// joins the block's flip word, which is XORed into the finished word anyway).  "never" (T = 2^30) and "always"
// (T = -kmax) need no special case: A saturates at 0 from below (cnt' >= 1) and nothing reaches 2^29.
// Checked against the plain integer test by enumeration on the host (tests/test_midt_cpu.py) and on the device bit for
// bit (tests/test_gpu_fused.py: the threshold planes equal the float epilogue's).
constexpr int kMidtBias = 1 << 20;  // > K / 2 for every K the thresholds kernel accepts (K < 2^20)
constexpr int kThrStride = 4;       // int32 per channel of the thresholds table: {T, flip | parity word of the block, A_nn, A_tp}
// J: the channel's bit in `parity` (the pass's slice of the block's parity word, thresholds.hip).
template <bool NN, int J>
__device__ __forceinline__ uint32_t midt2_shift_in(uint32_t word, int cnt, int A, uint32_t parity,
                                                   unsigned long long oddmask) {
#if defined(__HIP_DEVICE_COMPILE__)
  int tmp;
  if constexpr (NN) {
    asm("s_bitcmp0_b32 %[par], %[j]\n\t"          // SCC = T even
        "s_cselect_b64 vcc, %[odd], 0\n\t"
        "v_subbrev_co_u32 %[tmp], vcc, %[a], %[cnt], vcc\n\t"
        "v_addc_co_u32 %[w], vcc, %[w], %[w], vcc"
        : [w] "+v"(word), [tmp] "=&v"(tmp)
        : [par] "s"(parity), [j] "i"(J), [odd] "s"(oddmask), [cnt] "v"(cnt), [a] "v"(A)
        : "vcc", "scc");
  } else {
    asm("s_bitcmp1_b32 %[par], %[j]\n\t"          // SCC = T odd
        "s_cselect_b64 vcc, %[odd], 0\n\t"
        "v_subbrev_co_u32 %[tmp], vcc, %[a], %[cnt], vcc\n\t"
        "v_addc_co_u32 %[w], vcc, %[w], %[w], vcc"
        : [w] "+v"(word), [tmp] "=&v"(tmp)
        : [par] "s"(parity), [j] "i"(J), [odd] "s"(oddmask), [cnt] "v"(cnt), [a] "v"(A)
        : "vcc", "scc");
  }
  return word;
#else
  return word;
#endif
}

// Residual (shortcut) values of NACC channels for this lane's pixel.  Called at the START of a
// pass, before the popcount loop, so the loads land while the vector ALU is busy: the epilogue
// then finds them in registers instead of stalling on HBM once per pass.
template <int NACC, int EP, bool FULL = false>
__device__ __forceinline__ void prefetch_residual(const Geo& g, const Pix& px, int o0,
                                                  const EpiArgs& e, float (&resv)[NACC]) {
  const int f = ep_flags(EP, g.flags);
  if (EP == EP_PLAIN || !(f & EF_RES) || (f & EF_RAW)) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) resv[j] = 0.0f;
    return;
  }
  const int hw = g.Ho * g.Wo;
  const unsigned lane_off = px.out_base * 4u;  // bytes
#pragma unroll
  for (int j = 0; j < NACC; ++j)
    // lanes past the last pixel load too (they were clamped to it): they must compute the same sign bits as the
    // live copy, because store_packed() lets them store
    if constexpr (FULL) resv[j] = buf_ld(make_rsrc(e.res), lane_off, (unsigned)(o0 + j + g.c_off) * (unsigned)hw * 4u);
    else resv[j] = (o0 + j < g.O) ? ld_off(e.res + (size_t)(o0 + j + g.c_off) * hw, lane_off) : 0.0f;
}

// sign(y) of one 32-channel block: one half of a uint64 word of the [n][group][y][x] output planes.
// `rev`: the words were built by shift-in (straight-line epilogue, all 32 channels of the block): bit-reversed.
// `xorw` (EP_MIDT): the flip bits of the block's channels, applied to the finished P word.
// `keep` (two-instruction threshold form): the bits of channels that exist, applied last.
__device__ __forceinline__ void store_packed(const Geo& g, const Pix& px, int ob, uint32_t pbits,
                                             uint32_t mbits, const EpiArgs& e, bool rev = false, uint32_t xorw = 0u,
                                             uint32_t keep = ~0u) {
  // lanes past the last pixel hold a copy of it (decode_pixel): they store the same word to the same place
  if (!(g.flags & EF_PACK) || (g.flags & EF_RAW)) return;
  if (rev) {
    pbits = __builtin_bitreverse32(pbits);
    mbits = __builtin_bitreverse32(mbits);
  }
  pbits = (pbits ^ xorw) & keep;
  const int hw = g.Ho * g.Wo;
  const unsigned w = (((px.pk_base + (unsigned)((ob >> 1) * hw)) << 1) + (unsigned)(ob & 1)) * 4u;  // bytes
  st_off(e.outP, w, pbits);
  st_off(e.outM, w, mbits);
}

// Same, for a wave that produced only part `part` of PARTS of the block's 32 channels (their bits
// already sit at their final position inside the 32-bit word).
// `rev`: the part's 32 / PARTS bits were built by shift-in from bit 0 (bit-reversed, not yet at their position).
template <int PARTS>
__device__ __forceinline__ void store_packed_part(const Geo& g, const Pix& px, int ob, int part,
                                                  uint32_t pbits, uint32_t mbits, const EpiArgs& e, bool rev = false,
                                                  uint32_t xorw = 0u, uint32_t keep = ~0u) {
  static_assert(PARTS == 2 || PARTS == 4, "16- or 8-bit pieces");
  if (!(g.flags & EF_PACK) || (g.flags & EF_RAW)) return;
  if (rev) {  // bits 0 .. 32/PARTS-1 reversed -> the top of bitreverse32; move them to the part's position
    pbits = (__builtin_bitreverse32(pbits) >> (32 - 32 / PARTS)) << ((32 / PARTS) * part);
    mbits = (__builtin_bitreverse32(mbits) >> (32 - 32 / PARTS)) << ((32 / PARTS) * part);
  }
  pbits = (pbits ^ (xorw & (((1u << (32 / PARTS)) - 1u) << ((32 / PARTS) * part)))) & keep;
  const int hw = g.Ho * g.Wo;
  const size_t w = (((size_t)px.pk_base + (size_t)(ob >> 1) * hw) << 1) + (ob & 1);
  constexpr int BITS = 32 / PARTS;
  if constexpr (PARTS == 2) {
    reinterpret_cast<uint16_t*>(e.outP)[w * 2 + part] = (uint16_t)(pbits >> (BITS * part));
    reinterpret_cast<uint16_t*>(e.outM)[w * 2 + part] = (uint16_t)(mbits >> (BITS * part));
  } else {
    reinterpret_cast<uint8_t*>(e.outP)[w * 4 + part] = (uint8_t)(pbits >> (BITS * part));
    reinterpret_cast<uint8_t*>(e.outM)[w * 4 + part] = (uint8_t)(mbits >> (BITS * part));
  }
}

#define BNN_EPI_PARAMS                                                                          \
  const float *__restrict__ alpha, const float *__restrict__ bias, const float *__restrict__ scale, \
      const float *__restrict__ bn_a, const float *__restrict__ bn_b,                           \
      const float *__restrict__ prelu, const float *__restrict__ res, void *__restrict__ out,   \
      uint32_t *__restrict__ outP, uint32_t *__restrict__ outM,                                 \
      const float *__restrict__ pack_a, const float *__restrict__ pack_b, const int32_t *__restrict__ thr
#define BNN_EPI_INIT \
  EpiArgs epi{alpha, bias, scale, bn_a, bn_b, prelu, res, out, outP, outM, pack_a, pack_b, thr}

// The shortcut branch of a down-sampling residual block folded into the block's second convolution (bconv_sgpr_kernel<..,
// DS = true>): the residual of output channel o at a pixel is not read from an fp32 tensor but computed here,
//     r = fmaf(fmaf(alpha[o], dot1x1(sign planes sc_P, sc_W[o]), 0), bn_a[o], bn_b[o])
// with the float operations of the 1x1 convolution's own epilogue (EP_DS) — the fp32 shortcut tensor (103 MB at
// ResNet-18 layer2, batch 256, written once and read once) and the 1x1 launch disappear.  The shortcut input is the
// sign plane of an AvgPool of ReLU outputs: non-negative, P plane only, [N, ceil(C/64), Ho, Wo] uint64.
#define BNN_DS_PARAMS                                                                             \
  const uint32_t *__restrict__ dsP, const uint32_t *__restrict__ dsW, const float *__restrict__ ds_alpha, \
      const float *__restrict__ ds_a, const float *__restrict__ ds_b
struct ShortcutArgs {
  const uint32_t* P;
  const uint32_t* W;
  const float* alpha;
  const float* a;
  const float* b;
};

// The lane's pixel of the shortcut planes (ds_cw <= 8 words; unused registers stay 0) and its non-zero count.
__device__ __forceinline__ int load_shortcut_field(const Geo& g, const Pix& px, const ShortcutArgs& d, uint32_t (&dsr)[8]) {
  int nz = 0;
  if (g.ds_w > 0) {
    // un-pooled plane: sign(AvgPool2d(2, ceil_mode, count_include_pad=False)(x)) of x >= 0 is the OR of the window's
    // sign bits (what bnn_hip_orpool_packed writes) — four loads and three ORs per word here instead of a launch
    const unsigned hw = (unsigned)(g.ds_h * g.ds_w);
    const int y0 = 2 * px.oy, x0 = 2 * px.ox;
    const bool x1 = x0 + 1 < g.ds_w, y1 = y0 + 1 < g.ds_h;
    const unsigned base = (unsigned)px.n * (unsigned)(g.ds_cw >> 1) * hw + (unsigned)(y0 * g.ds_w + x0);
    const uint2* P2 = reinterpret_cast<const uint2*>(d.P);
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      uint2 v{0u, 0u};
      if (2 * gi < g.ds_cw) {
        const unsigned b = base + (unsigned)gi * hw;
        v = P2[b];
        const uint2 a = P2[x1 ? b + 1 : b], c = P2[y1 ? b + (unsigned)g.ds_w : b],
                    e = P2[(x1 && y1) ? b + (unsigned)g.ds_w + 1 : b];
        v.x |= a.x | c.x | e.x;
        v.y |= a.y | c.y | e.y;
      }
      dsr[2 * gi] = v.x;
      dsr[2 * gi + 1] = v.y;
      nz += __builtin_popcount(v.x) + __builtin_popcount(v.y);
    }
    return nz;
  }
  const unsigned hw = (unsigned)(g.Ho * g.Wo);
  const unsigned base = (unsigned)px.n * (unsigned)(g.ds_cw >> 1) * hw + (unsigned)px.r;  // uint64 words
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    uint2 v{0u, 0u};
    if (2 * gi < g.ds_cw) v = reinterpret_cast<const uint2*>(d.P)[base + (unsigned)gi * hw];
    dsr[2 * gi] = v.x;
    dsr[2 * gi + 1] = v.y;
    nz += __builtin_popcount(v.x) + __builtin_popcount(v.y);
  }
  return nz;
}

// The shortcut values of channels o0 .. o0+NACC-1 (wave-uniform weights and constants: scalar loads).
template <int NACC, int CW>
__device__ __forceinline__ void shortcut_values_cw(const Geo& g, int o0, const ShortcutArgs& d, const uint32_t (&dsr)[8],
                                                   int nz, float (&resv)[NACC]) {
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    if (j % 4 == 0) __builtin_amdgcn_sched_barrier(0);  // four channels' weights and constants in scalar registers at a time
    const int o = o0 + j;
    float r = 0.0f;
    if (o < g.O) {
      int agree = 0;
#pragma unroll
      for (int i = 0; i < CW; ++i) agree += __builtin_popcount(d.W[(size_t)o * CW + i] & dsr[i]);
      const float dot = (float)(2 * agree - nz);           // non-negative input: dot = 2 * agreements - non-zeros
      r = fmaf(fmaf(d.alpha[o], dot, 0.0f), d.a[o], d.b[o]);  // EP_DS: alpha * dot (no bias), then the folded BatchNorm
    }
    resv[j] = r;
  }
}
template <int NACC>
__device__ __forceinline__ void shortcut_values(const Geo& g, int o0, const ShortcutArgs& d, const uint32_t (&dsr)[8],
                                                int nz, float (&resv)[NACC]) {
  if (g.ds_cw == 2) shortcut_values_cw<NACC, 2>(g, o0, d, dsr, nz, resv);
  else if (g.ds_cw == 4) shortcut_values_cw<NACC, 4>(g, o0, d, dsr, nz, resv);
  else shortcut_values_cw<NACC, 8>(g, o0, d, dsr, nz, resv);
}

// ---------------------------------------------------------------------------------
// Tiled kernel, weights streamed through SGPRs (scalar cache).  Best when all waves in
// flight share one small weight block (large images, few output channels): BASELINE config 2.
// ---------------------------------------------------------------------------------
// Streams NACC x NW wave-uniform weight words (one contiguous run) through two 16-word SGPR
// buffers and accumulates the disagreement counts of NACC output channels.  SMEM returns out of
// order, so the only usable wait is lgkmcnt(0): `cur` is touched first so that this wait lands
// BEFORE block b+1 is requested; b+1 then has the whole VALU block (32 instructions) to arrive.
#ifndef BNN_NT_STORE  // fused epilogue: fp32 stores with the non-temporal hint
#define BNN_NT_STORE 0
#endif
#ifndef BNN_MULTI_RES_EARLY
#define BNN_MULTI_RES_EARLY 0
#endif

// One block of the weight stream = WB wave-uniform words, fetched as s_load_dwordx16/x8/x4 pieces.
template <int WB>
struct WStream {
  uint32_t v[WB];
};
template <int WB>
__device__ __forceinline__ void load_wblock(const uint32_t* __restrict__ src, WStream<WB>& d) {
  static_assert(WB % 4 == 0, "whole dwordx4 pieces");
  constexpr int N16 = WB / 16, R = WB % 16;
#pragma unroll
  for (int i = 0; i < N16; ++i) {
    const WBlock<16> t = *reinterpret_cast<const WBlock<16>*>(src + 16 * i);
#pragma unroll
    for (int e = 0; e < 16; ++e) d.v[16 * i + e] = t.v[e];
  }
  if constexpr (R >= 8) {
    const WBlock<8> t = *reinterpret_cast<const WBlock<8>*>(src + 16 * N16);
#pragma unroll
    for (int e = 0; e < 8; ++e) d.v[16 * N16 + e] = t.v[e];
  }
  if constexpr (R % 8 == 4) {
    const WBlock<4> t = *reinterpret_cast<const WBlock<4>*>(src + WB - 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) d.v[WB - 4 + e] = t.v[e];
  }
}

#ifndef BNN_OUT4_MINW  // waves per SIMD the conv2-type kernel on a 128-channel P-only field is allocated for
#define BNN_OUT4_MINW 5
#endif
#ifndef BNN_RES_UNROLL  // conv2-type single-chunk kernels: passes per iteration of the (otherwise rolled) pass loop
#define BNN_RES_UNROLL 1
#endif
#ifndef BNN_WSTREAM_BLOCK  // preferred words per block of the scalar weight stream
#define BNN_WSTREAM_BLOCK 32
#endif
constexpr int pick_wblock(int total) {
  constexpr int pref = BNN_WSTREAM_BLOCK;
  for (int wb = pref; wb >= 4; wb -= 4)
    if (total % wb == 0) return wb;
  return 4;
}

// The look-ahead is ONE block (see above), so the block must be long enough for the next one to
// arrive while it is consumed: a scalar load that misses to L2 takes ~700 cycles, a block of WB
// words keeps the SIMD busy for 8*WB cycles per resident wave.  With 16-word blocks that needs
// >= 6 waves per SIMD; 32-word blocks (two s_load_dwordx16) need 3.
// NN: `acc` counts AGREEMENTS, popcount(w & p) (one 4-byte VOP2 v_and + v_bcnt), instead of
// disagreements; the caller turns them into the dot product with dot = 2*agree - nonzeros.
// USEED: the counts start from the wave-uniform `useed` (single-chunk kernels) instead of from acc[].
// ONECHAIN: one popcount chain per channel instead of an even and an odd one (no t0 + t1 add at the end; the
// threshold epilogue then needs three instructions per channel).
// ILP: words whose v_bitop3 are issued back to back BEFORE their v_bcnt (1 = each v_bcnt right behind the v_bitop3 it
// depends on: fine when several waves interleave on the SIMD, but a wave that runs alone — the others blocked on
// memory, or gone at the tail of a workgroup — then waits out the VALU latency on every pair).
#ifndef BNN_STREAM_ILP  // 2: config-2 kernel 208 -> 201 us (0.905 -> 0.936 of the int-ALU roofline), round 3
#define BNN_STREAM_ILP 2
#endif
// VSEED (with USEED and ONECHAIN): the seed is a per-lane value (a VGPR addend of the chain's first v_bcnt: as free as the scalar one).
template <int NW, int NACC, bool NN = false, bool USEED = false, bool ONECHAIN = false, int ILP = BNN_STREAM_ILP,
          bool VSEED = false>
__device__ __forceinline__ void stream_weights(const uint32_t* __restrict__ wrun,
                                               const uint32_t (&pr)[NW], const uint32_t (&mr)[NW],
                                               int (&acc)[NACC], [[maybe_unused]] int useed = 0) {
  static_assert(!VSEED || (USEED && ONECHAIN), "a per-lane seed: single-chunk, one chain per channel");
  constexpr int WB = pick_wblock(NACC * NW);
  constexpr int NB = NACC * NW / WB;
  static_assert((NACC * NW) % WB == 0, "weight run must be a whole number of blocks");
  static_assert(ILP == 1 || ILP == 2 || ILP == 4, "words per group");
  static_assert(WB % ILP == 0, "whole groups per block");
  WStream<WB> cur;
  load_wblock<WB>(wrun, cur);
  int t0 = 0, t1 = 0;  // two accumulation chains per channel (even / odd words)
  static_for<NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(cur.v[0]), "s"(cur.v[WB > 16 ? 16 : 0]), "s"(cur.v[WB > 32 ? 32 : 0]),
                 "s"(cur.v[WB - 1]));  // one word of every piece: all of `cur` has landed
#endif
    __builtin_amdgcn_sched_barrier(0);
    WStream<WB> nxt;
    if constexpr (b + 1 < NB) load_wblock<WB>(wrun + (b + 1) * WB, nxt);
    __builtin_amdgcn_sched_barrier(0);
    static_for<WB / ILP>([&](auto gc) {
      constexpr int e0 = decltype(gc)::value * ILP;
      uint32_t d[ILP];
      static_for<ILP>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int i = (b * WB + e0 + k) % NW;
        d[k] = NN ? (cur.v[e0 + k] & pr[i]) : disagree(cur.v[e0 + k], mr[i], pr[i]);
      });
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (ILP == 2) asm volatile("" : "+v"(d[0]), "+v"(d[1]));
      if constexpr (ILP == 4) asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
#endif
      static_for<ILP>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int f = b * WB + e0 + k;
        constexpr int j = f / NW, i = f % NW;
        if constexpr (ONECHAIN) {
          acc[j] = (USEED && i == 0) ? (VSEED ? popc_acc(d[k], useed) : popc_acc_s(d[k], useed)) : popc_acc(d[k], acc[j]);
          (void)t0; (void)t1;
        } else {
          // the even chain continues from the running count (acc[j]: 0, the count seed, or the previous chunks' sum);
          // the first word of the odd chain uses the inline-constant form (v_bcnt d, 0)
          if constexpr (i == 0) t0 = USEED ? popc_acc_s(d[k], useed) : popc_acc(d[k], acc[j]);
          else if constexpr (i == 1) t1 = __builtin_popcount(d[k]);
          else if constexpr (i & 1) t1 = popc_acc(d[k], t1);
          else t0 = popc_acc(d[k], t0);
          if constexpr (i == NW - 1) acc[j] = t0 + (NW > 1 ? t1 : 0);
        }
      });
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (b + 1 < NB) cur = nxt;
  });
}

// Zero weights (sign(0) == 0: pruned nets, bnn/ops.py:66,136).  A second wave-uniform stream carries the
// non-zero mask Z (same layout as the sign bits).  Per word and output channel:
//     D_o  += popcount( ((W & M) | (~W & P)) & Z )     disagreements among positions whose weight is non-zero
//     nz_o += popcount( (P | M) & Z )                  non-zero products: now a per-CHANNEL count
// and dot = nz_o - 2*D_o.  Five VALU instructions per 32 MACs instead of two (bitop3, and, bcnt, bitop3, bcnt),
// but still the register-resident field and the scalar weight stream — the shape-generic kernel that pruned nets
// used to fall to is several times slower.  16-word blocks: two streams x (cur, nxt) = 64 SGPRs.
template <int NW, int NACC>
__device__ __forceinline__ void stream_weights_wz(const uint32_t* __restrict__ wrun,
                                                  const uint32_t* __restrict__ zrun,
                                                  const uint32_t (&pr)[NW], const uint32_t (&mr)[NW],
                                                  int (&acc)[NACC], int (&nzacc)[NACC]) {
  constexpr int total = NACC * NW;
  constexpr int WB = total % 16 == 0 ? 16 : total % 8 == 0 ? 8 : 4;
  constexpr int NB = total / WB;
  static_assert(total % WB == 0, "weight run must be a whole number of blocks");
  WStream<WB> cur, zcur;
  load_wblock<WB>(wrun, cur);
  load_wblock<WB>(zrun, zcur);
  int t0 = 0, t1 = 0;
  static_for<NB>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(cur.v[0]), "s"(cur.v[WB - 1]), "s"(zcur.v[0]), "s"(zcur.v[WB - 1]));
#endif
    __builtin_amdgcn_sched_barrier(0);
    WStream<WB> nxt, znxt;
    if constexpr (b + 1 < NB) {
      load_wblock<WB>(wrun + (b + 1) * WB, nxt);
      load_wblock<WB>(zrun + (b + 1) * WB, znxt);
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<WB>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      constexpr int f = b * WB + e;
      constexpr int j = f / NW, i = f % NW;
      const uint32_t d = disagree(cur.v[e], mr[i], pr[i]) & zcur.v[e];
      const uint32_t n = (pr[i] | mr[i]) & zcur.v[e];
      if constexpr (i == 0) { t0 = __builtin_popcount(d); t1 = __builtin_popcount(n); }
      else { t0 = popc_acc(d, t0); t1 = popc_acc(n, t1); }
      if constexpr (i == NW - 1) { acc[j] += t0; nzacc[j] += t1; }
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (b + 1 < NB) { cur = nxt; zcur = znxt; }
  });
}

// ---------------------------------------------------------------------------------
// host side: geometry of one launch
// ---------------------------------------------------------------------------------
// q / d for 0 <= q < 2^31 as (q * m) >> (32 + s)  (Granlund & Montgomery: m = ceil(2^(31+l) / d), l = ceil(log2 d),
// s = l - 1; m < 2^32 for d >= 2); d == 1 is flagged with s = -1.
// host-side launch helpers shared by bconv.hip and legacy/bconv_lds.hip
#define BNN_EPI_ACTUALS \
  p.alpha, p.bias, p.scale, p.bn_a, p.bn_b, p.prelu, p.res, p.out, p.outP, p.outM, p.pack_a, p.pack_b, p.thr
#define BNN_DS_ACTUALS p.ds_P, p.ds_W, p.ds_alpha, p.ds_a, p.ds_b

// grid.y: one block per 32 output channels; in pack mode also the (all-zero) tail words of the
// packed output row so that every word of the next layer's input is written.
inline unsigned oblocks(const ConvP& p) {
  const unsigned nb = (p.O + kOCB - 1) / kOCB;
  return (p.outP && p.outM) ? (unsigned)(2 * ((p.O + 63) / 64)) : nb;
}

inline void div_magic(uint32_t d, uint32_t& m, int& s) {
  if (d <= 1) { m = 0; s = -1; return; }
  int l = 0;
  while ((1ull << l) < d) ++l;
  m = (uint32_t)(((1ull << (31 + l)) + d - 1) / d);
  s = l - 1;
}

// The tiled kernels multiply indices with v_mul_u32_u24 / v_mul_i32_i24 (decode_pixel<true>): every factor must stay
// below 2^23.  Anything larger (4096 x 2048 images, batches of millions) takes the shape-generic kernel.
inline bool small_indices(const ConvP& p) {
  const long long lim = 1ll << 23;
  const long long c_tot = p.c_tot > 0 ? p.c_tot : p.O;
  return (long long)p.N * c_tot < lim && (long long)p.N * ((p.O + 63) / 64) < lim &&
         (long long)p.N * (p.cw32 >> 1) < lim && (long long)p.Ho * p.Wo < lim &&
         (long long)(p.H + p.ph + 2) * (p.Wd + p.pw + 2) < lim && (long long)p.sh * p.Wd < lim && p.sw < 4096;
}

inline Geo make_geo(const ConvP& p) {
  Geo g;
  div_magic((uint32_t)(p.Ho * p.Wo), g.m_hw, g.s_hw);
  div_magic((uint32_t)p.Wo, g.m_wo, g.s_wo);
  div_magic((uint32_t)(((p.npix + kWave - 1) / kWave + 7) / 8), g.m_tpx, g.s_tpx);
  g.in_bytes = (unsigned)((long long)p.N * (p.cw32 >> 1) * p.H * p.Wd * 8);  // capi.hip: below 2^32 - 8
  g.N = p.N; g.H = p.H; g.Wd = p.Wd; g.Ho = p.Ho; g.Wo = p.Wo; g.O = p.O;
  g.KH = p.KH; g.KW = p.KW; g.sh = p.sh; g.sw = p.sw; g.ph = p.ph; g.pw = p.pw;
  g.dh = p.dh; g.dw = p.dw; g.cw32 = p.cw32; g.cwc = p.cwc; g.nchunk = p.nchunk;
  g.npix = p.npix;
  g.cw32_out = 2 * ((p.O + 63) / 64);
  g.tiles = (p.npix + kWave - 1) / kWave;
  g.tiles_per_xcd = (g.tiles + 7) / 8;
  int f = 0;
  if (p.raw) f |= EF_RAW;
  if (p.bias) f |= EF_BIAS;
  if (p.scale) f |= EF_SCALE;
  if (p.bn_a && p.bn_b) f |= EF_BN;
  if (p.res || p.ds_P) f |= EF_RES;  // (a folded shortcut convolution supplies the residual in registers)
  if (p.relu) f |= EF_RELU;
  if (p.prelu) f |= EF_PRELU;
  if (p.out) f |= EF_OUTF;
  if (p.outP && p.outM) f |= EF_PACK;
  if (p.res && (p.eflags & BNN_HIP_EPI_RES_AFTER_ACT)) f |= EF_RES_LATE;
  if (p.res && (p.eflags & BNN_HIP_EPI_RES_AFTER_ACT) && (p.eflags & BNN_HIP_EPI_PACK_BEFORE_RES)) f |= EF_PACK_PRE;
  if (p.pack_a && p.pack_b) f |= EF_PACK_AFF;
  if (p.eflags & BNN_HIP_EPI_PACK_RELU) f |= EF_PACK_RELU;
  g.ds_cw = p.ds_P ? 2 * ((p.ds_C + 63) / 64) : 0;
  g.ds_h = p.ds_P ? p.ds_inH : 0;
  g.ds_w = p.ds_P ? p.ds_inW : 0;
  g.c_off = p.c_off;
  g.c_tot = p.c_tot > 0 ? p.c_tot : p.O;
  g.flags = f;
  return g;
}

}  // namespace bnn
