// grad.hip — binary-aware gradient kernels of a fake-binarised 3x3 / pad 1 (stride 1 or 2) or 1x1 / pad 0 convolution
// (reference: the autograd graph of bnn/layers/conv.py:90-97 with the straight-through
// estimator of bnn/ops.py:68-73, as trained by examples/imagenet.py:337-384).
//
// For out = conv2d(sign(x), What), What = sign(Wc) * alpha, and an incoming gradient g = dL/d out:
//
//     dL/d xhat [n,c,y,x]   = sum_{o,ky,kx} g[n,o,y+1-ky,x+1-kx] * What[o,c,ky,kx]          ("dgrad")
//     dL/d x                = dL/d xhat * 1[|x| < 1]                                       (hard-tanh STE, ops.py:68-73)
//     dL/d What[o,c,ky,kx]  = sum_{n,y,x} g[n,o,y,x] * sign(x)[n,c,y+ky-1,x+kx-1]           ("wgrad")
//
// Both are GEMMs with ONE real-valued operand (g) and one operand that is exactly {-1,0,+1} (sign(Wc), sign(x)) —
// the library evaluates them as full fp32 convolutions (157 TF peak).  Here the real operand is split into THREE
// bf16 terms hi + mid + lo (3 x 8 = 24 mantissa bits, and bf16 has the EXPONENT RANGE OF fp32: gradients of a
// mean-reduced loss at batch 256 are 1e-5 .. 1e-9, where an fp16 split — round 2 — lost everything below its 2^-24
// quantum), the ternary operand is EXACT in bf16, every product of two bf16 values is exact in fp32, and each
// product of the GEMM is three v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the rounding class of an fp32
// convolution at any magnitude, at 3/16 of the fp32-MFMA matrix time (the kernels are bound by the VALU work of the
// fills, not by the matrix pipe).  alpha[o] is folded into the real operand (g' = alpha[o] * g, one fp32 multiply
// when the tile enters LDS), so the weight operand stays ternary.
//
// K-slots.  A row of W <= 64 pixels is given a power-of-two slot (8, 16, 32 or 64 pixels, zero filled); a 64-pixel
// chunk of the GEMM's pixel dimension is 64/slot consecutive image rows.  An 8-element MFMA fragment never
// straddles an image row, and 7x7 .. 56x56 images all use >= 87.5 % of the chunk.
//
// Measured (MI355X, ResNet-18 224x224 batch 256, tools/bench_grad.py / tools/bench_train.py, round 3): dgrad 180-400 us
// per stride-1 layer (the library's fp32 backward: 530-630), 270-540 on the stride-2 layers (350-440), wgrad 205-335 us
// (310-705); the whole training step 30.2 ms against 41.7 ms with the library's gradient convolutions and 48.4 ms for
// the pure torch composition.  Both kernels were bound by HOW they loaded, not by the matrix pipe: guarded loads behind
// branches are waited for one by one (the dgrad epilogue paid one HBM round trip per output channel: 16-32 in a row),
// and 64-bit index arithmetic + selects per element were ~20 VALU instructions per loaded element.  Now every fill item
// keeps its place for the life of the workgroup: per-lane byte offsets computed once, wave-uniform offsets per block /
// chunk in scalar registers, buffer loads whose out-of-range offsets ARE the zero padding.
//
// Stride 2 (the first conv of a down-sampling stage) uses the same kernels: wgrad reads sign(x) at stride 2 (its
// shifted LDS copies are de-interleaved); dgrad runs the four parity classes of the x pixels as four convolutions over
// the grid of g with 1, 2, 2 and 4 taps (dgrad_kernel<.., S2 = true>, grid.z = class) — until round 3 it multiplied a
// zero-upsampled g with all nine taps (3/4 of its MFMAs and of its fill were zeros: 536 -> 380, 276 -> 206, 297 -> 144 us).  1x1 / stride 1 convolutions (the shortcut branches behind an AvgPool:
// 0.4 % of a ResNet-18's MACs) are the same kernels with one tap and no halo; the weight-gradient kernel then
// takes 128 input channels per workgroup instead of 32 so that a wave still has 8 accumulator tiles per g fragment.
#include "bconv_core.h"  // buffer-descriptor helpers

namespace bnn {

namespace grad {
constexpr int NT = 256;          // 4 waves
constexpr int APIX = 40;         // dgrad patch: halves per pixel (32 channels + 8 pad: 80 B, conflict-free b128 reads)
constexpr int AROW = 72;         // wgrad: halves per row of 64 k-slots (+8 pad: 144 B)
constexpr int SROW = 68;         // dgrad output staging: floats per channel row of 64 pixels
}  // namespace grad

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) __bf16;   // eight bf16 values: one MFMA fragment (name kept: 16 bytes)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u16 = unsigned short;                                   // the LDS planes are arrays of 16-bit containers
constexpr unsigned short kBf16One = 0x3F80, kBf16MinusOne = 0xBF80;

__device__ __forceinline__ half8 bf16_const8(unsigned short v) {
  const unsigned d = (unsigned)v * 0x00010001u;
  return __builtin_bit_cast(half8, u32x4{d, d, d, d});
}

struct GradGeo {
  int N, O, C;
  int H, W;      // the pixel domain the 64-slot chunks tile: x / gx for dgrad, g for wgrad
  int Hx, Wx;    // input tensor x (and gx): [N,C,Hx,Wx]
  int Hg, Wg;    // gradient tensor g:       [N,O,Hg,Wg],  Hg = (Hx - 1) / st + 1
  int st;        // convolution stride (1 or 2)
  int slot, RR;  // pixels per row slot (pow2 >= W, >= 8); row slots per 64-pixel chunk = 64 / slot
  int R;         // image rows a chunk advances by = min(RR, H)
  int chunks;    // chunks per image = ceil(H / R)
  int gshift;    // log2(slot / 8): 8-pixel groups per row slot
  unsigned g_bytes;  // bytes of the gradient tensor g (the range of its buffer descriptor)
  unsigned x_bytes;  // bytes of x / gx
  int cw64;          // XP kernels: 64-channel groups of the bit planes that stand for x (csrc/pack_ste.hip)
  unsigned p_bytes;  // bytes of one such plane, [N][cw64][Hx][Wx] uint64
};

// v = hi + mid + lo (+ at most 2^-24 |v|): each term the TRUNCATION of what is left to bf16 (the remainders are exact
// in fp32, so the three terms carry the leading 24 mantissa bits; truncation needs no rounding logic and one
// v_perm_b32 packs two terms into a dword).  NaN / infinities travel in `hi` (the remainder of an infinity is NaN,
// as in the fp32 product it stands for).
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& mid, half8& lo) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = __float_as_uint(v[e]) & 0xFFFF0000u;
    const float r1 = v[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m[e]);
    l[e] = __float_as_uint(r2);
  }
  u32x4 ph, pm, pl;
#pragma unroll
  for (int e = 0; e < 4; ++e) {  // the upper halves of two dwords -> one dword
    ph[e] = __builtin_amdgcn_perm(h[2 * e + 1], h[2 * e], 0x07060302u);
    pm[e] = __builtin_amdgcn_perm(m[2 * e + 1], m[2 * e], 0x07060302u);
    pl[e] = __builtin_amdgcn_perm(l[2 * e + 1], l[2 * e], 0x07060302u);
  }
  hi = __builtin_bit_cast(half8, ph);
  mid = __builtin_bit_cast(half8, pm);
  lo = __builtin_bit_cast(half8, pl);
}

// ------------------------------------------------------------------------------------------------- weight operand
// alpha[o] = max |What[o,:,:,:]| (What = +-alpha or 0 exactly) and sign(What) in MFMA B-fragment order for dgrad:
//   Bp[ob][tap][cs][lane][e] = sign(What[o = 32 ob + 8 (lane>>4) + e][c = 16 cs + (lane&15)][tap])   (fp16)
__global__ __launch_bounds__(64) void grad_alpha_kernel(const float* __restrict__ what, int per_o,
                                                        float* __restrict__ alpha) {
  const float* p = what + (size_t)blockIdx.x * per_o;
  float m = 0.0f;
  for (int i = threadIdx.x; i < per_o; i += 64) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) alpha[blockIdx.x] = m;
}

__global__ __launch_bounds__(64) void grad_pack_weight_kernel(const float* __restrict__ what, int O, int C,
                                                              int CS, int T, half8* __restrict__ Bp) {
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const int cs = blockIdx.x % CS, tap = (blockIdx.x / CS) % T, ob = blockIdx.x / (CS * T);
  const int c = 16 * cs + li;
  unsigned short b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int o = 32 * ob + 8 * lg + e;
    const float v = (o < O && c < C) ? what[((size_t)o * C + c) * T + tap] : 0.0f;
    b[e] = v > 0.0f ? kBf16One : v < 0.0f ? kBf16MinusOne : (unsigned short)0;
  }
  u32x4 pk;
#pragma unroll
  for (int e = 0; e < 4; ++e) pk[e] = (unsigned)b[2 * e] | ((unsigned)b[2 * e + 1] << 16);
  Bp[(size_t)blockIdx.x * 64 + lane] = __builtin_bit_cast(half8, pk);
}

// ------------------------------------------------------------------------------------------------- dgrad
// GEMM: M = pixels (one 64-slot chunk of one image per workgroup; 32 slots per wave), N = input channels c (32 * NSUB
// per wave, 64 * NSUB per workgroup, grid.y blocks), K = (o, tap): loop over blocks of 32 output channels; per block the
// chunk's rows +- 1 of g' = alpha[o] * g enter LDS as fp16 hi / lo, [pixel][32 o]; 9 taps = 9 k-steps of 32.
// Epilogue: accumulators -> LDS [c][pixel] -> coalesced NCHW stores with the STE mask 1[|x| < 1].
// S2 (stride 2, 3x3): the four parity classes (py, px) of the x pixels are four convolutions over the GRID OF g —
//     gx[2a + py, 2b + px] = sum over o and the taps with ky = py + 1, kx = px + 1 (mod 2) of
//                            g'[o, a + (py + 1 - ky)/2, b + (px + 1 - kx)/2] * sign(W)[o, c, ky, kx]
// with 1, 2, 2 and 4 taps (class = blockIdx.z; the chunks tile the g grid, the patch has one halo row below and one
// halo column to the right, the epilogue writes every second pixel of every second row).  Round 2 / 3a multiplied
// a zero-upsampled g with all nine taps: 3/4 of the MFMAs and of the fill were zeros.
// XP: `xin` is not the fp32 input but its straight-through mask as a bit plane, T = |x| < 1 ([N][C/64][H][W] uint64,
// csrc/pack_ste.hip): what a training step keeps of x for this kernel is 1 bit per element instead of 32.  Same results.
template <int NSUB, int KS, bool S2 = false, bool XP = false>
__global__ __launch_bounds__(grad::NT) void dgrad_kernel(const float* __restrict__ g,
                                                            const float* __restrict__ alpha,
                                                            const half8* __restrict__ Bp,
                                                            const float* __restrict__ xin,
                                                            float* __restrict__ gx, const GradGeo q) {
  using namespace grad;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  static_assert(!S2 || KS == 3, "parity classes: 3x3 / stride 2");
  constexpr int PD = KS / 2, T = KS * KS;
  constexpr int HALO = S2 ? 1 : 2 * PD;                   // patch rows / columns beyond the chunk's
  const int PW = q.W + HALO, PP = (q.R + HALO) * PW;     // patch width / pixels (halo included)
  const int py = S2 ? (int)(blockIdx.z >> 1) : 0, px = S2 ? (int)(blockIdx.z & 1) : 0;
  u16* pa_hi = reinterpret_cast<u16*>(lds_raw);
  u16* pa_mid = pa_hi + (size_t)PP * APIX;
  u16* pa_lo = pa_mid + (size_t)PP * APIX;
  float* stage = reinterpret_cast<float*>(lds_raw);  // reused after the K loop

  // (wave index through readfirstlane: what depends on it — channel group, alpha, plane offsets — is then wave-uniform
  // for the compiler too: scalar registers and scalar loads instead of per-lane copies and waterfall loops)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int n = blockIdx.x / q.chunks, y0 = (blockIdx.x - n * q.chunks) * q.R;
  const int HW = q.Hx * q.Wx, HGg = q.Hg * q.Wg;  // pixels of a channel plane of x / gx; of g
  const int CS = (q.C + 15) / 16, OB = (q.O + 31) / 32;
  // 2 x 2 register blocking: a wave owns TWO of the four 16-pixel sub-tiles and HALF of the workgroup's 4 NSUB channel
  // groups — 6 A-fragment reads (hi / mid / lo x 2 sub-tiles) per tap for 12 NSUB MFMAs.  (Until late round 4: all four
  // sub-tiles x a quarter of the channel groups, 12 reads for the same MFMAs — with 64 channels one 16-byte LDS read per
  // lane and MFMA, 256 B/clk per CU at the full matrix rate against the LDS's 128 B/clk: matrix pipe 27 % busy, LDS 61 %.)
  constexpr int NB = 2 * NSUB;                            // channel groups (B fragments) per wave
  const int ph = wave & 1, chh = wave >> 1;               // pixel half, channel half
  const int cs0 = blockIdx.y * 4 * NSUB + chh * NB;       // first 16-channel group of this wave

  // A-fragment base addresses of the 4 pixel sub-tiles: pixel m = 16 s + li -> (row ry, column x) of the chunk
  int abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = 2 * ph + i;
    const int m = 16 * s + li;
    int ry = m / q.slot, x = m - ry * q.slot;
    if (x >= q.W || ry >= q.R) { ry = 0; x = 0; }  // dead slot: any valid address, result is never stored
    // tap (ky,kx) reads patch pixel (ry+2PD-ky, x+2PD-kx); S2: (ry + dy, x + dx) from the patch's own origin
    abase[i] = S2 ? (ry * PW + x) * APIX + 8 * lg : ((ry + 2 * PD) * PW + (x + 2 * PD)) * APIX + 8 * lg;
  }

  f32x4 acc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ns = 0; ns < NB; ++ns) acc[i][ns] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fill roles, once per workgroup: wave w fills output channels 8 w .. 8 w + 7 of every 32-channel block (wave-
  // uniform: alpha and the channel's plane offset are scalar), lane l the patch pixels l, l + 64, ...  The pixel's byte
  // offset inside a channel plane of g (or "out of range": halo outside the image, odd positions of a zero-upsampled
  // stride-2 gradient) and its LDS address do not depend on the block: no address arithmetic in the K loop, the eight
  // loads of an item are buffer loads with a scalar plane offset (round 2/3a: ~60 VALU instructions of index arithmetic
  // and selects per item, the bulk of the kernel's instructions).
  constexpr int FK = 4;  // 64-pixel sweeps over the patch: (R + 2) (W + 2) <= 198
  const BufRsrc r_g = make_rsrc_sized(g, q.g_bytes);
  unsigned fvoff[FK];
  int flds[FK];
#pragma unroll
  for (int k = 0; k < FK; ++k) {
    const int pix = lane + 64 * k;
    const int pr = pix / PW, pc = pix - pr * PW;
    // (S2: the chunk's rows of g from its own origin, one halo row / column behind)
    const int y = S2 ? y0 + pr : y0 - PD + pr, x = S2 ? pc : pc - PD;
    const int yg = y, xg = x;
    const bool in = pix < PP && y >= 0 && x >= 0 && yg < q.Hg && xg < q.Wg;
    fvoff[k] = in ? (unsigned)(yg * q.Wg + xg) * 4u : 0xFFFFFFF0u;
    flds[k] = pix < PP ? pix * APIX + 8 * wave : -1;
  }
  const unsigned plane4 = (unsigned)HGg * 4u;
  // the taps of this launch: all T, or (S2) those of the parity class — ky = 1 | {0, 2}, kx = 1 | {0, 2}
  const int ntap = S2 ? (1 + py) * (1 + px) : T;
  auto tap_of = [&](int ti) {  // wave-uniform
    if (!S2) return ti;
    const int iy = px ? ti >> 1 : ti, ix = px ? ti & 1 : 0;
    return (py ? 2 * iy : 1) * KS + (px ? 2 * ix : 1);
  };

  for (int ob = 0; ob < OB; ++ob) {
    __syncthreads();  // previous block's patch is consumed
    auto load_w = [&](half8 (&dst)[NB], int tap) {
#pragma unroll
      for (int ns = 0; ns < NB; ++ns) {
        const int cs = cs0 + ns;
        dst[ns] = cs < CS ? Bp[((size_t)(ob * T + tap) * CS + cs) * 64 + lane] : bf16_const8(0);
      }
    };
    half8 b[NB];
    load_w(b, tap_of(0));
    const int o0 = 32 * ob + 8 * wave;
    float av[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) av[e] = o0 + e < q.O ? alpha[o0 + e] : 0.0f;
    // (all sweeps' loads up front — one round trip per block instead of one per sweep — measured 2-5 % slower: the dead
    // sweeps of the small patches then issue loads too)
#pragma unroll
    for (int k = 0; k < FK; ++k) {
      if (64 * k >= PP) break;  // workgroup-uniform
      float v[8];
      unsigned soff = (unsigned)(n * q.O + o0) * plane4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // channels past O: an out-of-range offset reads 0 (a select, not a branch: behind a branch every load is
        // waited for before the next one is issued)
        const unsigned vo = o0 + e < q.O ? fvoff[k] : 0xFFFFFFF0u;
        v[e] = buf_ld(r_g, vo, o0 + e < q.O ? soff : 0u) * av[e];
        soff += plane4;
      }
      half8 hi, mid, lo;
      split8(v, hi, mid, lo);
      if (flds[k] >= 0) {
        *reinterpret_cast<half8*>(pa_hi + flds[k]) = hi;
        *reinterpret_cast<half8*>(pa_mid + flds[k]) = mid;
        *reinterpret_cast<half8*>(pa_lo + flds[k]) = lo;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int ti = 0; ti < ntap; ++ti) {
      const int tap = tap_of(ti);
      const int ky = tap / KS, kx = tap - ky * KS;
      // (S2: dy = (py + 1 - ky) / 2, added; else the flipped tap, subtracted)
      const int toff = S2 ? -((((py + 1 - ky) >> 1) * PW + ((px + 1 - kx) >> 1)) * APIX) : (ky * PW + kx) * APIX;
      // the NEXT tap's sign(W) fragments are requested before this tap's MFMAs (tap 0's before the fill): a load in
      // front of its own MFMAs exposed one L2 round trip per tap, 18 per workgroup at 64 channels
      half8 bn[NB];
#pragma unroll
      for (int ns = 0; ns < NB; ++ns) bn[ns] = b[ns];
      if (ti + 1 < ntap) load_w(bn, tap_of(ti + 1));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const half8 ah = *reinterpret_cast<const half8*>(pa_hi + abase[i] - toff);
        const half8 am = *reinterpret_cast<const half8*>(pa_mid + abase[i] - toff);
        const half8 al = *reinterpret_cast<const half8*>(pa_lo + abase[i] - toff);
#pragma unroll
        for (int ns = 0; ns < NB; ++ns) {  // smallest terms first
          acc[i][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b[ns], acc[i][ns], 0, 0, 0);
          acc[i][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, b[ns], acc[i][ns], 0, 0, 0);
          acc[i][ns] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b[ns], acc[i][ns], 0, 0, 0);
        }
      }
#pragma unroll
      for (int ns = 0; ns < NB; ++ns) b[ns] = bn[ns];
    }
  }
  __syncthreads();  // the patch is dead: its LDS becomes the [channel][pixel] staging tile
  // D layout: column = li (channel within the 16-group), row = 4 lg + r (pixel within the sub-tile)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ns = 0; ns < NB; ++ns)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stage[((chh * NB + ns) * 16 + li) * SROW + 16 * (2 * ph + i) + 4 * lg + r] = acc[i][ns][r];
  __syncthreads();
  // A thread keeps ONE pixel of the chunk (its lane) and walks the channels wave, wave + 4, ...: the STE operand x of
  // all its 16 NSUB outputs is requested first (buffer loads: a dead pixel / channel is an out-of-range offset, the
  // plane offset is scalar), then the values leave — as a guarded load + store per iteration the loop paid one HBM
  // round trip per channel, 16 or 32 in a row (most of the kernel's time on the 56 x 56 layers).
  const int c_blk = blockIdx.y * 64 * NSUB;
  constexpr int IT = 16 * NSUB;
  const int ery = lane >> (q.gshift + 3), ex = lane & (q.slot - 1);
  // (S2: chunk pixel (a, b) of the g grid is x pixel (2a + py, 2b + px))
  const int eyy = S2 ? 2 * (y0 + ery) + py : y0 + ery, exx = S2 ? 2 * ex + px : ex;
  const bool elive = ex < q.W && ery < q.R && y0 + ery < q.H && eyy < q.Hx && exx < q.Wx;
  const unsigned elane = elive ? (unsigned)(eyy * q.Wx + exx) * 4u : 0xFFFFFFF0u;
  [[maybe_unused]] const BufRsrc r_gx = make_rsrc_sized(gx, q.x_bytes);
  const unsigned xplane4 = (unsigned)HW * 4u;
  if constexpr (XP) {
    // the mask bits of this thread's pixel for channels c_blk .. c_blk + 64 NSUB - 1: one dword per 32 channels
    const BufRsrc r_t = make_rsrc_sized(xin, q.p_bytes);
    const unsigned tlane = elive ? (unsigned)(eyy * q.Wx + exx) * 8u : 0xFFFFFFF0u;
    unsigned tw[2 * NSUB];
#pragma unroll
    for (int h = 0; h < 2 * NSUB; ++h) {
      const int ch = c_blk + 32 * h;  // workgroup-uniform
      const bool ok = ch < 64 * q.cw64;
      tw[h] = __float_as_uint(buf_ld(r_t, ok ? tlane : 0xFFFFFFF0u,
                                     ok ? (unsigned)((n * q.cw64 + (ch >> 6)) * HW) * 8u + 4u * ((ch >> 5) & 1) : 0u));
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      [[maybe_unused]] const int c = c_blk + wave + 4 * i;  // wave-uniform
      const float v = stage[(wave + 4 * i) * SROW + lane];
      const int lc = wave + 4 * i;                          // channel within the block: dword lc >> 5, bit lc & 31
      [[maybe_unused]] const float r = ((tw[lc >> 5] >> (lc & 31)) & 1u) ? v : 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), r_gx, (int)(c < q.C ? elane : 0xFFFFFFF0u),
                                            (int)(c < q.C ? (unsigned)(n * q.C + c) * xplane4 : 0u), 0);
#endif
    }
  } else {
    const BufRsrc r_x = make_rsrc_sized(xin, q.x_bytes);
    float xv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = c_blk + wave + 4 * i;  // wave-uniform
      xv[i] = buf_ld(r_x, c < q.C ? elane : 0xFFFFFFF0u, c < q.C ? (unsigned)(n * q.C + c) * xplane4 : 0u);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      [[maybe_unused]] const int c = c_blk + wave + 4 * i;
      const float v = stage[(wave + 4 * i) * SROW + lane];
      [[maybe_unused]] const float r = fabsf(xv[i]) < 1.0f ? v : 0.0f;  // hard-tanh STE (NaN x -> 0, like masked_fill)
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), r_gx, (int)(c < q.C ? elane : 0xFFFFFFF0u),
                                            (int)(c < q.C ? (unsigned)(n * q.C + c) * xplane4 : 0u), 0);
#endif
    }
  }
}

// ------------------------------------------------------------------------------------------------- wgrad
// GEMM: M = 64 output channels o (one 16-row sub-tile per wave), N = 32 input channels x 9 taps (18 sub-tiles of
// 16 columns, all of them in every wave: 72 accumulator registers), K = pixels: the workgroup walks the 64-slot
// chunks of its share of the images (split-K over grid.x; partial sums are added by the caller).
// Per chunk: g (fp16 hi / lo, [o][64 slots]) and sign(x) with one halo row above and below, stored THREE times,
// shifted by kx - 1 pixels, so that every tap's 8-pixel fragment is a 16-byte aligned LDS read.
// XP: sign(x) comes from the bit planes P = x > 0 (`xin`) and M = x < 0 (`xin2`), [N][C/64][Hx][Wx] uint64, instead
// of the fp32 tensor: 2 bits per element of saved state and of traffic.  Same values, same results.
// SI: sign(x) fill items per thread, ceil(16 NC BR groups / 256) — a template parameter so that the fill loops are
// straight-line code (a run-time trip count puts the loaded values behind phi copies, which wait for the loads).
template <int ST, int KS, int NC, bool XP, int SI>
__global__ __launch_bounds__(grad::NT, 2) void wgrad_kernel(const float* __restrict__ g,
                                                            const float* __restrict__ xin,
                                                            const float* __restrict__ xin2,
                                                            float* __restrict__ part, const GradGeo q,
                                                            int imgs_per_split) {
  using namespace grad;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int PD = KS / 2, T = KS * KS, NJ = NC * T;  // NC 16-channel sub-tiles of input channels per workgroup
  const int BR = ST * q.RR + 2 * PD;         // sign(x) rows per chunk (all row slots + halo, zero where no image row)
  const int BROW = q.slot + 8;               // halves per row (16-byte aligned, bank spreading)
  u16* a_hi = reinterpret_cast<u16*>(lds_raw);
  u16* a_mid = a_hi + 64 * AROW;
  u16* a_lo = a_mid + 64 * AROW;
  u16* bsx = a_lo + 64 * AROW;          // [kx][c 0..16 NC-1][row 0..BR-1][BROW]
  const int bplane = 16 * NC * BR * BROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int o0 = blockIdx.z * 64, c0 = blockIdx.y * 16 * NC;
  const int n_begin = blockIdx.x * imgs_per_split;
  const int n_end = min(q.N, n_begin + imgs_per_split);
  const int HW = q.H * q.W, HWx = q.Hx * q.Wx;
  const int groups = 1 << q.gshift;          // 8-pixel groups per row slot

  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses of the two k-steps: k = 32 ks + 8 lg -> (row, column group) of the chunk
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int kk = 32 * ks + 8 * lg;
    const int ry = kk / q.slot, xg = kk - ry * q.slot;
    a_off[ks] = (16 * wave + li) * AROW + kk;
    b_off[ks] = (li * BR + ST * ry) * BROW + xg;  // + (sub * 16 * BR + ky) * BROW + kx * bplane per sub-tile
  }

  // ---- fill roles, once per workgroup.  Every item a thread fills keeps its place in the tile for all chunks: its byte
  // offset inside the (image, first channel, first row) block of the tensor and its LDS address are computed HERE; per
  // chunk a wave-uniform offset moves the block, rows outside the image become out-of-range offsets (which read 0), and
  // the loads are buffer loads with immediate element offsets — no index arithmetic per element in the chunk loop
  // (round 2/3a: 64-bit address + select per element, ~20 VALU instructions per loaded element).
  const BufRsrc r_g = make_rsrc_sized(g, q.g_bytes), r_x = make_rsrc_sized(xin, XP ? q.p_bytes : q.x_bytes);
  [[maybe_unused]] const BufRsrc r_x2 = make_rsrc_sized(XP ? xin2 : xin, XP ? q.p_bytes : q.x_bytes);
  constexpr int XB = XP ? 8 : 4;          // bytes from one pixel to the next in the tensor that stands for x
  constexpr unsigned kOOB = 0xFFFFFF00u;  // +- a few elements stays out of range (make_geo keeps the tensors below it)
  // g: 64 channels x 8 groups of 8 k-slots = 512 items; thread t: group t & 7 of channels t >> 3 and (t >> 3) + 32
  const int g8 = tid & 7, gry = g8 >> q.gshift, gj = g8 & (groups - 1);
  const int g_nv = q.W - 8 * gj;  // elements of the group inside the row (>= 8: all)
  unsigned goff[2];
  int glds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ol = (tid >> 3) + 32 * i;
    const bool fix = o0 + ol < q.O && gry < q.R && g_nv > 0;
    goff[i] = fix ? (unsigned)(ol * HW + gry * q.W + 8 * gj) * 4u : kOOB;
    glds[i] = ol * AROW + 8 * g8;
  }
  // sign(x): (channel, row, 8-pixel group) items, at most four per thread; NE values x[ST*8j - PD .. ] per item
  constexpr int NE = 8 * ST + 2 * PD;
  const int sx_items = 16 * NC * BR * groups;
  int xbase[SI], xlds[SI], xpr[SI], x_nv[SI];
  [[maybe_unused]] int xbit[SI];              // XP: the item's channel as a bit of its 32-channel plane dword
  bool xfirst[SI];
#pragma unroll
  for (int i = 0; i < SI; ++i) {
    const int item = tid + NT * i;
    int j, cl, pr;
    if constexpr (XP) {
      // channel fastest: the 16 NC channels of a pixel are bits of the same one or two plane dwords, so the lanes of a
      // wave ask for 2-4 addresses per load instead of 64 (with the fp32 order — group fastest — every lane of a plane
      // load hit its own 64-byte segment and the texture addresser became the kernel's bound: 617 vs 333 us)
      cl = item % (16 * NC);
      const int rest = item / (16 * NC);
      j = rest & (groups - 1);
      pr = rest >> q.gshift;
    } else {
      j = item & (groups - 1);
      const int t = item >> q.gshift;
      cl = t / BR;
      pr = t - cl * BR;
    }
    const bool fix = item < sx_items && c0 + cl < q.C;
    xpr[i] = fix ? pr : -0x40000000;                          // (a row that is never inside the image)
    if constexpr (XP) {   // planes: [group of 64 channels][pixel] uint64; the channel is bit (c & 31) of dword (c >> 5) & 1
      const int c = c0 + cl;
      xbase[i] = (((c >> 6) - (c0 >> 6)) * HWx + pr * q.Wx + ST * 8 * j) * 8 + 4 * ((c >> 5) & 1);
      xbit[i] = c & 31;
    } else {
      xbase[i] = (cl * HWx + pr * q.Wx + ST * 8 * j) * 4;   // element PD of the item: x = ST*8j
    }
    x_nv[i] = q.Wx - (ST * 8 * j - PD);                       // elements e < x_nv lie left of the row's end
    xfirst[i] = j == 0;                                       // element 0 is x = -1: zero padding
    xlds[i] = item < sx_items ? (cl * BR + pr) * BROW + 8 * j : -1;
  }

  // ---- the chunk loop, software-pipelined: the loads of chunk i + 1 are issued BEFORE the MFMAs of chunk i and land under
  // them (raw values wait in registers: 16 of g, 10 per sign(x) item of the fp32 tensor — the kernel is held to two workgroups per
  // CU by its LDS, so up to 256 registers cost no occupancy); split / sign + LDS writes ("commit") follow the next
  // barrier.  Until round 4 a chunk was load -> wait -> commit -> barrier -> MFMA, and the matrix pipe idled through every
  // HBM round trip (0.22-0.25 of the bf16 peak at three products per MAC).
  float gq[2][8];                                   // raw g values of the chunk in flight
  [[maybe_unused]] float xq[XP ? 1 : SI][XP ? 1 : NE];          // raw x values (fp32 tensor)
  int cn = 0, cy0 = 0;                              // XP: the chunk being committed (its plane loads are not pipelined:
                                                    // 2 x 40 more registers would spill at two waves per SIMD)
  auto issue = [&](int n, int y0, unsigned kill) __attribute__((always_inline)) {   // kill: 0, or kOOB = load nothing
    // g rows y0 .. y0+R-1 of 64 output channels
    const unsigned g_soff = (unsigned)((n * q.O + o0) * HW + y0 * q.W) * 4u;
    const bool g_row = y0 + gry < q.H;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned vo = (g_row ? goff[i] : kOOB) | kill;
#pragma unroll
      for (int e = 0; e < 8; ++e) gq[i][e] = buf_ld(r_g, vo + 4u * e, g_soff);
    }
    if constexpr (!XP) {   // sign(x) rows ST*y0-PD .. of 16 NC input channels
      const unsigned x_soff = (unsigned)((n * q.C + c0) * HWx) * 4u;
      const int yrow = ST * y0 - PD, x_row4 = yrow * q.Wx * XB;
#pragma unroll
      for (int i = 0; i < SI; ++i) {
        const bool rowok = (unsigned)(yrow + xpr[i]) < (unsigned)q.Hx;
        const unsigned vo = (rowok ? (unsigned)(xbase[i] + x_row4) : kOOB) | kill;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          // element e sits (e - PD) pixels from the item's base; the one left of a row's first pixel is padding
          const unsigned ve = e < PD ? (xfirst[i] ? kOOB : vo - (unsigned)XB * (PD - e)) : vo + (unsigned)XB * (e - PD);
          xq[i][e] = buf_ld(r_x, ve, x_soff);
        }
      }
    }
  };
  auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < g_nv ? gq[i][e] : 0.0f;  // (past the row's end the next row was read)
      half8 hi, mid, lo;
      split8(v, hi, mid, lo);
      *reinterpret_cast<half8*>(a_hi + glds[i]) = hi;
      *reinterpret_cast<half8*>(a_mid + glds[i]) = mid;
      *reinterpret_cast<half8*>(a_lo + glds[i]) = lo;
    }
    // KS copies of sign(x): copy kx holds sx[ST*p + kx - PD] at slot p
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      u16 sv[NE];  // sx[ST*8j-PD .. ST*(8j+7)+PD]
      [[maybe_unused]] unsigned vo = 0u, x_soff = 0u;
      if constexpr (XP) {
        x_soff = (unsigned)((cn * q.cw64 + (c0 >> 6)) * HWx) * 8u;
        const int yrow = ST * cy0 - PD;
        vo = (unsigned)(yrow + xpr[i]) < (unsigned)q.Hx ? (unsigned)(xbase[i] + yrow * q.Wx * XB) : kOOB;
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const bool in = e < x_nv[i];
        if constexpr (XP) {
          // (one dword per pixel and plane: 16-byte loads of two pixels measured 5 % faster and returned wrong data —
          // hipcc 7.2 mis-lowers the multi-dword buffer-load builtins, see csrc/bconv_core.h)
          const unsigned ve = e < PD ? (xfirst[i] ? kOOB : vo - (unsigned)XB * (PD - e)) : vo + (unsigned)XB * (e - PD);
          const unsigned pw = __float_as_uint(buf_ld(r_x, ve, x_soff)), mw = __float_as_uint(buf_ld(r_x2, ve, x_soff));
          sv[e] = (in && ((pw >> xbit[i]) & 1u)) ? kBf16One : (in && ((mw >> xbit[i]) & 1u)) ? kBf16MinusOne
                                                                                             : (unsigned short)0;
        } else {
          const float xv = in ? xq[i][e] : 0.0f;
          sv[e] = xv > 0.0f ? kBf16One : xv < 0.0f ? kBf16MinusOne : (unsigned short)0;
        }
      }
      if (xlds[i] >= 0) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          u32x4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (unsigned)sv[ST * (2 * e) + kx] | ((unsigned)sv[ST * (2 * e + 1) + kx] << 16);
          *reinterpret_cast<u32x4*>(bsx + kx * bplane + xlds[i] + 0) = hv;
        }
      }
    }
  };
  // The loads of chunk i + 1 are issued in front of the MFMAs of chunk i and consumed behind them IN THE SAME ITERATION
  // (values that are live around the back edge meet the first chunk's in phi copies, and a copy waits for its load —
  // measured in the ISA: the pipelining was gone).
  const int per_img = (q.H + q.R - 1) / q.R, total = (n_end - n_begin) * per_img;
  int n = n_begin, y0 = 0;       // the chunk whose loads are issued next
  auto advance = [&]() __attribute__((always_inline)) {
    cn = n; cy0 = y0;
    y0 += q.R;
    if (y0 >= q.H) { y0 = 0; ++n; }
  };
  if (total > 0) {
    issue(n, y0, 0u);
    advance();
    commit();
    __syncthreads();
  }
  for (int it = 0; it < total; ++it) {
    // straight-line body: behind the last chunk the loads are all out of range (they return 0 without touching memory)
    // and the commit fills tiles nobody reads — a branch around them puts the loaded registers behind copies again
    issue(n, y0, it + 1 < total ? 0u : kOOB);
    if (it + 1 < total) advance();
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler pulls the first selects of commit() — and their vmcnt wait — up here)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const half8 ah = *reinterpret_cast<const half8*>(a_hi + a_off[ks]);
      const half8 am = *reinterpret_cast<const half8*>(a_mid + a_off[ks]);
      const half8 al = *reinterpret_cast<const half8*>(a_lo + a_off[ks]);
      // B fragments one sub-tile ahead: the LDS round trip of fragment j + 1 runs under the three MFMAs of fragment j
      auto bfrag = [&](int j) __attribute__((always_inline)) {
        const int sub = j / T, tap = j - sub * T, ky = tap / KS, kx = tap - ky * KS;
        return *reinterpret_cast<const half8*>(bsx + kx * bplane + b_off[ks] + (sub * 16 * BR + ky) * BROW);
      };
      half8 b = bfrag(0);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const half8 bn = bfrag(j + 1 < NJ ? j + 1 : j);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, b, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b, acc[j], 0, 0, 0);
        b = bn;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();            // every wave is done with the tiles of chunk `it`
    commit();
    __syncthreads();
  }
  // D layout: column = li (input channel within the half), row = 4 lg + r (output channel within the wave's 16)
  float* dst = part + (size_t)blockIdx.x * q.O * q.C * T;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int sub = j / T, tap = j - sub * T;
    const int c = c0 + 16 * sub + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = o0 + 16 * wave + 4 * lg + r;
      if (o < q.O && c < q.C) dst[((size_t)o * q.C + c) * T + tap] = acc[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------- host side
// dom_w / dom_h: the pixel domain the chunks tile (dgrad: x; wgrad: g).
static bool make_geo(int N, int O, int C, int Hx, int Wx, int st, bool dgrad, GradGeo* q) {
  if (N <= 0 || O <= 0 || C <= 0 || Hx <= 0 || Wx <= 0 || (st != 1 && st != 2)) return false;
  q->N = N; q->O = O; q->C = C; q->Hx = Hx; q->Wx = Wx; q->st = st;
  q->Hg = (Hx - 1) / st + 1;
  q->Wg = (Wx - 1) / st + 1;
  // the pixel domain the chunks tile: x for a stride-1 dgrad, the grid of g for wgrad and for the stride-2 dgrad
  // (whose four parity classes are convolutions over that grid)
  q->H = (dgrad && st == 1) ? Hx : q->Hg;
  q->W = (dgrad && st == 1) ? Wx : q->Wg;
  if (q->W > 64) return false;
  int slot = 8;
  while (slot < q->W) slot *= 2;
  q->slot = slot;
  q->RR = 64 / slot;
  q->R = q->RR < q->H ? q->RR : q->H;
  q->chunks = (q->H + q->R - 1) / q->R;
  q->gshift = 0;
  while ((8 << q->gshift) < slot) ++q->gshift;
  // the kernels address g, x and gx through 32-bit buffer offsets and mark dead elements with offsets >= 0xFFFFFE00
  const unsigned long long gb = (unsigned long long)N * O * q->Hg * q->Wg * 4ull;
  const unsigned long long xb = (unsigned long long)N * C * Hx * Wx * 4ull;
  if (gb > 0xFFFFFE00ull || xb > 0xFFFFFE00ull) return false;
  q->g_bytes = (unsigned)gb;
  q->x_bytes = (unsigned)xb;
  q->cw64 = (C + 63) / 64;
  q->p_bytes = (unsigned)((unsigned long long)N * q->cw64 * Hx * Wx * 8ull);   // <= x_bytes / 4
  return true;
}

static bool ks_ok(int ks, int stride) { return ks == 3 || (ks == 1 && stride == 1); }
static constexpr int kWgradNC1 = 8;  // 1x1 weight gradient: 128 input channels per workgroup

size_t grad_weight_pack_bytes(int O, int C, int ks) {
  return (size_t)((O + 31) / 32) * ks * ks * ((C + 15) / 16) * 64 * sizeof(half8);
}

int launch_grad_pack_weight(const float* what, int O, int C, int ks, void* packed, float* alpha, hipStream_t s) {
  if (!ks_ok(ks, 1)) return BNN_HIP_ERR_UNSUPPORTED;
  const int CS = (C + 15) / 16, OB = (O + 31) / 32, T = ks * ks;
  hipLaunchKernelGGL(grad_alpha_kernel, dim3(O), dim3(64), 0, s, what, C * T, alpha);
  hipLaunchKernelGGL(grad_pack_weight_kernel, dim3(OB * T * CS), dim3(64), 0, s, what, O, C, CS, T,
                     static_cast<half8*>(packed));
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

template <int KS, bool XP>
static int launch_dgrad_t(const float* g, const float* alpha, const void* packed, const float* xin, float* gx,
                          const GradGeo& q, hipStream_t s) {
  using namespace grad;
  const bool s2 = q.st == 2;
  const int halo = s2 ? 1 : 2 * (KS / 2);
  const int PP = (q.R + halo) * (q.W + halo);
  const int nsub = q.C > 64 ? 2 : 1;
  const size_t patch = (size_t)3 * PP * APIX * sizeof(u16);
  const size_t stage = (size_t)64 * nsub * SROW * sizeof(float);
  const size_t lds = patch > stage ? patch : stage;
  const dim3 grid((unsigned)(q.N * q.chunks), (unsigned)((q.C + 64 * nsub - 1) / (64 * nsub)), s2 ? 4u : 1u);
  if constexpr (KS == 3) {
    if (s2) {  // four parity classes (grid.z)
      if (nsub == 2)
        hipLaunchKernelGGL((dgrad_kernel<2, 3, true, XP>), grid, dim3(NT), lds, s, g, alpha, static_cast<const half8*>(packed),
                           xin, gx, q);
      else
        hipLaunchKernelGGL((dgrad_kernel<1, 3, true, XP>), grid, dim3(NT), lds, s, g, alpha, static_cast<const half8*>(packed),
                           xin, gx, q);
      return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
    }
  }
  if (nsub == 2)
    hipLaunchKernelGGL((dgrad_kernel<2, KS, false, XP>), grid, dim3(NT), lds, s, g, alpha, static_cast<const half8*>(packed), xin,
                       gx, q);
  else
    hipLaunchKernelGGL((dgrad_kernel<1, KS, false, XP>), grid, dim3(NT), lds, s, g, alpha, static_cast<const half8*>(packed), xin,
                       gx, q);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_dgrad(const float* g, const float* alpha, const void* packed, const void* xin, int x_planes, float* gx, int N,
                 int O, int C, int H, int W, int ks, int stride, hipStream_t s) {
  GradGeo q;
  if (!ks_ok(ks, stride) || !make_geo(N, O, C, H, W, stride, true, &q)) return BNN_HIP_ERR_UNSUPPORTED;
  const float* xf = static_cast<const float*>(xin);   // (XP: the T plane, read as dwords)
  if (x_planes)
    return ks == 3 ? launch_dgrad_t<3, true>(g, alpha, packed, xf, gx, q, s) : launch_dgrad_t<1, true>(g, alpha, packed, xf, gx, q, s);
  return ks == 3 ? launch_dgrad_t<3, false>(g, alpha, packed, xf, gx, q, s) : launch_dgrad_t<1, false>(g, alpha, packed, xf, gx, q, s);
}

int grad_wgrad_splits(int N, int O, int C, int ks) {
  const int cb = ks == 3 ? 32 : 16 * kWgradNC1;
  const int blocks = ((O + 63) / 64) * ((C + cb - 1) / cb);
  int s = (1024 + blocks - 1) / blocks;  // ~4 workgroups per CU in total
  if (s > N) s = N;
  if (s < 1) s = 1;
  const int per = (N + s - 1) / s;
  return (N + per - 1) / per;
}

template <int ST, int KS, int NC, bool XP, int SI>
static int launch_wgrad_si(const dim3& grid, size_t lds, hipStream_t s, const float* g, const float* x1, const float* x2,
                           float* part, const GradGeo& q, int per) {
  // more than 64 KB of dynamic LDS (82 KB for 7x7 outputs at stride 2, 69 KB for the 56x56 layers) needs the opt-in:
  // per device and per kernel, set on every launch, always to the same constant (the CU's whole LDS)
  if (KS == 3 && hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<ST, KS, NC, XP, SI>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  hipLaunchKernelGGL((wgrad_kernel<ST, KS, NC, XP, SI>), grid, dim3(grad::NT), lds, s, g, x1, x2, part, q, per);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

template <int ST, int KS, int NC, bool XP>
static int launch_wgrad_k(const dim3& grid, size_t lds, hipStream_t s, const float* g, const float* x1, const float* x2,
                          float* part, const GradGeo& q, int per) {
  const int items = 16 * NC * (ST * q.RR + 2 * (KS / 2)) * (q.slot / 8);   // sign(x) fill items of a chunk
  switch ((items + grad::NT - 1) / grad::NT) {
    case 1: return launch_wgrad_si<ST, KS, NC, XP, 1>(grid, lds, s, g, x1, x2, part, q, per);
    case 2: return launch_wgrad_si<ST, KS, NC, XP, 2>(grid, lds, s, g, x1, x2, part, q, per);
    case 3: return launch_wgrad_si<ST, KS, NC, XP, 3>(grid, lds, s, g, x1, x2, part, q, per);
    case 4: return launch_wgrad_si<ST, KS, NC, XP, 4>(grid, lds, s, g, x1, x2, part, q, per);
    default: return BNN_HIP_ERR_UNSUPPORTED;    // (make_geo's slots keep a chunk at <= 1024 items)
  }
}

// x_planes == 0: `xin` is the fp32 input (xin2 unused); else `xin` / `xin2` are its sign planes P / M.
int launch_wgrad(const float* g, const void* xin, const void* xin2, int x_planes, float* part, int splits, int N, int O,
                 int C, int H, int W, int ks, int stride, hipStream_t s) {
  using namespace grad;
  GradGeo q;
  if (!ks_ok(ks, stride) || !make_geo(N, O, C, H, W, stride, false, &q)) return BNN_HIP_ERR_UNSUPPORTED;
  if (splits < 1 || splits > N) return BNN_HIP_ERR_INVALID_ARG;
  const int per = (N + splits - 1) / splits;
  if ((N + per - 1) / per != splits) return BNN_HIP_ERR_INVALID_ARG;
  const int nc = ks == 3 ? 2 : kWgradNC1;
  const size_t lds =
      (size_t)(3 * 64 * AROW + ks * 16 * nc * (stride * q.RR + 2 * (ks / 2)) * (q.slot + 8)) * sizeof(u16);
  const dim3 grid((unsigned)splits, (unsigned)((C + 16 * nc - 1) / (16 * nc)), (unsigned)((O + 63) / 64));
  const float* x1 = static_cast<const float*>(xin);
  const float* x2 = static_cast<const float*>(xin2);
  if (x_planes) {
    if (ks == 1) return launch_wgrad_k<1, 1, kWgradNC1, true>(grid, lds, s, g, x1, x2, part, q, per);
    if (stride == 2) return launch_wgrad_k<2, 3, 2, true>(grid, lds, s, g, x1, x2, part, q, per);
    return launch_wgrad_k<1, 3, 2, true>(grid, lds, s, g, x1, x2, part, q, per);
  }
  if (ks == 1) return launch_wgrad_k<1, 1, kWgradNC1, false>(grid, lds, s, g, x1, x2, part, q, per);
  if (stride == 2) return launch_wgrad_k<2, 3, 2, false>(grid, lds, s, g, x1, x2, part, q, per);
  return launch_wgrad_k<1, 3, 2, false>(grid, lds, s, g, x1, x2, part, q, per);
}

}  // namespace bnn
