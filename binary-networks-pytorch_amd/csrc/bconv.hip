// bconv.hip — binary (ternary-activation x binary-weight) convolution on the gfx950 integer ALU.
//
// Replaces the arithmetic of bnn/layers/conv.py:90-97 (Conv2d.forward):
//     post( conv2d( sign(x), sign(W)*alpha, bias, stride, zero-pad ), x )
// evaluated exactly in integers:
//     D   = sum over taps/words of popcount( (W & M) | (~W & P) )   -> v_bitop3_b32 + v_bcnt_u32_b32
//     dot = nzc_window - 2*D
//     out = fmaf(alpha[o], (float)dot, bias[o]) [* post_scale[o]]
//
// Work decomposition of the tiled kernel (the hot one):
//   * lane  = one output pixel (flattened over N,Ho,Wo) -> NCHW stores are coalesced
//   * wave  = 64 pixels x 32 output channels (one weight block `ob`)
//   * the pixel's whole receptive field for one chunk of input channels
//     (taps x cwc words x 2 planes, <= 72 VGPRs) stays in registers and is reused
//     for all 32 output channels
//   * weights are wave-uniform: they stream through the scalar cache into SGPRs
//     (s_load_dwordx16) and feed v_bitop3_b32 directly as its scalar operand, so
//     the vector ALU executes nothing but bitop3 + bcnt in the main loop
//   * zero padding / ragged edges: out-of-image taps load P = M = 0, which contribute
//     nothing to D nor to nzc — no per-tap masks in the inner loop.
#include "bnn_dev.h"

namespace bnn {

// Geometry only (pointers travel as separate __restrict__ kernel arguments so that the
// compiler may keep wave-uniform reads on the scalar path).
struct Geo {
  int N, H, Wd, Ho, Wo, O;
  int KH, KW, sh, sw, ph, pw, dh, dw;
  int cw32, cwc, nchunk;
  int npix;
  int has_bias, has_scale;
};

template <bool RAW>
__device__ __forceinline__ void store_result(void* __restrict__ out, size_t idx, int dot, float a,
                                             float b, float sc, bool has_scale) {
  if (RAW) {
    static_cast<int32_t*>(out)[idx] = dot;
  } else {
    float v = fmaf(a, (float)dot, b);
    if (has_scale) v *= sc;
    static_cast<float*>(out)[idx] = v;
  }
}

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

// ---------------------------------------------------------------------------------
// Tiled kernel: KH x KW taps, CWC words per chunk, dilation 1, no zero weights.
// ---------------------------------------------------------------------------------
template <int KH, int KW, int CWC, bool RAW>
__global__ __launch_bounds__(64) void bconv_tiled_kernel(
    const uint32_t* __restrict__ P, const uint32_t* __restrict__ M,
    const uint16_t* __restrict__ nzc, const uint32_t* __restrict__ W,
    const float* __restrict__ alpha, const float* __restrict__ bias,
    const float* __restrict__ scale, void* __restrict__ out, const Geo g) {
  constexpr int T = KH * KW;
  constexpr int NW = T * CWC;             // words per (o, chunk)
  constexpr int LV = CWC >= 4 ? 4 : CWC;  // words per vector load

  int q = blockIdx.x * kWave + threadIdx.x;
  const bool live = q < g.npix;
  if (!live) q = g.npix - 1;
  const int hw = g.Ho * g.Wo;
  const int n = q / hw;
  const int r = q - n * hw;
  const int oy = r / g.Wo;
  const int ox = r - oy * g.Wo;
  const int ob = blockIdx.y;

  // receptive field: pixel index per tap (or -1 when the tap falls into the zero padding)
  int off[T];
  int nz = 0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int iy = oy * g.sh - g.ph + t / KW;
    const int ix = ox * g.sw - g.pw + t % KW;
    const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.Wd;
    const int pix = (n * g.H + iy) * g.Wd + ix;
    off[t] = ok ? pix : -1;
    nz += ok ? (int)nzc[ok ? pix : 0] : 0;
  }

  int acc[kOCB];
#pragma unroll
  for (int j = 0; j < kOCB; ++j) acc[j] = 0;

  const uint32_t* wblk = W + (size_t)ob * g.nchunk * (kOCB * NW);

  for (int ch = 0; ch < g.nchunk; ++ch) {
    uint32_t pr[NW], mr[NW];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const bool ok = off[t] >= 0;
      const size_t base = (size_t)(ok ? off[t] : 0) * g.cw32 + (size_t)ch * CWC;
#pragma unroll
      for (int c = 0; c < CWC; c += LV) {
        uint32_t pv[LV], mv[LV];
        load_words<LV>(P, base + c, pv);
        load_words<LV>(M, base + c, mv);
#pragma unroll
        for (int e = 0; e < LV; ++e) {
          pr[t * CWC + c + e] = ok ? pv[e] : 0u;
          mr[t * CWC + c + e] = ok ? mv[e] : 0u;
        }
      }
    }
    const uint32_t* wch = wblk + (size_t)ch * (kOCB * NW);
#pragma unroll
    for (int j = 0; j < kOCB; ++j) {
      const uint32_t* w = wch + j * NW;  // wave-uniform -> scalar loads
      int a0 = 0, a1 = 0;
#pragma unroll
      for (int i = 0; i + 1 < NW; i += 2) {
        a0 += __builtin_popcount(disagree(w[i], mr[i], pr[i]));
        a1 += __builtin_popcount(disagree(w[i + 1], mr[i + 1], pr[i + 1]));
      }
      if (NW & 1) a0 += __builtin_popcount(disagree(w[NW - 1], mr[NW - 1], pr[NW - 1]));
      acc[j] += a0 + a1;
    }
  }

  if (!live) return;
  const size_t obase = ((size_t)n * g.O) * hw + r;
  const int o0 = ob * kOCB;
  if (o0 + kOCB <= g.O) {  // full block: no per-channel bounds checks
#pragma unroll
    for (int j = 0; j < kOCB; ++j) {
      const int o = o0 + j;
      const float a = RAW ? 0.f : alpha[o];
      const float b = (!RAW && g.has_bias) ? bias[o] : 0.f;
      const float sc = (!RAW && g.has_scale) ? scale[o] : 1.f;
      store_result<RAW>(out, obase + (size_t)o * hw, nz - 2 * acc[j], a, b, sc, g.has_scale);
    }
  } else {
#pragma unroll
    for (int j = 0; j < kOCB; ++j) {
      const int o = o0 + j;
      if (o < g.O) {
        const float a = RAW ? 0.f : alpha[o];
        const float b = (!RAW && g.has_bias) ? bias[o] : 0.f;
        const float sc = (!RAW && g.has_scale) ? scale[o] : 1.f;
        store_result<RAW>(out, obase + (size_t)o * hw, nz - 2 * acc[j], a, b, sc, g.has_scale);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Generic kernel: any KH/KW/stride/pad/dilation/cwc, optional zero-weight mask.
// One lane = one output pixel, one wave = 64 pixels x OG output channels.
// ---------------------------------------------------------------------------------
constexpr int kOG = 8;

template <bool WZ, bool RAW>
__global__ __launch_bounds__(64) void bconv_generic_kernel(
    const uint32_t* __restrict__ P, const uint32_t* __restrict__ M,
    const uint16_t* __restrict__ nzc, const uint32_t* __restrict__ W,
    const uint32_t* __restrict__ Z, const float* __restrict__ alpha,
    const float* __restrict__ bias, const float* __restrict__ scale, void* __restrict__ out,
    const Geo g) {
  int q = blockIdx.x * kWave + threadIdx.x;
  const bool live = q < g.npix;
  if (!live) q = g.npix - 1;
  const int hw = g.Ho * g.Wo;
  const int n = q / hw;
  const int r = q - n * hw;
  const int oy = r / g.Wo;
  const int ox = r - oy * g.Wo;
  const int o0 = blockIdx.y * kOG;
  const int ob = o0 / kOCB, j0 = o0 % kOCB;
  const int taps = g.KH * g.KW;
  const int per_o = taps * g.cwc;

  int acc[kOG], nzw[kOG];
#pragma unroll
  for (int k = 0; k < kOG; ++k) { acc[k] = 0; nzw[k] = 0; }
  int nz = 0;

  for (int t = 0; t < taps; ++t) {
    const int ky = t / g.KW, kx = t - ky * g.KW;
    const int iy = oy * g.sh - g.ph + ky * g.dh;
    const int ix = ox * g.sw - g.pw + kx * g.dw;
    const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.Wd;
    const int pix = ok ? (n * g.H + iy) * g.Wd + ix : 0;
    if (!WZ) nz += ok ? (int)nzc[pix] : 0;
    for (int cw = 0; cw < g.cw32; ++cw) {
      const uint32_t pw = ok ? P[(size_t)pix * g.cw32 + cw] : 0u;
      const uint32_t mw = ok ? M[(size_t)pix * g.cw32 + cw] : 0u;
      const int ch = cw / g.cwc, c = cw - ch * g.cwc;
      const size_t wbase = ((size_t)(ob * g.nchunk + ch) * kOCB + j0) * per_o + t * g.cwc + c;
#pragma unroll
      for (int k = 0; k < kOG; ++k) {
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
  if (!live) return;
  const size_t obase = ((size_t)n * g.O) * hw + r;
#pragma unroll
  for (int k = 0; k < kOG; ++k) {
    const int o = o0 + k;
    if (o < g.O) {
      const float a = RAW ? 0.f : alpha[o];
      const float b = (!RAW && g.has_bias) ? bias[o] : 0.f;
      const float sc = (!RAW && g.has_scale) ? scale[o] : 1.f;
      store_result<RAW>(out, obase + (size_t)o * hw, (WZ ? nzw[k] : nz) - 2 * acc[k], a, b, sc,
                        g.has_scale);
    }
  }
}

// ---------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------
static Geo make_geo(const ConvP& p) {
  Geo g;
  g.N = p.N; g.H = p.H; g.Wd = p.Wd; g.Ho = p.Ho; g.Wo = p.Wo; g.O = p.O;
  g.KH = p.KH; g.KW = p.KW; g.sh = p.sh; g.sw = p.sw; g.ph = p.ph; g.pw = p.pw;
  g.dh = p.dh; g.dw = p.dw; g.cw32 = p.cw32; g.cwc = p.cwc; g.nchunk = p.nchunk;
  g.npix = p.npix;
  g.has_bias = p.bias != nullptr;
  g.has_scale = p.scale != nullptr;
  return g;
}

template <int KH, int KW, int CWC>
static void launch_tiled(const ConvP& p, bool raw, hipStream_t s) {
  const dim3 grid((p.npix + kWave - 1) / kWave, (p.O + kOCB - 1) / kOCB);
  const Geo g = make_geo(p);
  if (raw)
    hipLaunchKernelGGL((bconv_tiled_kernel<KH, KW, CWC, true>), grid, dim3(kWave), 0, s, p.P, p.M,
                       p.nzc, p.W, p.alpha, p.bias, p.scale, p.out, g);
  else
    hipLaunchKernelGGL((bconv_tiled_kernel<KH, KW, CWC, false>), grid, dim3(kWave), 0, s, p.P, p.M,
                       p.nzc, p.W, p.alpha, p.bias, p.scale, p.out, g);
}

template <bool WZ, bool RAW>
static void launch_generic_t(const ConvP& p, hipStream_t s) {
  const dim3 grid((p.npix + kWave - 1) / kWave, (p.O + kOG - 1) / kOG);
  hipLaunchKernelGGL((bconv_generic_kernel<WZ, RAW>), grid, dim3(kWave), 0, s, p.P, p.M, p.nzc, p.W,
                     p.Z, p.alpha, p.bias, p.scale, p.out, make_geo(p));
}

static void launch_generic(const ConvP& p, bool wz, bool raw, hipStream_t s) {
  if (wz) {
    if (raw) launch_generic_t<true, true>(p, s);
    else launch_generic_t<true, false>(p, s);
  } else {
    if (raw) launch_generic_t<false, true>(p, s);
    else launch_generic_t<false, false>(p, s);
  }
}

// Chunk width is a pure function of the weight geometry (shared with pack_weight).
int choose_cwc(int cw32, int KH, int KW) {
  if (KH == 3 && KW == 3) return (cw32 % 4 == 0) ? 4 : 2;
  if (KH == 1 && KW == 1) {
    if (cw32 % 16 == 0) return 16;
    if (cw32 % 8 == 0) return 8;
    if (cw32 % 4 == 0) return 4;
    return 2;
  }
  return 2;
}

int launch_bconv(const ConvP& p, int flags, bool raw, hipStream_t s) {
  const bool wz = (flags & BNN_HIP_FLAG_WEIGHT_ZEROS) != 0;
  const bool generic = (flags & BNN_HIP_FLAG_FORCE_GENERIC) || wz || p.dh != 1 || p.dw != 1;
  bool done = false;
  if (!generic) {
    done = true;
    if (p.KH == 3 && p.KW == 3 && p.cwc == 4) launch_tiled<3, 3, 4>(p, raw, s);
    else if (p.KH == 3 && p.KW == 3 && p.cwc == 2) launch_tiled<3, 3, 2>(p, raw, s);
    else if (p.KH == 1 && p.KW == 1 && p.cwc == 16) launch_tiled<1, 1, 16>(p, raw, s);
    else if (p.KH == 1 && p.KW == 1 && p.cwc == 8) launch_tiled<1, 1, 8>(p, raw, s);
    else if (p.KH == 1 && p.KW == 1 && p.cwc == 4) launch_tiled<1, 1, 4>(p, raw, s);
    else if (p.KH == 1 && p.KW == 1 && p.cwc == 2) launch_tiled<1, 1, 2>(p, raw, s);
    else done = false;
  }
  if (!done) launch_generic(p, wz, raw, s);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
