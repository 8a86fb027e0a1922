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
//     nothing to D nor to the non-zero count — no per-tap masks in the inner loop.
#include "bconv_core.h"

namespace bnn {

// The 32 output channels of the block are produced in PASSES runs of 32/PASSES channels: fewer
// live accumulators (more waves per SIMD) and 1/PASSES of the unrolled code, looped.
//   MULTI == false: the layer has ONE chunk (C <= 64*CWC/2); its field is loaded once and stays
//                   in registers across all passes.
//   MULTI == true : any number of chunks; each pass walks the chunks and re-loads the field.
//   GSPLIT: the PASSES runs of a block are separate waves (grid is PASSES times larger) instead of a
//           loop: finer work items for the small-image layers, whose few, long waves otherwise
//           quantise badly over the 1024 SIMDs and all reach their HBM epilogue at the same moment.
//   NN: non-negative activations (M plane all zero, BNN_HIP_FLAG_ACT_NONNEG): P-only field.
//   WZ: some weights are exactly zero (BNN_HIP_FLAG_WEIGHT_ZEROS): second scalar stream with the mask `Z`.
//   OBW: consecutive 32-channel blocks one wave produces from ONE load of its field (single-chunk layers: the
//        field load, its padding selects and the non-zero count are ~10 % of a block's instructions).
//   DS: the residual is the block's shortcut convolution, computed here (bconv_core.h, ShortcutArgs) instead of read.
template <int KH, int KW, int CWC, int EP, int MINW, int PASSES, bool MULTI, bool GSPLIT = false, bool NN = false,
          bool WZ = false, int OBW = 1, bool DS = false>
__global__ __launch_bounds__(64, MINW) void bconv_sgpr_kernel(
    const uint32_t* __restrict__ P, const uint32_t* __restrict__ M, const uint32_t* __restrict__ W,
    const uint32_t* __restrict__ Z, BNN_EPI_PARAMS, BNN_DS_PARAMS, const Geo g) {
  static_assert(!(WZ && (NN || GSPLIT)), "the zero-weight variant is two-plane, unsplit");
  static_assert(!DS || (EP == EP_OUT && OBW == 1 && !WZ), "the folded shortcut: conv2-type epilogue only");
  static_assert(OBW == 1 || (!MULTI && !GSPLIT), "several blocks per wave: single-chunk, unsplit kernels only");
  static_assert(!(WZ && EP == EP_MIDT), "the threshold epilogue takes the non-zero count per lane, not per channel");
  constexpr int T = KH * KW;
  constexpr int NW = T * CWC;  // words per (o, chunk)
  constexpr int NACC = kOCB / PASSES;
  BNN_EPI_INIT;
  // XCD-aware work order (1-D grid).  Workgroup b runs on XCD b % 8 (observed placement; only
  // speed depends on it).  Each XCD owns ONE contiguous eighth of the pixel tiles and walks it
  // once per output-channel block: its slice of the packed input (1/8 of a few tens of MB) stays
  // in that XCD's 4 MB L2 across the blocks and across the 3-row halos of neighbouring tiles,
  // instead of every XCD streaming the whole input once per block.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int obp = (int)fast_div((uint32_t)slot, g.m_tpx, g.s_tpx);  // slot / tiles_per_xcd: (block, part) when GSPLIT
  const int ob0 = GSPLIT ? obp / PASSES : obp * OBW;
  const int part = GSPLIT ? obp - ob0 * PASSES : 0;
  const int tile = xcd * g.tiles_per_xcd + (slot - obp * g.tiles_per_xcd);
  if (tile >= g.tiles) return;
  const Pix px = decode_pixel<true>(g, tile * kWave + threadIdx.x);
  [[maybe_unused]] const ShortcutArgs sc{dsP, dsW, ds_alpha, ds_a, ds_b};
  [[maybe_unused]] uint32_t dsr[8];
  [[maybe_unused]] int ds_nz = 0;
  if constexpr (DS) ds_nz = load_shortcut_field(g, px, sc, dsr);
  uint32_t pr[NW], mr[NW];
  int nz = 0;
  if constexpr (!MULTI) {
    if (ob0 * kOCB < g.O) {
      load_field<KH, KW, CWC, NN, true, true>(g, px, 0, P, M, pr, mr);
      nz = count_nonzero<NW, NN>(pr, mr, 0);
    }
  }
  // EP_MIDT on a single chunk: the popcount chains start from kMidtBias - nz / 2 (bconv_core.h: midt2_shift_in)
  constexpr bool MIDT2 = EP == EP_MIDT && !MULTI && !WZ;
  [[maybe_unused]] int midt_seed = 0;
  [[maybe_unused]] unsigned long long midt_odd = 0ull;
  if constexpr (MIDT2) {
    midt_seed = kMidtBias - (nz >> 1);
    midt_odd = __builtin_amdgcn_ballot_w64((nz & 1) != 0);
  }
#pragma unroll 1
  for (int obi = 0; obi < OBW; ++obi) {
  const int ob = ob0 + obi;
  uint32_t pbits = 0u, mbits = 0u;
  if (ob * kOCB < g.O) {
    const uint32_t* wblk = W + (size_t)ob * g.nchunk * (kOCB * NW);
    const uint32_t* zblk = WZ ? Z + (size_t)ob * g.nchunk * (kOCB * NW) : nullptr;
    // the parities of the block's thresholds (word 1 of its first ODD channel: thresholds.hip)
    [[maybe_unused]] uint32_t midt_par = 0u;
    if constexpr (MIDT2) midt_par = (uint32_t)epi.thr[kThrStride * (ob * kOCB + 1) + 1];
    // conv2-type single-chunk kernels (BN + residual + ReLU -> fp32 [+ packed]): ALL shortcut values of the block
    // are requested up front.  gfx950 counts loads and stores in one vmcnt, and once both kinds are pending the
    // compiler has to wait for vmcnt(0): a residual load issued after the previous pass's stores would make the
    // wave wait for those stores' acknowledgements in every pass.  Loads first, stores only after the last load
    // has been consumed: the stores of a pass are never waited for.  Costs 24 VGPRs (8 -> 6 waves per SIMD).
    constexpr bool RES_ALL = !DS && !MULTI && !GSPLIT && PASSES > 1 && (EP == EP_OUT || EP == EP_LAST);
    [[maybe_unused]] float resq[RES_ALL ? kOCB : 1];
    if constexpr (RES_ALL) {
      if ((ob + 1) * kOCB <= g.O) prefetch_residual<kOCB, EP, true>(g, px, ob * kOCB, epi, resq);
      else prefetch_residual<kOCB, EP>(g, px, ob * kOCB, epi, resq);
    }
    // One pass = NACC channels of the block.  `qc` (RES_ALL kernels): which NACC-slice of the shortcut queue is this
    // pass's (the queue moves up after every BNN_RES_UNROLL passes: registers cannot be indexed by the pass number).
    auto one_pass = [&, px](int ps, auto qc) __attribute__((always_inline)) {
      int acc[NACC];
      [[maybe_unused]] int nzacc[NACC];
      float resv[NACC];
      // single-chunk: shortcut values are requested before the popcount loop and land under it.
      // multi-chunk: the field + 32 accumulators already fill the 128-VGPR budget of 4 waves/SIMD;
      // holding NACC more values across the loop spills (39 VGPRs measured), so they are fetched late.
      constexpr bool RES_EARLY = !DS && !RES_ALL && (!MULTI || BNN_MULTI_RES_EARLY);
      // wave-uniform; the same for every pass of a block: a block either builds its sign words by shift-in (straight-
      // line epilogue) or by OR-ing bits into place (guarded epilogue), never both
      const bool fullb = (ob + 1) * kOCB <= g.O;
      if constexpr (RES_ALL) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) resv[j] = resq[decltype(qc)::value * NACC + j];
      }
      if constexpr (RES_EARLY) {
        if (fullb) prefetch_residual<NACC, EP, true>(g, px, ob * kOCB + ps * NACC, epi, resv);
        else prefetch_residual<NACC, EP>(g, px, ob * kOCB + ps * NACC, epi, resv);
      }
      // the folded shortcut values of the pass.  Single-chunk kernels: BEFORE the popcount loop, whose weight stream needs
      // the scalar registers (8 values wait in VGPRs); multi-chunk kernels (16 or 32 values, a register file at its
      // occupancy step): after it
      if constexpr (DS && !MULTI) shortcut_values<NACC>(g, ob * kOCB + ps * NACC, sc, dsr, ds_nz, resv);
      // two-instruction sign test: the pass's comparands as wave-uniform buffer loads (bconv_core.h: midt2_shift_in)
      [[maybe_unused]] int midt_a[NACC];
      if constexpr (MIDT2) {
        // (unconditional: under `if (fullb)` the compiler zero-initialises the eight registers first — eight v_mov and a
        // wait for every earlier load.  The loads of channels past O are safe because the TABLE is padded to whole
        // 32-channel blocks (bnn_hip_sign_thresholds_f32 writes o_pad rows; capi.hip documents the size) and the block
        // is guarded by ob * kOCB < O — not because of the descriptor: the offset travels in `soffset`, which raw
        // buffers do not range-check)
        const BufRsrc rt = make_rsrc_sized(epi.thr, (unsigned)((g.O + kOCB - 1) / kOCB * kOCB) * (unsigned)(kThrStride * 4));
#pragma unroll
        for (int j = 0; j < NACC; ++j)
          midt_a[j] = (int)buf_ld_u32s(rt, (unsigned)(kThrStride * (ob * kOCB + ps * NACC + j) + (NN ? 2 : 3)) * 4u);
      }
      // compile-time profiles with a float epilogue: the counts are kept as the bit pattern of 2^23 + count (epilogue())
      constexpr bool SEEDED = !WZ && NACC % 2 == 0 && EP != EP_RUNTIME && EP != EP_MIDT;
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        acc[j] = SEEDED ? (int)kCountSeed : 0;
        if constexpr (WZ) nzacc[j] = 0;
      }
      if constexpr (MULTI) {
        for (int ch = 0; ch < g.nchunk; ++ch) {
          load_field<KH, KW, CWC, NN, true>(g, px, ch, P, M, pr, mr);
          if (!WZ && (GSPLIT || ps == 0)) nz = count_nonzero<NW, NN>(pr, mr, nz);
          const size_t woff = ((size_t)ch * kOCB + ps * NACC) * NW;
          if constexpr (WZ) stream_weights_wz<NW, NACC>(wblk + woff, zblk + woff, pr, mr, acc, nzacc);
          else stream_weights<NW, NACC, NN, false, EP == EP_MIDT>(wblk + woff, pr, mr, acc);
        }
      } else {
        const size_t woff = (size_t)ps * (NACC * NW);
        if constexpr (WZ) stream_weights_wz<NW, NACC>(wblk + woff, zblk + woff, pr, mr, acc, nzacc);
        else if constexpr (MIDT2) stream_weights<NW, NACC, NN, true, true, BNN_STREAM_ILP, true>(wblk + woff, pr, mr, acc, midt_seed);
        else stream_weights<NW, NACC, NN, true, EP == EP_MIDT>(wblk + woff, pr, mr, acc, SEEDED ? (int)kCountSeed : 0);
      }
      // ONE branch on `fullb` around everything that differs: with the late shortcut fetch and the epilogue under two
      // separate ifs, the guarded side's 32 per-channel predicates are computed at the common dominator — in front of
      // the first if, on every wave — and spilled to VGPR lanes (215 v_readlane + 123 v_writelane in the 512->512 kernel).
      constexpr bool RES_LATE_FETCH = !DS && !RES_EARLY && !RES_ALL;
      const int o0 = ob * kOCB + ps * NACC;
      if constexpr (DS && MULTI) shortcut_values<NACC>(g, o0, sc, dsr, ds_nz, resv);
      [[maybe_unused]] int negnz = NN ? -nz : nz;  // EP_MIDT (see its epilogue): dot = +-2*count + negnz
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (EP == EP_MIDT) asm("" : "+v"(negnz));
#endif
      auto to_dot = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NACC; ++j)  // dot = non-zeros - 2*disagreements = 2*agreements - non-zeros
          acc[j] = WZ                 ? nzacc[j] - 2 * acc[j]
                   : EP == EP_MIDT    ? acc[j]          // the raw count: its epilogue scales and offsets it itself
                   : NN               ? 2 * acc[j] - nz
                                      : nz - 2 * acc[j];
      };
      if constexpr (EP == EP_RUNTIME) {  // (the run-time profile under its 128-register cap spills more that way: as before)
        if constexpr (RES_LATE_FETCH) {
          if (fullb) prefetch_residual<NACC, EP, true>(g, px, o0, epi, resv);
          else prefetch_residual<NACC, EP>(g, px, o0, epi, resv);
        }
        to_dot();
        if (fullb) epilogue<NACC, EP, true>(g, px, o0, acc, resv, epi, pbits, mbits, negnz);
        else epilogue<NACC, EP>(g, px, o0, acc, resv, epi, pbits, mbits, negnz);
      } else if constexpr (MIDT2) {
        // two vector instructions per channel (midt2_shift_in), for EVERY block — no branch on `fullb`: the pass stays one
        // basic block, so the comparands' loads stay in front of the popcount loop (behind a branch the compiler sinks
        // them to their use, and every pass would wait out a memory round trip).  The table covers whole blocks (pad
        // channels: "never"); their bits are cleared at the store (`keep`).
        const uint32_t par = midt_par >> (ps * NACC);
        static_for<NACC>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          pbits = midt2_shift_in<NN, j>(pbits, acc[j], midt_a[j], par, midt_odd);
        });
      } else if (fullb) {
        if constexpr (RES_LATE_FETCH) prefetch_residual<NACC, EP, true>(g, px, o0, epi, resv);
        if constexpr (SEEDED) {
          epilogue<NACC, EP, true, true>(g, px, o0, acc, resv, epi, pbits, mbits, 0, NN ? 2.0f : -2.0f,
                                         NN ? -(float)nz : (float)nz);
        } else {
          to_dot();
          epilogue<NACC, EP, true>(g, px, o0, acc, resv, epi, pbits, mbits, negnz, NN ? 2.0f : -2.0f);
        }
      } else {
        if constexpr (RES_LATE_FETCH) prefetch_residual<NACC, EP>(g, px, o0, epi, resv);
        if constexpr (SEEDED) {
#pragma unroll
          for (int j = 0; j < NACC; ++j) acc[j] -= (int)kCountSeed;
        }
        to_dot();
        epilogue<NACC, EP>(g, px, o0, acc, resv, epi, pbits, mbits, negnz, NN ? 2.0f : -2.0f);
      }
    };
    if constexpr (RES_ALL) {
      constexpr int UNR = BNN_RES_UNROLL;  // passes per iteration of the rolled loop
      static_assert(PASSES % UNR == 0, "whole iterations");
#pragma unroll 1
      for (int pg = 0; pg < PASSES / UNR; ++pg) {
        static_for<UNR>([&](auto uc) __attribute__((always_inline)) { one_pass(pg * UNR + decltype(uc)::value, uc); });
        if constexpr (UNR < PASSES) {  // the rest of the queue moves up
#pragma unroll
          for (int j = 0; j + UNR * NACC < kOCB; ++j) resq[j] = resq[j + UNR * NACC];
        }
      }
    } else {
#pragma unroll 1
      for (int ps = GSPLIT ? part : 0; ps < (GSPLIT ? part + 1 : PASSES); ++ps)
        one_pass(ps, std::integral_constant<int, 0>{});
    }
  }
  const bool rev = MIDT2 || (ob + 1) * kOCB <= g.O;  // == fullb of every pass of this block (MIDT2: always shift-in)
  uint32_t xorw = 0u;  // EP_MIDT: flip bits of this block's channels (blocks past O — zero tail words — have none)
  uint32_t keep = ~0u;  // MIDT2: the channels of the block that exist
  if constexpr (EP == EP_MIDT) {
    if (ob * kOCB < g.O) xorw = (uint32_t)epi.thr[kThrStride * (ob * kOCB) + 1];
    if (MIDT2 && NN) xorw = ~xorw;  // the agreement form shifts in the COMPLEMENT of every bit
    if constexpr (MIDT2) {
      const int left = g.O - ob * kOCB;
      keep = left >= kOCB ? ~0u : left <= 0 ? 0u : ((1u << left) - 1u);
    }
  }
  if constexpr (GSPLIT) store_packed_part<PASSES>(g, px, ob, part, pbits, mbits, epi, rev, xorw, keep);
  else store_packed(g, px, ob, pbits, mbits, epi, rev, xorw, keep);
  }
}

// ---------------------------------------------------------------------------------
// Generic kernel: any KH/KW/stride/pad/dilation/cwc, optional zero-weight mask.
// One lane = one output pixel, one wave = 64 pixels x 32 output channels in 4 passes of 8.
// ---------------------------------------------------------------------------------
constexpr int kOG = 8;

template <bool WZ>
__global__ __launch_bounds__(64) void bconv_generic_kernel(
    const uint32_t* __restrict__ P, const uint32_t* __restrict__ M, const uint32_t* __restrict__ W,
    const uint32_t* __restrict__ Z, BNN_EPI_PARAMS, const Geo g) {
  BNN_EPI_INIT;
  const Pix px = decode_pixel(g, blockIdx.x * kWave + threadIdx.x);
  const int ob = blockIdx.y;
  const int taps = g.KH * g.KW;
  const int per_o = taps * g.cwc;

  int dotv[kOCB];
#pragma unroll
  for (int j = 0; j < kOCB; ++j) dotv[j] = 0;

  if (ob * kOCB < g.O) {
#pragma unroll
    for (int pass = 0; pass < kOCB / kOG; ++pass) {
      const int j0 = pass * kOG;
      int acc[kOG], nzw[kOG];
#pragma unroll
      for (int k = 0; k < kOG; ++k) { acc[k] = 0; nzw[k] = 0; }
      int nz = 0;
      if (ob * kOCB + j0 < g.O) {
        for (int t = 0; t < taps; ++t) {
          const int ky = t / g.KW, kx = t - ky * g.KW;
          const int iy = px.oy * g.sh - g.ph + ky * g.dh;
          const int ix = px.ox * g.sw - g.pw + kx * g.dw;
          const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.Wd;
          const int pix = ok ? iy * g.Wd + ix : 0;
          for (int cw = 0; cw < g.cw32; ++cw) {
            // word cw of the pixel lives in plane (cw >> 1), half (cw & 1)
            const size_t aw =
                ((((size_t)px.n * (g.cw32 >> 1) + (cw >> 1)) * g.H * g.Wd + pix) << 1) + (cw & 1);
            const uint32_t pw = ok ? P[aw] : 0u;
            const uint32_t mw = ok ? M[aw] : 0u;
            if (!WZ) nz += __builtin_popcount(pw | mw);
            const int ch = cw / g.cwc, c = cw - ch * g.cwc;
            const size_t wbase =
                ((size_t)(ob * g.nchunk + ch) * kOCB + j0) * per_o + t * g.cwc + c;
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
      }
#pragma unroll
      for (int k = 0; k < kOG; ++k) dotv[j0 + k] = (WZ ? nzw[k] : nz) - 2 * acc[k];
    }
  }
  uint32_t pbits = 0u, mbits = 0u;
  float resv[kOCB];
  prefetch_residual<kOCB, EP_RUNTIME>(g, px, ob * kOCB, epi, resv);
  epilogue<kOCB, EP_RUNTIME>(g, px, ob * kOCB, dotv, resv, epi, pbits, mbits);
  store_packed(g, px, ob, pbits, mbits, epi);
}

// ---------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------
#ifndef BNN_SGPR_PASSES  // passes per 32-channel block, single-chunk 3x3 layers
#define BNN_SGPR_PASSES 4
#endif
#ifndef BNN_SGPR_PASSES_MULTI  // same, multi-chunk 3x3 layers (each pass re-loads the field)
#define BNN_SGPR_PASSES_MULTI 1
#endif

#ifndef BNN_SINGLE_GSPLIT  // pieces a 32-channel block of a SINGLE-chunk conv1-type 3x3 layer is split into (1 = off)
#define BNN_SINGLE_GSPLIT 1
#endif
#ifndef BNN_SINGLE_GSPLIT_MAX_WAVES  // ...when the unsplit launch has at most this many waves
#define BNN_SINGLE_GSPLIT_MAX_WAVES 8192
#endif
#ifndef BNN_MULTI_GSPLIT  // pieces a 32-channel block of a multi-chunk 3x3 layer is split into (1 = off)
#define BNN_MULTI_GSPLIT 2
#endif
#ifndef BNN_GSPLIT_MAX_WAVES  // ...when the unsplit launch has at most this many waves
#define BNN_GSPLIT_MAX_WAVES 16384
#endif
#ifndef BNN_GSPLIT4_MAX_WAVES  // four pieces instead, when two pieces give at most this many waves (0 = never)
#define BNN_GSPLIT4_MAX_WAVES 8192
#endif

#ifndef BNN_SGPR_OBW  // 32-channel blocks per wave of the single-chunk 3x3 EP_PLAIN kernels (1 = one block per wave)
#define BNN_SGPR_OBW 2
#endif

#ifndef BNN_NN_MULTI_MINW  // waves per SIMD the non-negative multi-chunk kernels are allocated for
#define BNN_NN_MULTI_MINW 1
#endif

// NN: the caller vouches for an all-zero M plane (BNN_HIP_FLAG_ACT_NONNEG); 3x3 kernels only.
// split_ok: the multi-chunk kernels may split a 32-channel block over two waves (lower latency of ONE batch; with
// several batches in flight — BNN_HIP_FLAG_THROUGHPUT — the unsplit kernel's single field load per block wins).
template <int KH, int KW, int CWC, int EP, bool NN>
static void launch_sgpr_t(const ConvP& p, const Geo& g, bool split_ok, hipStream_t s) {
  const dim3 grid((unsigned)(8 * g.tiles_per_xcd) * oblocks(p));  // see the XCD note in the kernel
  constexpr bool k3 = KH * KW > 1;
  constexpr int P1 = k3 ? BNN_SGPR_PASSES : 1, PM = k3 ? BNN_SGPR_PASSES_MULTI : 1;
  if (k3 && p.nchunk == 1) {
#if BNN_SGPR_OBW > 1
    // several blocks per wave, one field load: the drop-in (alpha [, bias] -> fp32) layer only.  128->128 56x56 b256:
    // 0.715 -> 0.73 of the int-ALU peak; with the fused residual-block epilogues the whole net LOSES 2 % with one
    // batch in flight (fewer, longer waves) and is unchanged with two.
    if constexpr (EP == EP_PLAIN) {
      if (oblocks(p) >= BNN_SGPR_OBW) {
        const dim3 grid2((unsigned)(8 * g.tiles_per_xcd) * ((oblocks(p) + BNN_SGPR_OBW - 1) / BNN_SGPR_OBW));
        hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, 1, P1, false, false, NN, false, BNN_SGPR_OBW>), grid2,
                           dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
        return;
      }
    }
#endif
#if BNN_SINGLE_GSPLIT > 1
    // few waves (the stride-2 first convs of a stage at batch 256: 6-12 k waves over ~7 k slots — pure wave
    // quantisation): the passes of a block become separate waves, each with its own load of the field
    if constexpr (EP == EP_MID || EP == EP_MIDT) {
      if (split_ok && (long long)grid.x <= BNN_SINGLE_GSPLIT_MAX_WAVES) {
        const dim3 grid2(grid.x * BNN_SINGLE_GSPLIT);
        hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, 1, BNN_SINGLE_GSPLIT, false, true, NN>), grid2,
                           dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
        return;
      }
    }
#endif
    // conv2-type epilogue on a 128-channel P-only field: 97 VGPRs uncapped, one over the budget of 5 waves per SIMD
    constexpr int MW1 = (NN && CWC == 4 && EP == EP_OUT) ? BNN_OUT4_MINW : 1;
    if constexpr (EP == EP_OUT && NN) {
      if (p.ds_P) {  // the block's shortcut convolution folded in (ds_fold_applies() vouches for the shape)
        hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, MW1, P1, false, false, NN, false, 1, true>), grid,
                           dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
        return;
      }
    }
    hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, MW1, P1, false, false, NN>), grid,
                       dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
    return;
  }
  if constexpr (k3 && CWC == 4) {  // the only shape class with several chunks of a large field
    // multi-chunk 3x3: all 32 accumulators live (one pass over the chunks); with both planes in
    // registers the file is capped for 4 waves per SIMD (512->512 7x7 b256: 100 us vs 131 us
    // uncapped / 2 passes), the P-only field fits without a cap.
    constexpr int MW = NN ? BNN_NN_MULTI_MINW : 4;
#if BNN_MULTI_GSPLIT > 1
    // few pixel tiles per SIMD (ResNet layer3/4 at batch 256: 6 and 3 waves per SIMD): split the block
    // throughput mode keeps the split for launches of fewer than 2048 waves (two per SIMD): those need it even
    // beside another batch (config-5 net at batch 128: 127 k images/s with the split, 119 k without)
    if ((split_ok || grid.x < 2048u) && (long long)grid.x <= BNN_GSPLIT_MAX_WAVES) {
      // conv1-type layers, one batch at a time, still fewer than BNN_GSPLIT4_MAX_WAVES waves when split in two (layer 4
      // at batch 256: 6272 equal waves over 1024 SIMDs = a seventh round for an eighth of them): four pieces, 12.25
      // waves per SIMD.  Measured (round 5): layer4.0.conv1 + layer4.1.conv1 99.4 -> 94.6 us; the conv2-type layers do
      // not gain (64.0 -> 63.6 us) or lose (folded shortcut: 73.3 -> 76.3 us: every piece recomputes the shortcut
      // field) and keep two pieces; with EVERY multi-chunk layer in four the net lost (tools/experiments/README.md 20).
      if constexpr (EP == EP_MID || EP == EP_MIDT) {
        if (split_ok && (long long)grid.x * BNN_MULTI_GSPLIT <= BNN_GSPLIT4_MAX_WAVES) {
          const dim3 grid4(grid.x * 4);
          hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, MW, 4, true, true, NN>), grid4, dim3(kWave), 0, s, p.P,
                             p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
          return;
        }
      }
      const dim3 grid2(grid.x * BNN_MULTI_GSPLIT);
      if constexpr (EP == EP_OUT && NN) {
        if (p.ds_P) {
          hipLaunchKernelGGL(
              (bconv_sgpr_kernel<KH, KW, CWC, EP, MW, BNN_MULTI_GSPLIT, true, true, NN, false, 1, true>), grid2,
              dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
          return;
        }
      }
      hipLaunchKernelGGL(
          (bconv_sgpr_kernel<KH, KW, CWC, EP, MW, BNN_MULTI_GSPLIT, true, true, NN>), grid2,
          dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
      return;
    }
#endif
    if constexpr (EP == EP_OUT && NN) {
      if (p.ds_P) {
        hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, MW, PM, true, false, NN, false, 1, true>), grid,
                           dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
        return;
      }
    }
    hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, MW, PM, true, false, NN>), grid,
                       dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
    return;
  }
  if constexpr (EP == EP_OUT && NN) {
    if (p.ds_P) {
      hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, 1, PM, true, false, NN, false, 1, true>), grid,
                         dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
      return;
    }
  }
  hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, 1, PM, true, false, NN>), grid,
                     dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
}

template <int KH, int KW, int CWC, int EP>
static void launch_sgpr_e(const ConvP& p, const Geo& g, bool nn, bool split_ok, hipStream_t s) {
  if constexpr (KH * KW > 1) {
    if (nn) return launch_sgpr_t<KH, KW, CWC, EP, true>(p, g, split_ok, s);
  }
  launch_sgpr_t<KH, KW, CWC, EP, false>(p, g, split_ok, s);
}

// Zero-weight variant (BNN_HIP_FLAG_WEIGHT_ZEROS): two-plane field, 4 passes of 8 channels, run-time epilogue.
template <int KH, int KW, int CWC, int EP>
static void launch_sgpr_wz(const ConvP& p, const Geo& g, hipStream_t s) {
  const dim3 grid((unsigned)(8 * g.tiles_per_xcd) * oblocks(p));
  if (KH * KW > 1 && p.nchunk == 1)
    hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, 1, 4, false, false, false, true>), grid,
                       dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
  else
    hipLaunchKernelGGL((bconv_sgpr_kernel<KH, KW, CWC, EP, 1, 4, true, false, false, true>), grid,
                       dim3(kWave), 0, s, p.P, p.M, p.W, p.Z, BNN_EPI_ACTUALS, BNN_DS_ACTUALS, g);
}

// PROFILES: whether the compile-time epilogue profiles exist for this shape (3x3 only).
template <int KH, int KW, int CWC, bool PROFILES>
static void launch_sgpr(const ConvP& p, int flags, hipStream_t s) {
  const Geo g = make_geo(p);
  const bool nn = (flags & BNN_HIP_FLAG_ACT_NONNEG) != 0;
  const bool so = (flags & BNN_HIP_FLAG_THROUGHPUT) == 0;
  // (the int32 "raw" output is a run-time switch: EP_PLAIN keeps its counts in float form and cannot serve it)
  const bool fused = (g.flags & (EF_BN | EF_RES | EF_RELU | EF_PRELU | EF_PACK | EF_RAW)) != 0;
  if (flags & BNN_HIP_FLAG_WEIGHT_ZEROS) {
    if (fused) launch_sgpr_wz<KH, KW, CWC, EP_RUNTIME>(p, g, s);
    else launch_sgpr_wz<KH, KW, CWC, EP_PLAIN>(p, g, s);
    return;
  }
  if constexpr (PROFILES) {
    if (g.flags == kFlagsMid && p.thr) return launch_sgpr_e<KH, KW, CWC, EP_MIDT>(p, g, nn, so, s);
    if (g.flags == kFlagsMid) return launch_sgpr_e<KH, KW, CWC, EP_MID>(p, g, nn, so, s);
    if (g.flags == kFlagsOut) return launch_sgpr_e<KH, KW, CWC, EP_OUT>(p, g, nn, so, s);
    if (g.flags == kFlagsOutP) return launch_sgpr_e<KH, KW, CWC, EP_OUTP>(p, g, nn, so, s);
    if (g.flags == kFlagsLast) return launch_sgpr_e<KH, KW, CWC, EP_LAST>(p, g, nn, so, s);
    if (g.flags == kFlagsHb) return launch_sgpr_e<KH, KW, CWC, EP_HB>(p, g, nn, so, s);
    if (g.flags == kFlagsHb3) return launch_sgpr_e<KH, KW, CWC, EP_HB3>(p, g, nn, so, s);
  } else {
    if (g.flags == kFlagsDs) return launch_sgpr_e<KH, KW, CWC, EP_DS>(p, g, nn, so, s);
  }
  if (fused) launch_sgpr_e<KH, KW, CWC, EP_RUNTIME>(p, g, nn, so, s);
  else launch_sgpr_e<KH, KW, CWC, EP_PLAIN>(p, g, nn, so, s);
}

static void launch_generic(const ConvP& p, bool wz, hipStream_t s) {
  const dim3 grid((p.npix + kWave - 1) / kWave, oblocks(p));
  if (wz)
    hipLaunchKernelGGL((bconv_generic_kernel<true>), grid, dim3(kWave), 0, s, p.P, p.M, p.W, p.Z,
                       BNN_EPI_ACTUALS, make_geo(p));
  else
    hipLaunchKernelGGL((bconv_generic_kernel<false>), grid, dim3(kWave), 0, s, p.P, p.M, p.W, p.Z,
                       BNN_EPI_ACTUALS, make_geo(p));
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

// Weight source: the scalar-cache stream into SGPRs.  The alternative north_star names — weight tiles staged in LDS per
// 4-wave workgroup — was built and measured slower on every ResNet-18 shape (512->512 7x7 b256: 216 vs 100 us; a
// vector-broadcast weight path measured 162 us); it lives on as a test-only cross-check in csrc/legacy/bconv_lds.hip
// (libbnn_hip_legacy.so), not in this library.

// Whether launch_bconv() would run this convolution in a kernel that can take the folded shortcut branch (ConvP::ds_*):
// a tiled 3x3 layer on non-negative activations without zero weights, conv2-type epilogue (BN + residual + ReLU ->
// fp32 + sign planes), 64 / 128 / 256 shortcut channels.
bool ds_fold_applies(const ConvP& p, int flags) {
  if ((flags & (BNN_HIP_FLAG_FORCE_GENERIC | BNN_HIP_FLAG_WEIGHT_ZEROS)) ||
      !(flags & BNN_HIP_FLAG_ACT_NONNEG))
    return false;
  if (p.KH != 3 || p.KW != 3 || p.dh != 1 || p.dw != 1 || !small_indices(p) || (p.cwc != 2 && p.cwc != 4)) return false;
  if (p.ds_C != 64 && p.ds_C != 128 && p.ds_C != 256) return false;
  ConvP q = p;
  q.ds_P = reinterpret_cast<const uint32_t*>(&q);  // (any non-null value: only the flag word is looked at)
  return make_geo(q).flags == kFlagsOut;
}

int launch_bconv(const ConvP& p, int flags, hipStream_t s) {
  const bool wz = (flags & BNN_HIP_FLAG_WEIGHT_ZEROS) != 0;
  const bool generic = (flags & BNN_HIP_FLAG_FORCE_GENERIC) || p.dh != 1 || p.dw != 1 || !small_indices(p);
  bool done = false;
  if (!generic) {
    done = true;
#define BNN_PICK(KH_, KW_, C_, PROF_)                           \
  if (p.KH == KH_ && p.KW == KW_ && p.cwc == C_) {              \
    launch_sgpr<KH_, KW_, C_, PROF_>(p, flags, s);              \
  } else
    BNN_PICK(3, 3, 4, true) BNN_PICK(3, 3, 2, true) BNN_PICK(1, 1, 16, false)
    BNN_PICK(1, 1, 8, false) BNN_PICK(1, 1, 4, false) BNN_PICK(1, 1, 2, false) { done = false; }
#undef BNN_PICK
  }
  if (!done) launch_generic(p, wz, s);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
