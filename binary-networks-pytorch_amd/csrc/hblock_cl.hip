// hblock_cl.hip — the hierarchical block in one launch for SMALL images (14 x 14, 7 x 7): lanes are OUTPUT CHANNELS.
//
// csrc/hblock.hip maps a wave's lanes to 64 output pixels.  A 14 x 14 image has 196 pixels — four pixel groups, the
// last one 4 / 64 full — and a 7 x 7 image 49: 23 % of every instruction works on nothing, and a workgroup (one image:
// the planes live in its LDS) cannot borrow pixels from another image.  Here a wave owns 64 output CHANNELS of a tile of
// whole image rows (28 or 21 pixels):
//   * one accumulator per pixel of the tile, in registers;
//   * the lane's weights of the current K-step (9 taps x 4 words = 128 input channels) in registers, loaded from
//     global memory as one 16-byte load per tap (layout [channel group][K-step][tap][lane][4]: 1 KB per instruction);
//   * the input plane's cells broadcast out of LDS (every lane reads the SAME 16 bytes: conflict-free), each cell used
//     for up to nine (tap, output pixel) pairs: v_and_b32 + v_bcnt_u32_b32, both operands in vector registers;
//   * taps that fall into the zero padding LEFT and RIGHT of the image are not computed at all (the loops over cells and
//     taps are unrolled, the column of a cell is a constant), rows above / below the image are zero cells;
//   * sign bits leave as ballots: one v_cmp per pixel gives the 64 channels' bits of that pixel — the plane word itself.
// Exact pixel counts, so the instruction count per image is 0.91 (14 x 14) / 0.82 .. 0.9 (7 x 7) of the nine-tap count.
// Same integers and float operations as csrc/hblock.hip: bit-identical results (tests/test_gpu_hblock.py).
#include <cstring>

#include "bconv_core.h"

namespace bnn {

struct ClPhase {
  int O, ncg;         // output channels, groups of 64
  int cw, nks;        // words per cell of its input plane, K-steps of KW words
  int c_off;          // first channel of its slice of the block's output
  unsigned w_off;     // words: its weights, [cg][K-step][tap][lane][KW]
  unsigned a_off, pa_off, pb_off;  // floats
  unsigned lds_in;    // bytes: its input plane [(H + 2) rows][W][cw words]; row 0 and row H + 1 stay zero
  unsigned lds_nz;    // bytes: non-zero inputs under every pixel's window, int32 [(H + 2)][(W + 2)]
};

struct ClGeo {
  int N, C;           // images, block width
  int ng_in;          // 64-channel groups of the block's input planes
  ClPhase ph[3];
  unsigned lds_out;   // next block's planes [H * W cells][C / 32 words]
  unsigned lds_done;  // completion counters [2][tiles]
  unsigned lds16;
  unsigned na_off, nb_off;
  unsigned f32_bytes;
  unsigned lds_sc;    // DSW > 0: the shortcut's sign planes [H * W pixels][P: DSW words | M: DSW words], then int32 [H * W]:
  unsigned lds_scnz;  //          non-zero inputs of every pixel
};

// DSW > 0 (csrc/hblock.hip: HbDs): the block's shortcut convolution — binary 1 x 1 over another binarisation of the block's
// input — computed here instead of read as a residual tensor.
struct ClDs {
  const uint64_t* P;
  const uint64_t* M;
  const uint32_t* W;   // [C][DSW]
  const float* A;      // [C]
};

namespace {

#ifndef CL_MAX_WAVES  // waves per workgroup the kernel is register-allocated for (12: three per SIMD, 168 registers; 16 spills)
#define CL_MAX_WAVES 12
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char cl_smem[];

__device__ __forceinline__ uint32_t cl_uniform(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

// acc += popcount(w[j] & a[j]), j < KW: one asm statement per tap (hipcc puts an s_nop behind every asm statement)
template <int KW>
__device__ __forceinline__ void cl_tap(int& acc, const uint32_t (&w)[KW], const uint32_t (&a)[KW]) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t t0, t1;
  if constexpr (KW == 4) {
    asm("v_and_b32 %1, %3, %7\n\tv_and_b32 %2, %4, %8\n\tv_bcnt_u32_b32 %0, %1, %0\n\tv_and_b32 %1, %5, %9\n\t"
        "v_bcnt_u32_b32 %0, %2, %0\n\tv_and_b32 %2, %6, %10\n\tv_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0"
        : "+v"(acc), "=&v"(t0), "=&v"(t1)
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
  } else {
    asm("v_and_b32 %1, %3, %5\n\tv_and_b32 %2, %4, %6\n\tv_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0"
        : "+v"(acc), "=&v"(t0), "=&v"(t1)
        : "v"(w[0]), "v"(w[1]), "v"(a[0]), "v"(a[1]));
  }
#else
  for (int j = 0; j < KW; ++j) acc += __builtin_popcount(w[j] & a[j]);
#endif
}

// Two taps of one cell (two accumulators) in one statement: the two popcount chains alternate, so that no instruction
// waits for the one in front of it (with two waves per SIMD a single chain is latency-bound: 7 cycles per instruction measured).
__device__ __forceinline__ void cl_tap2(int& accA, int& accB, const uint32_t (&wA)[4], const uint32_t (&wB)[4],
                                        const uint32_t (&a)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t t0, t1, t2, t3;
  asm("v_and_b32 %2, %6, %14\n\tv_and_b32 %3, %10, %14\n\tv_and_b32 %4, %7, %15\n\tv_and_b32 %5, %11, %15\n\t"
      "v_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %1, %3, %1\n\tv_and_b32 %2, %8, %16\n\tv_and_b32 %3, %12, %16\n\t"
      "v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %5, %1\n\tv_and_b32 %4, %9, %17\n\tv_and_b32 %5, %13, %17\n\t"
      "v_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %1, %3, %1\n\tv_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %5, %1"
      : "+v"(accA), "+v"(accB), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(wA[0]), "v"(wA[1]), "v"(wA[2]), "v"(wA[3]), "v"(wB[0]), "v"(wB[1]), "v"(wB[2]), "v"(wB[3]), "v"(a[0]), "v"(a[1]),
        "v"(a[2]), "v"(a[3]));
#else
  for (int j = 0; j < 4; ++j) {
    accA += __builtin_popcount(wA[j] & a[j]);
    accB += __builtin_popcount(wB[j] & a[j]);
  }
#endif
}

// The k-th tap (dy * 3 + dx) of input cell (cy, cx) that has an output pixel inside an R x W tile, or -1.
constexpr int cl_valid_tap(int cy, int cx, int R, int W, int k) {
  for (int t = 0; t < 9; ++t) {
    const int oy = cy - t / 3, ox = cx - t % 3 + 1;
    if (oy < 0 || oy >= R || ox < 0 || ox >= W) continue;
    if (k-- == 0) return t;
  }
  return -1;
}

// lo[lane P] = low half, hi[lane P] = high half of a ballot.  The ballot is a VALU compare writing a scalar pair, and an
// inline-asm v_writelane that reads it right behind gets no wait states from the compiler (it does not know what the asm
// reads the register for): without the s_nop the lanes received stale words whenever nothing else sat between the two.
template <int P>
__device__ __forceinline__ void cl_writelane2(uint32_t& lo, uint32_t& hi, unsigned long long m) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 3\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
               : "+v"(lo), "+v"(hi)
               : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "n"(P));
#else
  (void)lo; (void)hi; (void)m;
#endif
}

template <int KW>
struct ClVec;
template <>
struct ClVec<4> { using type = uint4; };
template <>
struct ClVec<2> { using type = uint2; };

// One unit: 64 output channels (group cg) x the R image rows from r0 on, of convolution K of the block.
//   W: image width (= height: 14 or 7);  R: rows of the tile;  KW: words per K-step
template <int W, int R, int KW, int K, bool NEXT, int DSW = 0>
__device__ __forceinline__ void cl_unit(const uint32_t* __restrict__ Wt, const float* __restrict__ Kc,
                                        const float* __restrict__ res, float* __restrict__ out, const ClDs& ds, const ClGeo& g,
                                        unsigned char* smem, int n, int cg, int r0, int lane,
                                        typename ClVec<KW>::type (&wnext)[9], int cg_following) {
  constexpr int H = W, T = R * W, HW = H * W;
  constexpr bool LAST = K == 2;
  using V = typename ClVec<KW>::type;
  const ClPhase& ph = g.ph[K];
  const int c = cg * 64 + lane;   // this lane's output channel of the convolution
  const int co = ph.c_off + c;    // ... of the block
  // per-lane constants
  const float alpha = Kc[ph.a_off + c];
  [[maybe_unused]] float pa = 0.0f, pb = 0.0f, na = 0.0f, nb = 0.0f;
  if constexpr (!LAST) {
    pa = Kc[ph.pa_off + c];
    pb = Kc[ph.pb_off + c];
  }
  if constexpr (NEXT) {
    na = Kc[g.na_off + co];
    nb = Kc[g.nb_off + co];
  }
  // the tile's shortcut values: lane = channel, one load per pixel, all requested before the popcount loop
  const BufRsrc rres = make_rsrc_sized(res, g.f32_bytes), rout = make_rsrc_sized(out, g.f32_bytes);
  const unsigned voff = (unsigned)(((n * g.C + co) * HW + r0 * W) * 4);
  // (W = 14: the 14 pixels of a lane's channel in this row are 56 contiguous, 8-byte aligned bytes: seven 8-byte loads /
  // stores — half the instructions and of the partial-line writes L2 has to merge; 7 x 7 images: 196-byte planes, dwords)
  constexpr bool VEC2 = W == 14;
  float resv[T];
  [[maybe_unused]] uint32_t wsc[DSW > 0 ? DSW : 1];
  [[maybe_unused]] float asc = 0.0f;
  // DSW > 0: the shortcut value of (pixel, this lane's channel) is computed in the epilogue, where the K-step's weight
  // registers are free; this lane's 1 x 1 weights travel under the popcount loop where the registers allow it (W = 14: 142
  // of 168; the 7 x 7 form is at 168 without them and fetches them in front of the epilogue)
  constexpr bool SC_EARLY = DSW > 0 && W == 14;
  auto load_sc = [&]() {
    if constexpr (DSW > 0) {
      const uint4* wp = reinterpret_cast<const uint4*>(ds.W + (size_t)co * DSW);
#pragma unroll
      for (int q = 0; q < DSW / 4; ++q) {
        const uint4 v = wp[q];
        wsc[4 * q] = v.x; wsc[4 * q + 1] = v.y; wsc[4 * q + 2] = v.z; wsc[4 * q + 3] = v.w;
      }
      asc = ds.A[co];
    }
  };
  if constexpr (DSW > 0) {
    if constexpr (SC_EARLY) load_sc();
  } else if constexpr (VEC2) {
#pragma unroll
    for (int p = 0; p < T; p += 2) {
      const float2 v2 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(res) + voff + (unsigned)(p * 4));
      resv[p] = v2.x; resv[p + 1] = v2.y;
    }
  } else {
#pragma unroll
    for (int p = 0; p < T; ++p) resv[p] = buf_ld(rres, voff + (unsigned)(p * 4), 0u);
  }
  // non-zero inputs under the windows of the tile's pixels (lane = pixel here)
  const int* nzmap = reinterpret_cast<const int*>(smem + ph.lds_nz);
  int nzv = 0;
  if (lane < T) nzv = nzmap[(r0 + 1 + lane / W) * (W + 2) + lane % W + 1];
  int acc[T];
#pragma unroll
  for (int p = 0; p < T; ++p) acc[p] = 0;
  const uint32_t* plane = reinterpret_cast<const uint32_t*>(smem + ph.lds_in);
  // `wnext` holds the unit's first K-step already (requested by the previous unit of this wave, or by cl_phase): a unit
  // is 1-6 us long and a weight fetch from L2 ~1 us, so it must not start with one
  const V* wq = reinterpret_cast<const V*>(Wt + ph.w_off) + (size_t)cg * ph.nks * (9 * 64) + lane;
  const V* wq_following = reinterpret_cast<const V*>(Wt + ph.w_off) + (size_t)max(cg_following, 0) * ph.nks * (9 * 64) + lane;
#pragma unroll 1
  for (int s = 0; s < ph.nks; ++s) {
    uint32_t w[9][KW];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const uint32_t* e = reinterpret_cast<const uint32_t*>(&wnext[t]);
#pragma unroll
      for (int j = 0; j < KW; ++j) w[t][j] = e[j];
    }
    if (s + 1 < ph.nks) {   // the next K-step's weights travel while this one is consumed
#pragma unroll
      for (int t = 0; t < 9; ++t) wnext[t] = wq[((size_t)(s + 1) * 9 + t) * 64];
    } else if (cg_following >= 0) {   // ... or the first K-step of this wave's next unit of the same convolution
#pragma unroll
      for (int t = 0; t < 9; ++t) wnext[t] = wq_following[t * 64];
    }
    // plane row r0 + cy holds image row r0 + cy - 1 (row 0 / row H + 1: zeros)
    const uint32_t* prow = plane + (size_t)(r0 * W) * ph.cw + s * KW;
    // the cells of the window, in order, read AHEAD of their use: the broadcast read of cell i + kAhead is issued before
    // cell i is consumed (left to itself hipcc issues every ds_read right in front of its use and waits out the whole LDS
    // latency per cell: one register quad, `s_waitcnt lgkmcnt(0)` behind each of the 84 reads of a K-step)
    constexpr int NCELL = (R + 2) * W, kAhead = 2;
    V ring[kAhead + 1];
    static_for<kAhead>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      ring[i] = *reinterpret_cast<const V*>(prow + (size_t)i * ph.cw);
    });
    static_for<NCELL>([&](auto cc) {
      constexpr int ci = decltype(cc)::value, cy = ci / W, cx = ci % W;
      if constexpr (ci + kAhead < NCELL)
        ring[(ci + kAhead) % (kAhead + 1)] = *reinterpret_cast<const V*>(prow + (size_t)(ci + kAhead) * ph.cw);
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
      uint32_t a[KW];
      {
        const uint32_t* e = reinterpret_cast<const uint32_t*>(&ring[ci % (kAhead + 1)]);
#pragma unroll
        for (int j = 0; j < KW; ++j) a[j] = e[j];
      }
      // (cy, cx are constants: the valid taps of the cell are known at compile time)
      static_for<5>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int tA = cl_valid_tap(cy, cx, R, W, 2 * q), tB = cl_valid_tap(cy, cx, R, W, 2 * q + 1);
        if constexpr (tA >= 0 && tB >= 0) {
          if constexpr (KW == 4) {
            cl_tap2(acc[(cy - tA / 3) * W + cx - tA % 3 + 1], acc[(cy - tB / 3) * W + cx - tB % 3 + 1], w[tA], w[tB], a);
          } else {
            cl_tap<KW>(acc[(cy - tA / 3) * W + cx - tA % 3 + 1], w[tA], a);
            cl_tap<KW>(acc[(cy - tB / 3) * W + cx - tB % 3 + 1], w[tB], a);
          }
        } else if constexpr (tA >= 0) {
          cl_tap<KW>(acc[(cy - tA / 3) * W + cx - tA % 3 + 1], w[tA], a);
        }
      });
    });
  }
  if constexpr (DSW > 0 && !SC_EARLY) load_sc();
  // epilogue, pixel by pixel (lane = channel): the float operations of csrc/hblock.hip
  uint32_t ilo = 0u, ihi = 0u, nlo = 0u, nhi = 0u;   // lane p: the 64 sign bits of pixel p (internal / next block)
  const unsigned ovoff = voff;
  static_for<T>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    const int nz = __builtin_amdgcn_readlane(nzv, p);
#else
    const int nz = nzv;
#endif
    const float dot = (float)(2 * acc[p] - nz);
    const float ov = __builtin_fmaf(alpha, dot, 0.0f);
    if constexpr (!LAST) {
      [[maybe_unused]] const float v = __builtin_fmaf(ov, pa, pb);
#if defined(__HIP_DEVICE_COMPILE__)
      const unsigned long long m = __builtin_amdgcn_ballot_w64(is_pos(v));
      cl_writelane2<p>(ilo, ihi, m);
#endif
    }
    if constexpr (DSW > 0) {
      // dot = 2 * (agreeing non-zero inputs) - (non-zero inputs) over the pixel's two plane words (wave-uniform: broadcast
      // reads), value = fmaf(alpha, dot, 0) as bnn_hip_bconv2d rounds it
      const uint32_t* scp = reinterpret_cast<const uint32_t*>(smem + g.lds_sc) + (size_t)(r0 * W + p) * (2 * DSW);
      int agree = 0;
#pragma unroll
      for (int q = 0; q < DSW / 4; ++q) {
        const uint4 pv = *reinterpret_cast<const uint4*>(scp + 4 * q);
        const uint4 mv = *reinterpret_cast<const uint4*>(scp + DSW + 4 * q);
        agree += __builtin_popcount((pv.x & wsc[4 * q]) | (mv.x & ~wsc[4 * q]));
        agree += __builtin_popcount((pv.y & wsc[4 * q + 1]) | (mv.y & ~wsc[4 * q + 1]));
        agree += __builtin_popcount((pv.z & wsc[4 * q + 2]) | (mv.z & ~wsc[4 * q + 2]));
        agree += __builtin_popcount((pv.w & wsc[4 * q + 3]) | (mv.w & ~wsc[4 * q + 3]));
      }
      const int scnz = reinterpret_cast<const int*>(smem + g.lds_scnz)[r0 * W + p];
      resv[p] = __builtin_fmaf(asc, (float)(2 * agree - scnz), 0.0f);
    }
    const float y = ov + resv[p];
    if constexpr (VEC2) {
      resv[p] = y;   // (kept: stored two pixels at a time)
      if constexpr (p % 2 == 1)
        *reinterpret_cast<float2*>(reinterpret_cast<char*>(out) + ovoff + (unsigned)((p - 1) * 4)) = float2{resv[p - 1], resv[p]};
    } else {
      buf_st(rout, ovoff + (unsigned)(p * 4), 0u, y);
    }
    if constexpr (NEXT) {
      [[maybe_unused]] const float v = __builtin_fmaf(y, na, nb);
#if defined(__HIP_DEVICE_COMPILE__)
      const unsigned long long m = __builtin_amdgcn_ballot_w64(is_pos(v));
      cl_writelane2<p>(nlo, nhi, m);
#endif
    }
  });
  if (lane < T) {
    const int oy = lane / W, ox = lane % W;
    if constexpr (!LAST) {
      const ClPhase& pn = g.ph[K < 2 ? K + 1 : 2];
      uint32_t* pl = reinterpret_cast<uint32_t*>(smem + pn.lds_in);
      *reinterpret_cast<uint2*>(pl + (size_t)((r0 + 1 + oy) * W + ox) * pn.cw + 2 * cg) = uint2{ilo, ihi};
      // this group's share of the non-zero counts of the nine windows the pixel lies in
      const int cnt = __builtin_popcount(ilo) + __builtin_popcount(ihi);
      int* nzn = reinterpret_cast<int*>(smem + pn.lds_nz) + (r0 + oy) * (W + 2) + ox;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        __hip_atomic_fetch_add(nzn + (t / 3) * (W + 2) + t % 3, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if constexpr (NEXT) {
      uint32_t* po = reinterpret_cast<uint32_t*>(smem + g.lds_out);
      *reinterpret_cast<uint2*>(po + (size_t)((r0 + oy) * W + ox) * (g.C >> 5) + 2 * (co >> 6)) = uint2{nlo, nhi};
    }
  }
}

// tiles of a W x W image: 14 x 14 -> fourteen single rows; 7 x 7 -> rows {0,1} {2,3} {4,5} {6}.  Small tiles: 14 (or 7)
// accumulators per lane keep the kernel under 128 registers — sixteen waves per workgroup, four per SIMD — and give a
// convolution of 64 output channels fourteen (four) units instead of seven (two).
template <int W>
struct ClTiles;
template <>
struct ClTiles<14> {
  static constexpr int N = 14;
};
template <>
struct ClTiles<7> {
  static constexpr int N = 4;
};

template <int W, int KW, int K, bool NEXT, int DSW>
__device__ __forceinline__ void cl_unit_of_tile(const uint32_t* __restrict__ Wt, const float* __restrict__ Kc,
                                                const float* __restrict__ res, float* __restrict__ out, const ClDs& ds,
                                                const ClGeo& g, unsigned char* smem, int n, int cg, int tile, int lane,
                                                typename ClVec<KW>::type (&wnext)[9], int cg_following) {
  if constexpr (W == 14) {
    cl_unit<14, 1, KW, K, NEXT, DSW>(Wt, Kc, res, out, ds, g, smem, n, cg, tile, lane, wnext, cg_following);
  } else {
    if (tile < 3) cl_unit<7, 2, KW, K, NEXT, DSW>(Wt, Kc, res, out, ds, g, smem, n, cg, 2 * tile, lane, wnext, cg_following);
    else cl_unit<7, 1, KW, K, NEXT, DSW>(Wt, Kc, res, out, ds, g, smem, n, cg, 6, lane, wnext, cg_following);
  }
}

// Convolution K: units (tile, channel group) from a ticket counter; a unit waits for the tiles of convolution K - 1 that
// hold the rows under its windows (all their channel groups) — no barrier between the convolutions (csrc/hblock.hip).
template <int W, int KW, int K, bool NEXT, int DSW>
__device__ __forceinline__ void cl_phase(const uint32_t* __restrict__ Wt, const float* __restrict__ Kc,
                                         const float* __restrict__ res, float* __restrict__ out, const ClDs& ds, const ClGeo& g,
                                         unsigned char* smem, int n, int lane) {
  constexpr int NT = ClTiles<W>::N;
  const ClPhase& ph = g.ph[K];
  uint32_t* ctl = reinterpret_cast<uint32_t*>(smem);
  uint32_t* done = reinterpret_cast<uint32_t*>(smem + g.lds_done);   // [2][NT]
  const uint32_t nunits = (uint32_t)(NT * ph.ncg);
  // (units handed out statically, wave w takes w, w + waves, ...: they are equally long, and the dependencies only point
  // to smaller unit numbers of the previous convolution, which the same waves finished earlier or other waves hold)
  const uint32_t wave = cl_uniform(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  (void)ctl;
  using V = typename ClVec<KW>::type;
  V wnext[9];
  if (wave < nunits) {   // the first K-step of this wave's first unit (before it waits for its inputs)
    const int cg0 = (int)wave % ph.ncg;
    const V* wq = reinterpret_cast<const V*>(Wt + ph.w_off) + (size_t)cg0 * ph.nks * (9 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t) wnext[t] = wq[t * 64];
  }
  for (uint32_t u = wave; u < nunits; u += nwaves) {
    const int tile = (int)u / ph.ncg, cg = (int)u - tile * ph.ncg;
    const int cg_following = u + nwaves < nunits ? (int)((u + nwaves) % (uint32_t)ph.ncg) : -1;
    if constexpr (K > 0) {
      const uint32_t want = (uint32_t)g.ph[K - 1].ncg;
      const uint32_t* dp = done + (K - 1) * NT;
      const int lo = max(tile - 1, 0), hi = min(tile + 1, NT - 1);
      for ([[maybe_unused]] unsigned idle = 0;; ++idle) {
        bool missing = false;
        if (lane <= hi - lo)
          missing = __hip_atomic_load(&dp[lo + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want;
#if defined(__HIP_DEVICE_COMPILE__)
        if (__builtin_amdgcn_ballot_w64(missing) == 0ull) break;
        __builtin_amdgcn_s_sleep(4);
        if (idle > (1u << 24)) __builtin_trap();
#else
        (void)missing;
        break;
#endif
      }
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
    }
#if !defined(CL_DBG_STAGE) || CL_DBG_STAGE != 4
    cl_unit_of_tile<W, KW, K, NEXT, DSW>(Wt, Kc, res, out, ds, g, smem, n, cg, tile, lane, wnext, cg_following);
#endif
#if defined(CL_DBG_STAGE) && CL_DBG_STAGE == 3
    break;
#endif
    if constexpr (K < 2) {
      if (lane == 0) __hip_atomic_fetch_add(&done[K * NT + tile], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// One workgroup = one image.  KW3: words per K-step of conv3 (2 when its input has 64 channels).
template <int W, int KW3, bool NEXT, int DSW = 0>
__global__ __launch_bounds__(CL_MAX_WAVES * 64) void hblock_cl_kernel(const uint64_t* __restrict__ inP, const uint32_t* __restrict__ Wt,
                                                        const float* __restrict__ Kc, const float* __restrict__ res,
                                                        float* __restrict__ out, uint64_t* __restrict__ outP,
                                                        const ClGeo g, const ClDs ds) {
  constexpr int H = W, HW = H * W;
  unsigned char* smem = cl_smem;
  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int n = blockIdx.x;
  {
    uint4* z = reinterpret_cast<uint4*>(smem);
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (unsigned i = tid; i < g.lds16; i += nthr) z[i] = zero;
  }
  __syncthreads();
  {  // the block's input planes -> [row + 1][col][cw words]; non-zero counts of the nine windows each cell lies in
    const ClPhase& p0 = g.ph[0];
    uint32_t* pl = reinterpret_cast<uint32_t*>(smem + p0.lds_in);
    int* nz0 = reinterpret_cast<int*>(smem + p0.lds_nz);
    for (int i = tid; i < g.ng_in * HW; i += nthr) {
      const int gq = i / HW, r = i - gq * HW;
      const uint64_t v = inP[((size_t)n * g.ng_in + gq) * HW + r];
      *reinterpret_cast<uint2*>(pl + (size_t)(W + r) * p0.cw + 2 * gq) = uint2{(uint32_t)v, (uint32_t)(v >> 32)};
      const int cnt = __builtin_popcountll(v);
      int* q = nz0 + (r / W) * (W + 2) + r % W;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        __hip_atomic_fetch_add(q + (t / 3) * (W + 2) + t % 3, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if constexpr (DSW > 0) {   // the shortcut's planes, pixel-major, and the non-zero inputs of every pixel
    uint32_t* sp = reinterpret_cast<uint32_t*>(smem + g.lds_sc);
    int* snz = reinterpret_cast<int*>(smem + g.lds_scnz);
    for (int i = tid; i < (DSW / 2) * HW; i += nthr) {
      const int gq = i / HW, r = i - gq * HW;
      const uint64_t pv = ds.P[((size_t)n * (DSW / 2) + gq) * HW + r], mv = ds.M[((size_t)n * (DSW / 2) + gq) * HW + r];
      *reinterpret_cast<uint2*>(sp + (size_t)r * (2 * DSW) + 2 * gq) = uint2{(uint32_t)pv, (uint32_t)(pv >> 32)};
      *reinterpret_cast<uint2*>(sp + (size_t)r * (2 * DSW) + DSW + 2 * gq) = uint2{(uint32_t)mv, (uint32_t)(mv >> 32)};
      __hip_atomic_fetch_add(snz + r, __builtin_popcountll(pv | mv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
#if defined(CL_DBG_STAGE) && CL_DBG_STAGE == 0
  return;
#endif
  cl_phase<W, 4, 0, NEXT, DSW>(Wt, Kc, res, out, ds, g, smem, n, lane);
#if defined(CL_DBG_STAGE) && (CL_DBG_STAGE == 1 || CL_DBG_STAGE == 3 || CL_DBG_STAGE == 4)
  return;
#endif
  cl_phase<W, 4, 1, NEXT, DSW>(Wt, Kc, res, out, ds, g, smem, n, lane);
#if defined(CL_DBG_STAGE) && CL_DBG_STAGE == 2
  return;
#endif
  cl_phase<W, KW3, 2, NEXT, DSW>(Wt, Kc, res, out, ds, g, smem, n, lane);
  if constexpr (NEXT) {
    __syncthreads();
    const int ngo = g.C >> 6;
    const uint2* po = reinterpret_cast<const uint2*>(smem + g.lds_out);
    for (int i = tid; i < ngo * HW; i += nthr) {
      const int gq = i / HW, r = i - gq * HW;
      const uint2 v = po[(size_t)r * ngo + gq];
      outP[((size_t)n * ngo + gq) * HW + r] = (uint64_t)v.x | ((uint64_t)v.y << 32);
    }
  }
}

// standard packed weights -> [cg][K-step][tap][lane][KW]
__global__ __launch_bounds__(256) void hblock_cl_pack_weight_kernel(const uint32_t* __restrict__ src,
                                                                    uint32_t* __restrict__ dst, int O, int cw, int kw,
                                                                    int cw_s, int cwc_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O * 9 * cw) return;
  const int cwi = i % cw, tap = (i / cw) % 9, o = i / (9 * cw);
  const int ob = o >> 5, j = o & 31, nchunk_s = cw_s / cwc_s;
  const uint32_t v = src[((size_t)(ob * nchunk_s + cwi / cwc_s) * 32 + j) * 9 * cwc_s + tap * cwc_s + cwi % cwc_s];
  const int nks = cw / kw;
  dst[((((size_t)(o >> 6) * nks + cwi / kw) * 9 + tap) * 64 + (o & 63)) * kw + cwi % kw] = v;
}

struct ClShape {
  int cin[3], O[3], cw[3], kw[3];
};

bool cl_shape(int C_in, int planes, ClShape& s) {
  if (C_in <= 0 || C_in > 4096 || planes < 256 || planes % 256 != 0 || planes > 4096) return false;
  s.cin[0] = C_in; s.cin[1] = planes / 2; s.cin[2] = planes / 4;
  s.O[0] = planes / 2; s.O[1] = planes / 4; s.O[2] = planes / 4;
  for (int k = 0; k < 3; ++k) {
    if (s.cin[k] % 64 != 0) return false;
    s.cw[k] = s.cin[k] / 32;
    s.kw[k] = (k == 2 && s.cw[k] == 2) ? 2 : 4;
    if (s.cw[k] % s.kw[k] != 0) return false;
  }
  return true;   // (every convolution has a multiple of 64 output channels: planes % 256 == 0)
}

long long cl_lds(const ClShape& s, int W, int planes, bool next, ClGeo* g, int dsw = 0) {
  const int H = W;
  long long off = 16;
  const int nt = W == 14 ? 14 : 4;
  if (g) g->lds_done = (unsigned)off;
  off += (2 * nt * 4 + 15) / 16 * 16;
  for (int k = 0; k < 3; ++k) {
    if (g) g->ph[k].lds_in = (unsigned)off;
    off += ((long long)(H + 2) * W * s.cw[k] * 4 + 15) / 16 * 16;
    if (g) g->ph[k].lds_nz = (unsigned)off;
    off += ((long long)(H + 2) * (W + 2) * 4 + 15) / 16 * 16;
  }
  if (g) g->lds_out = (unsigned)off;
  if (next) off += ((long long)H * W * (planes / 32) * 4 + 15) / 16 * 16;
  if (dsw > 0) {
    if (g) g->lds_sc = (unsigned)off;
    off += ((long long)H * W * 2 * dsw * 4 + 15) / 16 * 16;
    if (g) g->lds_scnz = (unsigned)off;
    off += ((long long)H * W * 4 + 15) / 16 * 16;
  }
  return off;
}

}  // namespace

bool hblock_cl_supported(const bnn_hip_hblock_desc* d) {
  ClShape s;
  if (!cl_shape(d->C_in, d->planes, s)) return false;
  if (!((d->H == 14 && d->W == 14) || (d->H == 7 && d->W == 7))) return false;
  if (d->rows_per_band > 0 && d->rows_per_band != d->H) return false;
  if (d->images_per_band > 1) return false;
  return cl_lds(s, d->W, d->planes, true, nullptr) <= 64 * 1024;
}

int hblock_cl_weight_words(int C_in, int planes, long long off[3]) {
  ClShape s;
  if (!cl_shape(C_in, planes, s)) return -1;
  long long w = 0;
  for (int k = 0; k < 3; ++k) {
    off[k] = w;
    w += (long long)s.O[k] * 9 * s.cw[k];
  }
  return (int)w;
}

int launch_hblock_cl_pack_weights(int C_in, int planes, const uint32_t* const w[3], uint32_t* dst, hipStream_t st) {
  ClShape s;
  long long off[3];
  if (!cl_shape(C_in, planes, s) || hblock_cl_weight_words(C_in, planes, off) < 0) return BNN_HIP_ERR_UNSUPPORTED;
  for (int k = 0; k < 3; ++k) {
    const int cw_s = 2 * ((s.cin[k] + 63) / 64), cwc_s = choose_cwc(cw_s, 3, 3);
    const int n = s.O[k] * 9 * s.cw[k];
    hipLaunchKernelGGL(hblock_cl_pack_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w[k], dst + off[k], s.O[k],
                       s.cw[k], s.kw[k], cw_s, cwc_s);
  }
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

namespace {
int cl_dsw(const bnn_hip_hblock_desc* d) { return d->C_in / 32; }
int cl_run(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* res,
           float* out, uint64_t* outP, hipStream_t stream, const ClDs* dsp);
}  // namespace

bool hblock_cl_ds_supported(const bnn_hip_hblock_desc* d) {
  ClShape s;
  if (!hblock_cl_supported(d) || !cl_shape(d->C_in, d->planes, s) || d->planes != 2 * d->C_in) return false;
  if (d->C_in != 128 && d->C_in != 256) return false;
  return cl_lds(s, d->W, d->planes, true, nullptr, cl_dsw(d)) <= 64 * 1024;
}

int launch_hblock_cl(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc,
                     const float* res, float* out, uint64_t* outP, hipStream_t stream) {
  return cl_run(d, inP, W, Kc, res, out, outP, stream, nullptr);
}

// ... with the shortcut convolution inside (csrc/hblock.hip: launch_hblock_ds)
int launch_hblock_cl_ds(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc,
                        const uint64_t* dsP, const uint64_t* dsM, const uint32_t* dsW, const float* dsA, float* out,
                        uint64_t* outP, hipStream_t stream) {
  if (!hblock_cl_ds_supported(d) || outP == nullptr) return BNN_HIP_ERR_UNSUPPORTED;
  const ClDs ds{dsP, dsM, dsW, dsA};
  return cl_run(d, inP, W, Kc, nullptr, out, outP, stream, &ds);
}

namespace {
int cl_run(const bnn_hip_hblock_desc* d, const uint64_t* inP, const uint32_t* W, const float* Kc, const float* res,
           float* out, uint64_t* outP, hipStream_t stream, const ClDs* dsp) {
  ClShape s;
  bnn_hip_hblock_layout L;
  long long woff[3];
  if (!hblock_cl_supported(d) || !cl_shape(d->C_in, d->planes, s) || hblock_layout(d->C_in, d->planes, &L) != BNN_HIP_OK ||
      hblock_cl_weight_words(d->C_in, d->planes, woff) < 0)
    return BNN_HIP_ERR_UNSUPPORTED;
  ClGeo g;
  std::memset(&g, 0, sizeof(g));
  g.N = d->N;
  g.C = d->planes;
  g.ng_in = d->C_in / 64;
  const bool next = outP != nullptr;
  const int dsw = dsp ? cl_dsw(d) : 0;
  g.lds16 = (unsigned)((cl_lds(s, d->W, d->planes, next, &g, dsw) + 15) / 16);
  int c_off = 0;
  for (int k = 0; k < 3; ++k) {
    ClPhase& p = g.ph[k];
    p.O = s.O[k];
    p.ncg = s.O[k] / 64;
    p.cw = s.cw[k];
    p.nks = s.cw[k] / s.kw[k];
    p.c_off = c_off;
    c_off += s.O[k];
    p.w_off = (unsigned)woff[k];
    p.a_off = (unsigned)L.alpha_off[k];
    p.pa_off = k < 2 ? (unsigned)L.pack_a_off[k] : 0u;
    p.pb_off = k < 2 ? (unsigned)L.pack_b_off[k] : 0u;
  }
  g.na_off = (unsigned)L.next_a_off;
  g.nb_off = (unsigned)L.next_b_off;
  g.f32_bytes = (unsigned)((long long)d->N * d->planes * d->H * d->W * 4);
  const int waves = d->waves > 0 ? std::min(d->waves, CL_MAX_WAVES) : CL_MAX_WAVES;
  const size_t lds = (size_t)g.lds16 * 16;
  const ClDs none{nullptr, nullptr, nullptr, nullptr};
#define CL_LAUNCH_DS(W_, KW3_, DSW_)                                                                                    \
  hipLaunchKernelGGL((hblock_cl_kernel<W_, KW3_, true, DSW_>), dim3((unsigned)d->N), dim3((unsigned)waves * kWave), lds,    \
                     stream, inP, W, Kc, res, out, outP, g, *dsp)
  if (dsp != nullptr) {   // planes == 2 * C_in: 128 -> 256 (conv3 reads 64 channels: KW3 = 2), 256 -> 512 (KW3 = 4)
    if (d->W == 14) { if (dsw == 4) CL_LAUNCH_DS(14, 2, 4); else CL_LAUNCH_DS(14, 4, 8); }
    else { if (dsw == 4) CL_LAUNCH_DS(7, 2, 4); else CL_LAUNCH_DS(7, 4, 8); }
    return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  }
#undef CL_LAUNCH_DS
#define CL_LAUNCH(W_, KW3_, NEXT_)                                                                                      \
  hipLaunchKernelGGL((hblock_cl_kernel<W_, KW3_, NEXT_>), dim3((unsigned)d->N), dim3((unsigned)waves * kWave), lds, stream, \
                     inP, W, Kc, res, out, outP, g, none)
  if (d->W == 14) {
    if (s.kw[2] == 2) { if (next) CL_LAUNCH(14, 2, true); else CL_LAUNCH(14, 2, false); }
    else { if (next) CL_LAUNCH(14, 4, true); else CL_LAUNCH(14, 4, false); }
  } else {
    if (s.kw[2] == 2) { if (next) CL_LAUNCH(7, 2, true); else CL_LAUNCH(7, 2, false); }
    else { if (next) CL_LAUNCH(7, 4, true); else CL_LAUNCH(7, 4, false); }
  }
#undef CL_LAUNCH
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}
}  // namespace

}  // namespace bnn
