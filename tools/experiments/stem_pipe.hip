// stem_pipe.hip — the fused stem of stem_split.hip (same arithmetic, same tile geometry, same LDS layout) with the
// phases of consecutive tiles SOFTWARE-PIPELINED inside every wave:
//
//   barrier A | phase 1: MFMAs of tile i  ||  max-pool + stores of tile i-1  ||  sign words of tile i-2
//             |          ||  global loads of tile i+1's patch                      (ONE basic block)
//   barrier B | phase 2: BN + ReLU of tile i -> staged conv tile;  patch of tile i+1 -> LDS (fp16 hi / lo)
//
// In stem_split.hip the phases of a workgroup run one after the other (ablations: matrix 153 + pooling 94 + fetch 54
// + skeleton 20 us of 346) and only a second, independent workgroup on the CU overlaps them.  Here the VALU / LDS /
// VMEM instructions of the pooling and the fetch sit in the issue shadow of the MFMAs (16 cycles each) of the SAME wave.
// That needs phase 1 to be free of control flow, so every guarded access is a buffer access whose dead lanes carry an
// out-of-range offset (loads return 0 = the zero padding, stores are dropped), and a null output is a descriptor
// with zero records.
#include "bnn_dev.h"

namespace bnn {

namespace stem3 {
constexpr int CIN = 3, KS = 7, COUT = 64;
constexpr int KROWS = 24, KSTEPS = KROWS / 4;
constexpr int PTH = 4, PTW = 8;
constexpr int NT = 256, SUBS = 5;
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1;
constexpr int MPIX = CTH * CTW;
constexpr int ITH = 2 * CTH + 5;
constexpr int ITWP = 2 * CTW + 6;
constexpr int ROWH = 96;
constexpr int ICHP = ITH * ROWH;
constexpr int NINP = CIN * ICHP;
constexpr int NROW = CIN * ITH;
constexpr int NPC = ITWP / 2;
constexpr int SC = COUT + 4;
constexpr int NW = NT / 64, MG = NW / 2;
constexpr int PJ = COUT / (NW * 8);
constexpr int RSTEP = NT / NPC;
constexpr int PER_T = (NROW + RSTEP - 1) / RSTEP;
constexpr int TT = 2;
constexpr int OFF_HI = 0;
constexpr int OFF_LO = OFF_HI + ((NINP * 2 + 15) / 16) * 16;
constexpr int OFF_STAGE = OFF_LO + ((NINP * 2 + 15) / 16) * 16;
constexpr int OFF_BITS = OFF_STAGE + (MPIX + 1) * SC * 4;  // sign bytes: [2][32 pixels][8]
constexpr int OFF_DUMMY = OFF_BITS + 2 * PTH * PTW * 8;    // 256 bytes nobody reads (non-leader lanes' byte writes)
constexpr int OFF_S5 = OFF_DUMMY + NT;                     // last sub-tile of each pixel group, staged apart: [MG*16][SC]
constexpr int OFF_BN = OFF_S5 + MG * 16 * SC * 4;          // folded BN constants [2][COUT]
constexpr int LDS_BYTES = OFF_BN + 2 * COUT * 4;
constexpr int SMAIN = SUBS - 1;                            // sub-tiles whose accumulators live across barrier B
constexpr unsigned OOB = 0xFFFFFFF0u;                      // beyond every descriptor's num_records
constexpr int POOL_STEPS = PJ * (CTH + PTH);               // micro-steps of one tile's pooling
constexpr int GROUPS = KSTEPS * (SMAIN / 2);               // MFMA groups the pooling is spread over
}  // namespace stem3

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

#if defined(__HIP_DEVICE_COMPILE__)
using SRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ SRsrc srsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ float sld(SRsrc r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void sst(SRsrc r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}
__device__ __forceinline__ void sst2(SRsrc r, unsigned off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, 0);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else  // the host pass of hipcc only parses these
struct SRsrc {};
__device__ __forceinline__ SRsrc srsrc(const void*, unsigned) { return {}; }
__device__ __forceinline__ float sld(SRsrc, unsigned) { return 0.0f; }
__device__ __forceinline__ void sst(SRsrc, unsigned, float) {}
__device__ __forceinline__ void sst2(SRsrc, unsigned, u32x2) {}
__device__ __forceinline__ void lds_barrier() {}
#endif

struct TileAt {  // wave-uniform
  int n, py0, px0;
  bool valid;
};

template <bool HALF>
__global__ __launch_bounds__(stem3::NT, 2) void stem_pipe_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int N, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
    int tiles_x, int per_xcd, float* __restrict__ out, uint64_t* __restrict__ P, uint64_t* __restrict__ M,
    unsigned x_bytes, unsigned out_bytes, unsigned plane_bytes) {
  using namespace stem3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* hiP = reinterpret_cast<_Float16*>(lds_raw + OFF_HI);
  _Float16* loP = reinterpret_cast<_Float16*>(lds_raw + OFF_LO);
  float* stage = reinterpret_cast<float*>(lds_raw + OFF_STAGE);
  float* stage5 = reinterpret_cast<float*>(lds_raw + OFF_S5);
  float* bnc = reinterpret_cast<float*>(lds_raw + OFF_BN);
  uint8_t* bits = lds_raw + OFF_BITS;
  static_assert(SUBS % 2 == 1 && SMAIN % 2 == 0, "one odd sub-tile goes first, the rest in pairs");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int mg = wave % MG, nh = wave / MG;

  const SRsrc rx = srsrc(x, x_bytes), ro = srsrc(out, out_bytes), rP = srsrc(P, plane_bytes), rM = srsrc(M, plane_bytes);

  // ---- once: B fragments (hi, lo) of this wave's 2 channel tiles x 6 k-steps, in registers.
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
    int m = (SUBS * mg + i) * 16 + li;
    if (m >= MPIX) m = MPIX - 1;
    const int cy = m / CTW, cx = m - cy * CTW;
    abase[i] = 2 * cy * ROWH + 2 * cx;
  }
  if (tid < COUT) {  // BN constants live in LDS: 4 VGPRs less across the matrix phase
    bnc[tid] = bn_a[tid];
    bnc[COUT + tid] = bn_b[tid];
  }
  const int fpc = tid % NPC, frow0 = tid / NPC;
  const bool fetcher = tid < NPC * RSTEP;
  const int pchl = lane & 7, pplx = lane >> 3;
  const int pch0 = wave * 8 + pchl;

  const int ntiles = N * tiles_y * tiles_x;
  const int nseq = per_xcd * 8;
  auto tile_of = [&](int seq) { return (seq & 7) * per_xcd + (seq >> 3); };
  auto locate = [&](int seq) {
    TileAt t;
    const int tile = seq < nseq ? tile_of(seq) : ntiles;
    t.valid = tile < ntiles;
    const int tl = t.valid ? tile : 0;
    t.n = tl / (tiles_y * tiles_x);
    const int tr = tl - t.n * tiles_y * tiles_x;
    const int ty = tr / tiles_x;
    t.py0 = ty * PTH;
    t.px0 = (tr - ty * tiles_x) * PTW;
    return t;
  };

  // ---- patch fetch: every load is a buffer load; elements outside the image (zero padding), past the patch and of
  // a non-existent tile carry an out-of-range offset and read as 0.
  float nx0[PER_T], nx1[PER_T];
  auto fetch = [&](const TileAt& t) {
    const int iy0 = 4 * t.py0 - 5, ix0 = 4 * t.px0 - 5;  // 2 * (2 * p0 - 1) - 3
    const int ix = ix0 + 2 * fpc;
    // (plain & and | below, and "offset | dead" instead of "live ? offset : OOB": with && / ?: hipcc builds scalar
    //  branches and exec-mask regions around the address arithmetic, which cuts phase 1 into many basic blocks)
    const bool okc0 = t.valid & ((unsigned)ix < (unsigned)W);
    const bool okc1 = t.valid & (2 * fpc + 1 < ITWP - 1) & ((unsigned)(ix + 1) < (unsigned)W);  // last column: zero (kx = 7)
    const int base = (t.n * CIN * H + iy0) * W + ix;  // element index of x[n][0][iy0][ix]; only used where valid
    int fr = frow0;
    asm volatile("" : "+v"(fr));  // per-row offsets are recomputed per tile: as loop invariants they only get spilled
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int R = fr + RSTEP * u;
      const int c = (R >= 2 * ITH) + (R >= ITH), r = R - c * ITH;
      const bool okr = fetcher & (R < NROW) & ((unsigned)(iy0 + r) < (unsigned)H);
      const unsigned off = (unsigned)(base + (c * H + r) * W) * 4u;
      nx0[u] = sld(rx, off | ((okr & okc0) ? 0u : OOB));
      nx1[u] = sld(rx, (off + 4u) | ((okr & okc1) ? 0u : OOB));
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

  // ---- sign words of a finished tile: its bytes sit in bits[b]; PTH*PTW threads send whole 64-bit words
  auto flush_bits = [&](const TileAt& t, int b) {
    const int ft = tid < PTH * PTW ? tid : 0;
    const int ply = ft / PTW, plx = ft - ply * PTW;
    const int py = t.py0 + ply, px = t.px0 + plx;
    const bool ok = t.valid & (tid < PTH * PTW) & (py < Hp) & (px < Wp);
    const unsigned off = ((unsigned)((t.n * Hp + py) * Wp + px) * 8u) | (ok ? 0u : OOB);
    const u32x2 word = *reinterpret_cast<const u32x2*>(bits + (b * PTH * PTW + ft) * 8);
    sst2(rP, off, word);
    sst2(rM, off, u32x2{0u, 0u});  // nothing is negative after ReLU
  };

  // ---- max pool 3x3 / 2 of a staged tile in POOL_STEPS micro-steps (per pass pj: CTH row maxima, PTH outputs)
  // Step order per pass: row 0, then (row 2k+1, row 2k+2, output k) — at most 3 row maxima are live.
  float hm[CTH];
  auto pool_step = [&](int m, const TileAt& t, int b) {
    const int pj = m / (CTH + PTH), s = m - pj * (CTH + PTH);
    const int pch = pch0 + NW * 8 * pj;
    const bool is_out = s > 0 && (s % 3) == 0;
    if (!is_out) {
      const int row = s == 0 ? 0 : s - s / 3;  // s = 1,2 -> rows 1,2; s = 4,5 -> rows 3,4; ...
      const float* sp = stage + (2 * pplx) * SC + pch + (row * CTW) * SC;
      hm[row] = fmaxf(fmaxf(sp[0], sp[SC]), sp[2 * SC]);
    } else {
      const int ply = s / 3 - 1;
      const int py = t.py0 + ply, px = t.px0 + pplx;
      const bool live = t.valid & (px < Wp) & (py < Hp);
      const float v = fmaxf(fmaxf(hm[2 * ply], hm[2 * ply + 1]), hm[2 * ply + 2]);
      sst(ro, ((unsigned)(((t.n * COUT + pch) * Hp + py) * Wp + px) * 4u) | (live ? 0u : OOB), v);
      const unsigned long long mask = __ballot(live & is_pos(v));
      // lanes 8*plx .. 8*plx+7 hold the 8 channels of one byte of pixel (ply, plx); the leader parks it
      uint8_t* dst = pchl == 0 ? bits + ((b * PTH + ply) * PTW + pplx) * 8 + wave + NW * pj : lds_raw + OFF_DUMMY + tid;
      *dst = (uint8_t)(mask >> (8 * pplx));
    }
  };

  // ---- prologue: tile 0's patch
  int seq = blockIdx.x;
  TileAt cur = locate(seq), p1 = {0, 0, 0, false}, p2 = {0, 0, 0, false};
  fetch(cur);
  commit();
  int buf = 0;
  for (; seq < nseq; seq += gridDim.x) {
    const TileAt nxt = locate(seq + gridDim.x);
    lds_barrier();  // A: patch of `cur` and the staged tile of `p1` are complete
    // ================= phase 1 (one basic block) =================
    fetch(nxt);
    flush_bits(p2, buf);        // tile i-2: its bytes were parked in bits[buf] two phases ago
    const int cy0 = 2 * cur.py0 - 1, cx0 = 2 * cur.px0 - 1;  // conv origin (pool pad 1)
    // (a) the odd sub-tile of this wave's pixel group goes first and is staged apart (stage5): the staged tile itself
    //     is still being pooled, and 8 more accumulator registers across barrier B do not fit.
    {
      f32x4 a5[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) a5[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const half8 ah = load_a(hiP, abase[SUBS - 1] + koff[ks]);
        if constexpr (!HALF) {
          const half8 al = load_a(loP, abase[SUBS - 1] + koff[ks]);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) a5[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks][tt], a5[tt], 0, 0, 0);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) a5[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks][tt], a5[tt], 0, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) a5[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks][tt], a5[tt], 0, 0, 0);
      }
      int m0 = (SUBS * mg + SUBS - 1) * 16 + lg * 4;
      asm volatile("" : "+v"(m0));  // (cy, cx) of the 4 rows are recomputed per tile, not kept in 8 registers
      float* s5 = stage5 + (mg * 16 + lg * 4) * SC + 32 * nh + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + r;
        const int cy = m / CTW, cx = m - cy * CTW;
        const bool inside = ((unsigned)(cy0 + cy) < (unsigned)Hc) & ((unsigned)(cx0 + cx) < (unsigned)Wc);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const float v = fmaxf(fmaf(a5[tt][r], bnc[32 * nh + 16 * tt + li], bnc[COUT + 32 * nh + 16 * tt + li]), 0.0f);
          s5[r * SC + 16 * tt] = inside ? v : 0.0f;  // MaxPool padding: 0 never beats a ReLU output
        }
      }
    }
    // (b) the other sub-tiles in pairs, the previous tile's pooling spread over their MFMA groups
    f32x4 acc[SMAIN][TT];
#pragma unroll
    for (int i = 0; i < SMAIN; ++i)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int g = 0;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
      for (int ip = 0; ip < SMAIN; ip += 2, ++g) {
        half8 ah[2], al[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          ah[d] = load_a(hiP, abase[ip + d] + koff[ks]);
          if constexpr (!HALF) al[d] = load_a(loP, abase[ip + d] + koff[ks]);
        }
        if constexpr (!HALF) {
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[d], bh[ks][tt], acc[ip + d][tt], 0, 0, 0);
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[d], bl[ks][tt], acc[ip + d][tt], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[d], bh[ks][tt], acc[ip + d][tt], 0, 0, 0);
#pragma unroll
        for (int m = (POOL_STEPS * g) / GROUPS; m < (POOL_STEPS * (g + 1)) / GROUPS; ++m) pool_step(m, p1, buf ^ 1);
      }
    }
    lds_barrier();  // B: everybody is done reading the patch and the staged tile
    // ================= phase 2 =================
    {
      float ba[TT], bb[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        ba[tt] = bnc[32 * nh + 16 * tt + li];
        bb[tt] = bnc[COUT + 32 * nh + 16 * tt + li];
      }
      const bool interior = cy0 >= 0 && cx0 >= 0 && cy0 + CTH <= Hc && cx0 + CTW <= Wc;  // workgroup-uniform
      float* sdst = stage + ((SUBS * mg) * 16 + lg * 4) * SC + 32 * nh + li;
      if (interior) {
#pragma unroll
        for (int i = 0; i < SMAIN; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              sdst[(i * 16 + r) * SC + 16 * tt] = fmaxf(fmaf(acc[i][tt][r], ba[tt], bb[tt]), 0.0f);
          }
      } else {
#pragma unroll
        for (int i = 0; i < SMAIN; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = (SUBS * mg + i) * 16 + lg * 4 + r;
            const int cy = m / CTW, cx = m - cy * CTW;
            const bool inside = (unsigned)(cy0 + cy) < (unsigned)Hc && (unsigned)(cx0 + cx) < (unsigned)Wc;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
              const float v = fmaxf(fmaf(acc[i][tt][r], ba[tt], bb[tt]), 0.0f);
              sdst[(i * 16 + r) * SC + 16 * tt] = inside ? v : 0.0f;  // MaxPool padding: 0 never beats a ReLU output
            }
          }
        }
      }
      // the odd sub-tiles' rows: stage5 -> their place in the staged tile (8 floats per thread)
      static_assert(MG * 16 * COUT == NT * 8, "copy mapping");
      const int row = tid >> 3, c8 = (tid & 7) * 8;
      const int m = (SUBS * (row >> 4) + SUBS - 1) * 16 + (row & 15);
      if (m < MPIX) {
        const float4* src = reinterpret_cast<const float4*>(stage5 + row * SC + c8);
        float4* dst = reinterpret_cast<float4*>(stage + m * SC + c8);
        dst[0] = src[0];
        dst[1] = src[1];
      }
    }
    commit();  // next patch: registers -> LDS (fp16 hi/lo)
    p2 = p1;
    p1 = cur;
    cur = nxt;
    buf ^= 1;
  }
  // ---- drain: pool the last tile, send the last two tiles' sign words
  lds_barrier();
  flush_bits(p2, buf);
#pragma unroll
  for (int m = 0; m < POOL_STEPS; ++m) pool_step(m, p1, buf ^ 1);
  lds_barrier();
  flush_bits(p1, buf ^ 1);
}

template <bool HALF>
static int launch_stem_pipe_t(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
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
  const long long want = (long long)cus * (8 / NW);  // 8 waves (two workgroups) per CU
  const unsigned grid = (unsigned)(ntiles < want ? ((ntiles + 7) / 8 * 8) : want);
  static bool attr_set[64] = {false};  // > 64 KB of dynamic LDS needs the opt-in, once per device
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_pipe_kernel<HALF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  // byte sizes of the three streams (the C-ABI caps every tensor below 2^32 bytes)
  const unsigned x_bytes = (unsigned)((long long)N * CIN * H * W * 4);
  const unsigned out_bytes = (unsigned)((long long)N * COUT * Hp * Wp * 4);
  const unsigned plane_bytes = (unsigned)((long long)N * Hp * Wp * 8);
  hipLaunchKernelGGL(stem_pipe_kernel<HALF>, dim3(grid), dim3(NT), LDS_BYTES, stream, x, w, bn_a, bn_b, N, H, W, Hc,
                     Wc, Hp, Wp, tiles_y, tiles_x, per_xcd, out, P, M, x_bytes, out_bytes, plane_bytes);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_stem_pipe(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H, int W,
                     int half, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  return half ? launch_stem_pipe_t<true>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream)
              : launch_stem_pipe_t<false>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream);
}

}  // namespace bnn
