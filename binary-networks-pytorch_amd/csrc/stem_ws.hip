// stem_ws.hip — the fused real-valued stem (conv 7x7/2/3 -> BN -> ReLU -> MaxPool 3/2/1 -> fp32 + sign planes;
// bnn/models/resnet.py:93-96,150-153) as a WAVE-SPECIALISED persistent kernel.  Same arithmetic, same bits as
// stem_split.hip (fp32 operands split into fp16 hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16, fp32
// accumulation, every accumulator sees the same MFMA sequence); what changes is who does what, and when.
//
// stem_split.hip walks every tile through fetch -> matrix -> BN/stage -> pool/store with all 8 waves in lock step:
// the phases use different units (HBM, matrix core, LDS, VALU + stores) and add up (13.8 k cycles per tile, of
// which 4.6 k are MFMA issue).  Here the 8 waves of the workgroup (one per CU, persistent) split into two roles,
// one wave of each role on every SIMD:
//
//   waves 0..3  MATRIX : the implicit GEMM of tile t (M = 17x15 conv pixels, N = 64, K = 24 rows x 8), each wave
//                        8 sub-tiles of 16 pixels x 32 channels with its B fragments (hi/lo, 6 k-steps) resident in
//                        96 VGPRs and the A fragments software-pipelined one step ahead (nobody else hides its LDS
//                        latency now); then BN + ReLU of its 64 accumulators into the LDS conv tile ("stage").
//   waves 4..7  HELPER : while the matrix waves multiply tile t — (1) issue the global loads of tile t+1's input
//                        patch, (2) max-pool tile t-1 out of `stage`, store fp32 + sign bytes, (3) split the loaded
//                        patch into fp16 hi/lo and write it to the OTHER patch buffer.
//
// Two barriers per tile:  [matrix t | pool t-1, patch t+1]  X  [BN/ReLU t -> stage | sign words t-1 -> HBM]  Y.
// `stage` is single-buffered (pool t-1 is over before X), the patch is double-buffered; with the sign-byte
// scratch that is 160.4 KB of the CU's 160 KiB LDS.  The tile time becomes max(matrix, helper work) + the
// epilogue instead of their sum.
#include "bnn_dev.h"

namespace bnn {

namespace stem5 {
constexpr int CIN = 3, KS = 7, COUT = 64;
constexpr int KROWS = 24, KSTEPS = KROWS / 4;        // 6 k-steps of 32 (4 rows of 8)
constexpr int PTH = 8, PTW = 7;                      // pooled tile
constexpr int CTH = 2 * PTH + 1, CTW = 2 * PTW + 1;  // conv tile 17 x 15 (pool halo included)
constexpr int MPIX = CTH * CTW;                      // 255
constexpr int ITH = 2 * CTH + 5;                     // 39 input rows
constexpr int ITWP = 36;                             // 35 input columns + 1 zero column (kx = 7)
constexpr int ROWH = 96;                             // halves between patch rows (bank spreading, see stem_split.hip)
constexpr int ICHP = ITH * ROWH;                     // halves per channel plane
constexpr int NINP = CIN * ICHP;                     // halves per plane (22.5 KB)
constexpr int NROW = CIN * ITH;                      // 117 patch rows
constexpr int NPC = ITWP / 2;                        // 18 column pairs per row
constexpr int SC = COUT + 4;                         // stage row stride (floats): conflict-free both ways
constexpr int NT = 512, MWAVES = 4, HT = NT - MWAVES * 64;  // 256 helper threads
constexpr int RSTEP = HT / NPC;                      // 14 rows per sweep (252 fetching threads)
constexpr int PER_T = (NROW + RSTEP - 1) / RSTEP;    // 9 column pairs per helper thread
constexpr int SUBS = 8, TT = 2;                      // sub-tiles and channel tiles per matrix wave
constexpr int STEPS = KSTEPS * (SUBS / 2);           // 24 pipeline steps of 12 MFMAs
constexpr int PLANE_B = ((NINP * 2 + 15) / 16) * 16;
constexpr int OFF_PATCH = 0;                         // [buf][hi, lo]
constexpr int OFF_STAGE = OFF_PATCH + 4 * PLANE_B;
constexpr int OFF_BITS = OFF_STAGE + (MPIX + 1) * SC * 4;  // sign bytes of one tile: [56 pixels][8]
constexpr int LDS_BYTES = OFF_BITS + PTH * PTW * 8;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
}  // namespace stem5

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <bool HALF>
__global__ __launch_bounds__(stem5::NT, 2) void stem_ws_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int N, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
    int tiles_x, int per_xcd, float* __restrict__ out, uint64_t* __restrict__ P,
    uint64_t* __restrict__ M) {
  using namespace stem5;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* stage = reinterpret_cast<float*>(lds_raw + OFF_STAGE);
  uint8_t* bits = lds_raw + OFF_BITS;
  auto hi_plane = [&](int buf) { return reinterpret_cast<_Float16*>(lds_raw + OFF_PATCH + (2 * buf) * PLANE_B); };
  auto lo_plane = [&](int buf) { return reinterpret_cast<_Float16*>(lds_raw + OFF_PATCH + (2 * buf + 1) * PLANE_B); };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_matrix = wave < MWAVES;  // wave-uniform (scalar branch)
  const int li = lane & 15, lg = lane >> 4;

  const int ntiles = N * tiles_y * tiles_x;
  const int nseq = per_xcd * 8;
  // Tile order: workgroup b sits on XCD b % 8 (observed placement, used for speed only).  Each XCD walks ONE
  // contiguous eighth of the tile list: x-neighbours (shared halo, shared output lines) meet in the same L2.
  auto tile_of = [&](int seq) { return (seq & 7) * per_xcd + (seq >> 3); };

  // ------------------------------------------------------------------ matrix role: loop-invariant state
  const int mg = wave & 1, nh = (wave >> 1) & 1;  // pixel half (sub-tiles 8*mg..8*mg+7), channel half
  half8 bh[KSTEPS][TT], bl[KSTEPS][TT];
  int koff[KSTEPS], abase[SUBS];
  float ba[TT], bb[TT];
  if (is_matrix) {
    // MFMA 16x16x32 B operand: lane holds B[k = 8*lg + e][j = li], e = 0..7  ->  row 4*ks + lg, kx = e.
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int krow = 4 * ks + lg;
      const int c = krow / KS, ky = krow - c * KS;
      koff[ks] = krow < CIN * KS ? c * ICHP + ky * ROWH : 0;  // zero-weight rows: any valid address
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
    // A operand: lane holds A[i = li][k = 8*lg + e] = patch[c][2*cy + ky][2*cx + e] of conv pixel m = 16*sub + li.
#pragma unroll
    for (int i = 0; i < SUBS; ++i) {
      int m = (SUBS * mg + i) * 16 + li;
      if (m >= MPIX) m = MPIX - 1;
      const int cy = m / CTW, cx = m - cy * CTW;
      abase[i] = 2 * cy * ROWH + 2 * cx;
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      ba[tt] = bn_a[32 * nh + 16 * tt + li];
      bb[tt] = bn_b[32 * nh + 16 * tt + li];
    }
  }

  // ------------------------------------------------------------------ helper role: loop-invariant state
  const int htid = tid - MWAVES * 64;          // 0..255 for helpers
  const int hwave = (wave - MWAVES) & 3;
  const int fpc = htid % NPC, frow0 = htid / NPC;  // fetch: column pair `fpc` of patch rows frow0 + 14*u
  const bool fetcher = !is_matrix && htid < NPC * RSTEP;
  const int pchl = lane & 7, pplx = lane >> 3;     // pooling: channel within a byte, pooled column (7 = idle)
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
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int R = frow0 + RSTEP * u;
      const int c = R / ITH, r = R - c * ITH;
      const bool okr = fetcher && R < NROW && (unsigned)(iy0 + r) < (unsigned)H;
      const int goff = (c * H + r) * W + 2 * fpc;
      nx0[u] = (okr && okc0) ? xb[goff] : 0.0f;
      nx1[u] = (okr && okc1) ? xb[goff + 1] : 0.0f;
    }
  };
  auto commit = [&](int buf) {
    _Float16* hiP = hi_plane(buf);
    _Float16* loP = lo_plane(buf);
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
  // 3x3 / stride-2 max pool of the staged tile (tn, tpy0, tpx0): a thread owns one pooled COLUMN of one channel
  // (17 row maxima of 3 conv columns -> 8 outputs); 4 helper waves x 2 passes x 8 channels per wave.  The sign
  // bits of 8 channels are gathered with one ballot into a byte of the tile's scratch.
  auto pool = [&](bool tvalid, int tn, int tpy0, int tpx0) {
    const int px = tpx0 + pplx;
    const bool col_live = tvalid && pplx < PTW && px < Wp;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int byte = hwave + 4 * pass;
      const int pch = 8 * byte + pchl;
      float hm[CTH];
      const float* sp = stage + (2 * (pplx < PTW ? pplx : 0)) * SC + pch;
#pragma unroll
      for (int r = 0; r < CTH; ++r)
        hm[r] = fmaxf(fmaxf(sp[(r * CTW) * SC], sp[(r * CTW + 1) * SC]), sp[(r * CTW + 2) * SC]);
#pragma unroll
      for (int ply = 0; ply < PTH; ++ply) {
        const int py = tpy0 + ply;
        const bool live = col_live && py < Hp;
        const float v = fmaxf(fmaxf(hm[2 * ply], hm[2 * ply + 1]), hm[2 * ply + 2]);
        if (live && out) out[(((size_t)tn * COUT + pch) * Hp + py) * Wp + px] = v;
        if (P) {  // lanes 8*plx .. 8*plx+7 hold the 8 channels of byte `byte` of pixel (ply, plx)
          const unsigned long long mask = __ballot(live && is_pos(v));
          if (pchl == 0 && pplx < PTW) bits[(ply * PTW + pplx) * 8 + byte] = (uint8_t)(mask >> (8 * pplx));
        }
      }
    }
  };
  // the tile's 56 sign words leave as whole 64-bit stores (byte stores from several waves into one word are slow)
  auto flush_bits = [&](bool tvalid, int tn, int tpy0, int tpx0) {
    if (P && tvalid && htid < PTH * PTW) {
      const int ply = htid / PTW, plx = htid - ply * PTW;
      const int py = tpy0 + ply, px = tpx0 + plx;
      if (py < Hp && px < Wp) {
        const size_t o = ((size_t)tn * Hp + py) * Wp + px;
        P[o] = *reinterpret_cast<const uint64_t*>(bits + htid * 8);
        M[o] = 0;  // nothing is negative after ReLU
      }
    }
  };
  auto load_a = [&](const _Float16* plane, int off) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(plane) + (off >> 1);
    u32x4 v;
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    return __builtin_bit_cast(half8, v);
  };

  // Decode a tile index (workgroup-uniform).
  struct TileAt { bool valid; int n, py0, px0, cy0, cx0; };
  auto tile_at = [&](int seq) {
    TileAt t;
    const int tile = tile_of(seq);
    t.valid = tile < ntiles;
    const int tl = t.valid ? tile : 0;
    t.n = tl / (tiles_y * tiles_x);
    const int tr = tl - t.n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    t.py0 = ty * PTH;                                    // pooled origin
    t.px0 = tx * PTW;
    t.cy0 = 2 * t.py0 - 1;                               // conv origin (pool pad 1)
    t.cx0 = 2 * t.px0 - 1;
    return t;
  };

  // The two roles run SEPARATE loops over the same tile sequence and meet at the same barriers (one before the
  // loop, two per tile, one after): with a shared loop body the register allocator sees every loop invariant of
  // one role as live through the other role's code (it cannot know the branches are exclusive per wave) and
  // spills ~90 registers around the MFMA block.
  if (is_matrix) {
    int cur = 0;
    __syncthreads();  // patch[0] holds the first tile
    for (int seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
      const TileAt t = tile_at(seq);
      // ---- implicit GEMM: 8 sub-tiles x 2 channel tiles x 6 k-steps x (lo*hi + hi*lo + hi*hi), two sub-tiles
      // per step (4 independent accumulators between two MFMAs on the same one), A fragments one step ahead.
      const _Float16* hiP = hi_plane(cur);
      const _Float16* loP = lo_plane(cur);
      f32x4 acc[SUBS][TT];
#pragma unroll
      for (int i = 0; i < SUBS; ++i)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[i][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      // The LDS offsets abase[i] + koff[ks] are loop-invariant; left alone, the compiler keeps all 48 sums in
      // registers across the persistent loop.  Laundering the addends makes it re-add per tile.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int i = 0; i < SUBS; ++i) asm volatile("" : "+v"(abase[i]));
#endif
      half8 ah[2][2], al[2][2];
      auto lda = [&](int step, int b) {
        const int ks = step / (SUBS / 2), ip = 2 * (step % (SUBS / 2));
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          ah[b][d] = load_a(hiP, abase[ip + d] + koff[ks]);
          if constexpr (!HALF) al[b][d] = load_a(loP, abase[ip + d] + koff[ks]);
        }
      };
      lda(0, 0);
#pragma unroll
      for (int step = 0; step < STEPS; ++step) {
        const int b = step & 1, ks = step / (SUBS / 2), ip = 2 * (step % (SUBS / 2));
        __builtin_amdgcn_sched_barrier(0);  // keep the pipeline one step deep: more prefetch only buys spills
        if (step + 1 < STEPS) lda(step + 1, b ^ 1);
        if constexpr (!HALF) {
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b][d], bh[ks][tt], acc[ip + d][tt], 0, 0, 0);
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][d], bl[ks][tt], acc[ip + d][tt], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            acc[ip + d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][d], bh[ks][tt], acc[ip + d][tt], 0, 0, 0);
      }
      __syncthreads();  // X: `stage` is free (the helpers pooled tile t-1), patch[cur^1] is written
      // ---- BN + ReLU, conv tile -> LDS.  D layout: column = li (channel), row = 4*lg + r (pixel).
      // Three quarters of the tiles lie entirely inside the conv output: no per-pixel range tests there.
      const bool interior = t.cy0 >= 0 && t.cx0 >= 0 && t.cy0 + CTH <= Hc && t.cx0 + CTW <= Wc;  // uniform
      float* sdst = stage + ((SUBS * mg) * 16 + lg * 4) * SC + 32 * nh + li;
      if (interior) {
#pragma unroll
        for (int i = 0; i < SUBS; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)  // row MPIX (the 256th pixel) exists in `stage` and is never read
              sdst[(i * 16 + r) * SC + 16 * tt] = fmaxf(fmaf(acc[i][tt][r], ba[tt], bb[tt]), 0.0f);
      } else {
#pragma unroll
        for (int i = 0; i < SUBS; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = (SUBS * mg + i) * 16 + lg * 4 + r;
            const int cy = m / CTW, cx = m - cy * CTW;
            const bool inside = (unsigned)(t.cy0 + cy) < (unsigned)Hc && (unsigned)(t.cx0 + cx) < (unsigned)Wc;
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
              const float v = fmaxf(fmaf(acc[i][tt][r], ba[tt], bb[tt]), 0.0f);
              // positions outside the conv output are MaxPool padding: 0 never beats a ReLU output
              sdst[(i * 16 + r) * SC + 16 * tt] = inside ? v : 0.0f;
            }
          }
        }
      }
      __syncthreads();  // Y: `stage` holds this tile
      cur ^= 1;
    }
    __syncthreads();    // the helpers pool the last tile ...
    return;             // ... and send its sign words
  }

  // ------------------------------------------------------------------ helper role
  int cur = 0;
  if (blockIdx.x < nseq) {
    fetch(tile_of(blockIdx.x));
    commit(0);
  }
  __syncthreads();  // patch[0] holds the first tile
  TileAt prev;
  prev.valid = false;
  prev.n = prev.py0 = prev.px0 = prev.cy0 = prev.cx0 = 0;
  for (int seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
    const int seq_next = seq + gridDim.x;
    if (seq_next < nseq) fetch(tile_of(seq_next));  // global loads fly while the previous tile is pooled
    if (prev.valid) pool(true, prev.n, prev.py0, prev.px0);
    if (seq_next < nseq) commit(cur ^ 1);
    __syncthreads();  // X: pool done (`stage` free, sign bytes complete), patch[cur^1] written
    if (prev.valid) flush_bits(true, prev.n, prev.py0, prev.px0);
    __syncthreads();  // Y: `stage` holds tile `seq`
    prev = tile_at(seq);
    cur ^= 1;
  }
  if (prev.valid) pool(true, prev.n, prev.py0, prev.px0);
  __syncthreads();
  if (prev.valid) flush_bits(true, prev.n, prev.py0, prev.px0);
}

template <bool HALF>
static int launch_stem_ws_t(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                            int W, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  using namespace stem5;
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
  const long long want = cus;  // one resident workgroup (8 waves) per CU
  const unsigned grid = (unsigned)(ntiles < want ? ((ntiles + 7) / 8 * 8) : want);
  static bool attr_set[64] = {false};  // > 64 KB of dynamic LDS needs the opt-in, once per device
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_ws_kernel<HALF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(stem_ws_kernel<HALF>, dim3(grid), dim3(NT), LDS_BYTES, stream, x, w, bn_a, bn_b, N, H,
                     W, Hc, Wc, Hp, Wp, tiles_y, tiles_x, per_xcd, out, P, M);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_stem_ws(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                   int W, int half, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  return half ? launch_stem_ws_t<true>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream)
              : launch_stem_ws_t<false>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream);
}

}  // namespace bnn
