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
//   waves 0..3  MATRIX : issue the global loads of tile t+2's input patch; the implicit GEMM of tile t (M = 17x15
//                        conv pixels, N = 64, K = 24 rows x 8), each wave 8 sub-tiles of 16 pixels x 32 channels
//                        with its B fragments (hi/lo, 6 k-steps) resident in 96 VGPRs and the A fragments
//                        software-pipelined one step ahead (nobody else hides its LDS latency now); then BN + ReLU
//                        of its 64 accumulators into the LDS conv tile ("stage") and the loaded patch, split into
//                        fp16 hi/lo, into the patch buffer the GEMM has just finished with.
//   waves 4..7  POOL   : while the matrix waves multiply tile t, max-pool tile t-1 out of `stage` and store its
//                        fp32 values and sign masks.
//
// Two barriers per tile:  [loads t+2, matrix t | pool t-1]  X  [BN/ReLU t -> stage, patch t+2 -> LDS | sign words
// t-1 -> HBM]  Y.  `stage` is single-buffered (pool t-1 is over before X), the patch is double-buffered; with the
// sign-mask scratch that is 160.0 KB of the CU's 160 KiB LDS.  The waves that LOAD never STORE to global memory:
// gfx950 has one vmcnt for both, a wave that needs a load result while it has stores in flight waits for the
// stores' acknowledgements too (measured: 7-10 k cycles per tile when the pooling waves also fetched).
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
constexpr int OFF_BITS = OFF_STAGE + (MPIX + 1) * SC * 4;  // sign masks of one tile: [8 channel groups][8 rows] x 64 bit
constexpr int LDS_BYTES = OFF_BITS + 8 * PTH * 8;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
}  // namespace stem5

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: the helper waves would sit
// at every barrier until their global STORES (pooled outputs, sign words) are acknowledged by the memory system —
// a ~1 us round trip per tile on the critical path of the matrix waves (measured: +76 us per launch for the sign
// words alone).  Nothing another wave reads goes through global memory here, so only lgkmcnt must reach zero.
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

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
  const int htid = tid - MWAVES * 64;          // 0..255 for the pooling waves
  const int hwave = (wave - MWAVES) & 3;
  // fetch (matrix threads 0..251): column pair `fpc` of patch rows frow0 + 14*u
  const int fpc = tid % NPC;
  int frow0 = tid / NPC;
  const bool fetcher = is_matrix && tid < NPC * RSTEP;
  const int pchl = lane & 7, pplx = lane >> 3;     // pooling: channel within a byte, pooled column (7 = idle)
  float nx0[PER_T], nx1[PER_T];
  unsigned nxmask = 0;  // bit 2u / 2u+1: element u of nx0 / nx1 lies inside the image (else it is zero padding)
  // The helper waves share their SIMD's issue slots with a matrix wave: what they cost is their INSTRUCTION COUNT
  // (first version: 1240 per tile and wave — one exec-mask region per guarded load/store, 64-bit address
  // arithmetic per access, 16 ballots + 16 byte stores — 10 k cycles per tile, more than the matrix phase).
  // Everything below is written to be branch-free per element: clamped addresses + selects instead of guarded
  // loads, one uniform base pointer + one 32-bit lane offset per access stream, one exec region per group.
  const bool last_pair = fpc == NPC - 1;       // its second column is the zero column (kx = 7)

  auto fetch = [&](int tile) {
    const bool valid = tile < ntiles;
    const int tl = valid ? tile : 0;
    const int n = tl / (tiles_y * tiles_x);
    const int tr = tl - n * tiles_y * tiles_x;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int iy0 = 2 * (2 * ty * PTH - 1) - 3, ix0 = 2 * (2 * tx * PTW - 1) - 3;
    const float* xn = x + (size_t)n * CIN * H * W;          // uniform: SGPR base of every load
    const int org = iy0 * W + ix0;                            // may be negative; only used when in range
    const int ix = ix0 + 2 * fpc;
    const bool okc0 = valid && (unsigned)ix < (unsigned)W;
    const bool okc1 = valid && !last_pair && (unsigned)(ix + 1) < (unsigned)W;
    // (row, channel) of the 9 patch rows are loop-invariant; recomputed per tile (a few VALU ops) rather than held in
    // 18 registers next to the GEMM's 226
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(frow0));
#endif
    unsigned mask = 0;
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int R = frow0 + RSTEP * u;
      const int c = (R >= ITH) + (R >= 2 * ITH);
      const int r = R - c * ITH;
      const bool okr = (unsigned)(iy0 + r) < (unsigned)H;
      const bool k0 = okr && okc0, k1 = okr && okc1;
      const int o = org + (c * H + r) * W + 2 * fpc;
      // out-of-image elements read element 0 of the image (always valid); commit() zeroes them.  Nothing here
      // may depend on the loaded VALUES: the loads must stay in flight across the GEMM that follows.
      nx0[u] = xn[k0 ? o : 0];
      nx1[u] = xn[k1 ? o + 1 : 0];
      mask |= (k0 ? 1u : 0u) << (2 * u) | (k1 ? 2u : 0u) << (2 * u);
    }
    nxmask = mask;
  };
  auto commit = [&](int buf) {
    half2v* hiP = reinterpret_cast<half2v*>(hi_plane(buf)) + frow0 * (ROWH / 2) + fpc;
    half2v* loP = reinterpret_cast<half2v*>(lo_plane(buf)) + frow0 * (ROWH / 2) + fpc;
    if (fetcher) {  // one exec region; rows frow0 + 14*u, u < 8, exist for every fetcher
#pragma unroll
      for (int u = 0; u < PER_T; ++u) {
        const float x0 = (nxmask >> (2 * u)) & 1u ? nx0[u] : 0.0f;
        const float x1 = (nxmask >> (2 * u)) & 2u ? nx1[u] : 0.0f;
        half2v h, l;
        h[0] = (_Float16)x0;
        h[1] = (_Float16)x1;
        l[0] = (_Float16)(x0 - (float)h[0]);
        l[1] = (_Float16)(x1 - (float)h[1]);
        if (u < PER_T - 1 || frow0 + RSTEP * u < NROW) {
          hiP[u * RSTEP * (ROWH / 2)] = h;
          if constexpr (!HALF) loP[u * RSTEP * (ROWH / 2)] = l;
        }
      }
    }
  };
  // 3x3 / stride-2 max pool of the staged tile (tn, tpy0, tpx0): a thread owns one pooled COLUMN of one channel
  // (17 row maxima of 3 conv columns -> 8 outputs); 4 helper waves x 2 passes x 8 channels per wave.  Sign bits:
  // lanes 8*plx .. 8*plx+7 hold the 8 channels of group g of pixel column plx, so the ballot of one pooled row IS
  // the row's 7 sign bytes of that group; lane 0 parks the 8 row masks in LDS (`bits`: [group][row] 64-bit).
  auto pool = [&](bool tvalid, int tn, int tpy0, int tpx0) {
    const bool full = tvalid && tpy0 + PTH <= Hp && tpx0 + PTW <= Wp;  // uniform; every tile of a 224x224 input
    const int px = tpx0 + pplx;
    const bool col_live = tvalid && pplx < PTW && px < Wp;
    const unsigned long long colmask = __ballot(col_live);
    float* ob = out ? out + (((size_t)tn * COUT) * Hp + tpy0) * Wp + tpx0 : nullptr;  // uniform base
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int g = hwave + 4 * pass;
      const int pch = 8 * g + pchl;
      float hm[CTH], v[PTH];
      const float* sp = stage + (2 * (pplx < PTW ? pplx : PTW - 1)) * SC + pch;
#pragma unroll
      for (int r = 0; r < CTH; ++r)
        hm[r] = fmaxf(fmaxf(sp[(r * CTW) * SC], sp[(r * CTW + 1) * SC]), sp[(r * CTW + 2) * SC]);
#pragma unroll
      for (int ply = 0; ply < PTH; ++ply) v[ply] = fmaxf(fmaxf(hm[2 * ply], hm[2 * ply + 1]), hm[2 * ply + 2]);
      if (ob) {
        const unsigned voff = (unsigned)(pch * Hp * Wp + pplx);  // lane part of the address, 32 bits
        if (full) {
          if (pplx < PTW) {
#pragma unroll
            for (int ply = 0; ply < PTH; ++ply) ob[voff + (unsigned)(ply * Wp)] = v[ply];
          }
        } else {
#pragma unroll
          for (int ply = 0; ply < PTH; ++ply)
            if (col_live && tpy0 + ply < Hp) ob[voff + (unsigned)(ply * Wp)] = v[ply];
        }
      }
      if (P) {
        unsigned long long rowmask[PTH];
#pragma unroll
        for (int ply = 0; ply < PTH; ++ply)
          rowmask[ply] = (tpy0 + ply < Hp) ? (__ballot(is_pos(v[ply])) & colmask) : 0ull;
        if (lane == 0) {
          unsigned long long* bw = reinterpret_cast<unsigned long long*>(bits) + g * PTH;
#pragma unroll
          for (int ply = 0; ply < PTH; ++ply) bw[ply] = rowmask[ply];
        }
      }
    }
  };
  // The tile's 56 sign words: thread (row, column) collects its byte from the 8 channel groups and sends one
  // whole 64-bit word (byte stores from several waves into one word are slow).
  auto flush_bits = [&](bool tvalid, int tn, int tpy0, int tpx0) {
    if (P && tvalid && htid < PTH * PTW) {
      const int ply = htid / PTW, plx = htid - ply * PTW;
      if (tpy0 + ply < Hp && tpx0 + plx < Wp) {
        const uint8_t* bp = bits + ply * 8 + plx;
        unsigned long long word = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) word |= (unsigned long long)bp[g * PTH * 8] << (8 * g);
        uint64_t* pb = P + ((size_t)tn * Hp + tpy0) * Wp + tpx0;   // uniform base
        const unsigned o = (unsigned)(ply * Wp + plx);
        pb[o] = word;
#ifndef BNN_STEM_TIMING
        (M + ((size_t)tn * Hp + tpy0) * Wp + tpx0)[o] = 0;  // nothing is negative after ReLU
#endif
      }
    }
  };
  auto load_a = [&](const _Float16* plane, int off) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(plane) + (off >> 1);
    u32x4 v;
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    return __builtin_bit_cast(half8, v);
  };

#ifdef BNN_STEM_TIMING  // per-segment cycle sums of every wave, dumped over the M plane (debug builds only)
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define STEM_T(k) { const unsigned long long tn = __builtin_readcyclecounter(); tph[k] += tn - tlast; tlast = tn; }
#define STEM_T_DUMP() if (lane == 0 && M) for (int k = 0; k < 6; ++k) M[((size_t)blockIdx.x * 8 + wave) * 6 + k] = tph[k];
#else
#define STEM_T(k)
#define STEM_T_DUMP()
#endif
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
    // patch pipeline: tiles t0, t1 are split into LDS up front; inside the loop the loads of tile t+2 are issued
    // before the GEMM of tile t and land in LDS (the buffer that GEMM has just released) after barrier X
    if (blockIdx.x < nseq) {
      fetch(tile_of(blockIdx.x));
      commit(0);
    }
    if (blockIdx.x + gridDim.x < nseq) {
      fetch(tile_of(blockIdx.x + gridDim.x));
      commit(1);
    }
    lds_barrier();  // patch[0], patch[1] hold the first two tiles
    for (int seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
      const TileAt t = tile_at(seq);
      const int seq2 = seq + 2 * gridDim.x;
      STEM_T(5)
      if (seq2 < nseq) fetch(tile_of(seq2));  // in flight during the GEMM
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
        // [loads of step+1] | [12 MFMAs of step]: the fences keep the scheduler from sinking the loads next to
        // their uses (it does, to save registers) — they must be a whole step (~200 cycles) ahead, nobody else
        // on this SIMD hides LDS latency for the matrix wave.
        __builtin_amdgcn_sched_barrier(0);
        if (step + 1 < STEPS) lda(step + 1, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
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
      STEM_T(0)
      lds_barrier();  // X: `stage` is free (tile t-1 is pooled), every matrix wave is done with patch[cur]
      STEM_T(1)
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
        // (cy, cx) of the 32 pixels are loop-invariant: laundering the base keeps the compiler from holding (and
        // spilling) 64 quotients/remainders across the persistent loop — two VALU ops each to recompute.
        int mbase = (SUBS * mg) * 16 + lg * 4;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(mbase));
#endif
#pragma unroll
        for (int i = 0; i < SUBS; ++i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mbase + i * 16 + r;
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
      if (seq2 < nseq) commit(cur);  // tile t+2 takes the place of tile t
      STEM_T(2)
      lds_barrier();  // Y: `stage` holds this tile, patch[cur] the tile after next
      STEM_T(3)
      cur ^= 1;
    }
    lds_barrier();    // the pooling waves pool the last tile ...
    STEM_T_DUMP()
    return;             // ... and send its sign words
  }

  // ------------------------------------------------------------------ pooling role
  lds_barrier();
  TileAt prev;
  prev.valid = false;
  prev.n = prev.py0 = prev.px0 = prev.cy0 = prev.cx0 = 0;
  for (int seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
    STEM_T(5)
    if (prev.valid) pool(true, prev.n, prev.py0, prev.px0);
    STEM_T(1)
    lds_barrier();  // X: pool done (`stage` free, sign masks complete)
    STEM_T(2)
    if (prev.valid) flush_bits(true, prev.n, prev.py0, prev.px0);
    STEM_T(3)
    lds_barrier();  // Y: `stage` holds tile `seq`
    STEM_T(4)
    prev = tile_at(seq);
  }
  if (prev.valid) pool(true, prev.n, prev.py0, prev.px0);
  lds_barrier();
  if (prev.valid) flush_bits(true, prev.n, prev.py0, prev.px0);
  STEM_T_DUMP()
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
