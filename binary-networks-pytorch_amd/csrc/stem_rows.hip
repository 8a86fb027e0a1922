// stem_rows.hip — the fused stem (see stem.hip for the layer) with the max-pool done IN THE ACCUMULATORS.
// Arithmetic as in stem_split.hip: fp32 operands split into fp16 hi + lo, product = lo*hi + hi*lo + hi*hi on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation (the rounding class of an fp32 convolution); results are
// bit-identical to that kernel's.
//
// What is different (round 3; stem_split.hip spent more cycles outside its matrix phase than inside):
//  * the GEMM is TRANSPOSED: A = weights (16 channels x K), B = patch (K x 16 conv pixels).  A lane's four
//    accumulator registers are then four CHANNELS of ONE conv pixel, and the 16 lanes of a DPP row are 16
//    neighbouring conv pixels of one conv row.  The 3x3 / stride-2 max-pool becomes
//        vertical:   v_max3_f32 over the accumulators of three conv rows (registers of the same lane),
//        horizontal: two v_max_f32 with row_shr:1 / row_shl:1 DPP operands,
//    i.e. 4 VALU instructions per pooled value and register — the conv tile is never staged through LDS (the
//    old kernel: 40 ds_write + 54 ds_read per lane and tile, two more barriers, 1.5 k + 3 k cycles of 13 k);
//  * a wave owns a strip of 7 pooled columns (15 conv columns = one 16-pixel MFMA tile) x 32 channels and walks
//    DOWN the image: per pooled row two new conv rows (four independent accumulator chains) + the last conv row of
//    the step before, which it keeps in registers — also across tiles, so that no conv row is computed twice
//    (the 9th row of a 4-row tile with its own halo was 11 % of the MFMAs); 16 accumulator registers instead of 40;
//  * a pooled row leaves as soon as it exists (fp32 stores and sign bits sit between the MFMA batches of the
//    same wave and under the MFMAs of the other wave of the SIMD);
//  * the B fragments are requested one step (12 MFMAs) ahead of their use, across pooled rows;
//  * patch rows hold the hi pairs and, 40 dwords further, the lo pairs; one conv row = two patch rows further
//    is 640 B further: ONE address register per k-step reaches every fragment of a pooled-row step through the
//    instruction's immediate offsets (the old kernel: 95 v_add_u32 per tile for LDS addresses);
//  * the patch is double-buffered in LDS (2 x 22 KB per workgroup): one barrier per tile;
//  * the patch is fetched with buffer loads (zero padding = an offset beyond the descriptor; a tile-invariant
//    per-lane offset per load + one wave-uniform offset per tile: no address arithmetic in the loop for the tiles
//    whose patch lies inside the image); it starts at an EVEN input column (kx slot e = kx + 1, e = 0 carries
//    zero weights), so that a thread's column pair is one dword of the fp16 planes;
//  * tiles are walked strip-major (image, column strip, row): the next tile is one increment away (the old
//    kernel: two integer divisions per tile on the scalar unit); the strips of an image are walked side by side by
//    neighbouring workgroups of one XCD, so that their 56-byte pieces of shared output lines meet in that L2.
// Tile: 4 x 14 pooled outputs per 4-wave workgroup, two workgroups per CU; wave = (column half, channel half).
#include <type_traits>

#include "bnn_dev.h"

#ifndef BNN_ROWS_K16TAIL  // 1: the last k-step (k-row 20 and three zero rows) as v_mfma_f32_16x16x16_f16 (k-rows 20, 21)
#define BNN_ROWS_K16TAIL 1
#endif
#ifndef BNN_ROWS_LEAN  // 1: fewer live registers (BatchNorm constants read from LDS where a pooled row is finished, one
#define BNN_ROWS_LEAN 0  // walking fragment base per k-step kept across tiles) — so that two stem waves leave room on a SIMD
#endif                   // for waves of ANOTHER kernel (the other stream's binary convolutions: VALU work beside the MFMAs)
#ifndef BNN_ROWS_PRIO   // s_setprio of the stem's waves (0 = default priority)
#define BNN_ROWS_PRIO 0
#endif
#ifdef BNN_ROWS_VGPRS   // register budget of the kernel (allocation granule 8; two waves per SIMD leave 512 - 2 * budget)
#define BNN_ROWS_VGPR_ATTR __attribute__((amdgpu_num_vgpr(BNN_ROWS_VGPRS)))
#else
#define BNN_ROWS_VGPR_ATTR
#endif
#ifndef BNN_ROWS_ABL  // timing ablations only (wrong results): 1 matrix, 2 epilogue, 4 fragment reads, 8 fetch + commit,
#define BNN_ROWS_ABL 0  // 16 fp32 stores, 32 sign words
#endif

namespace bnn {

namespace stemr {
constexpr int CIN = 3, KS = 7, COUT = 64;
constexpr int KSTEPS = 6;                            // 24 k-rows (c, ky) of 8 kx slots; rows 21..23 are zero
constexpr int PTH = 4, PTW = 14, WPW = PTW / 2;      // pooled tile; pooled columns per wave
constexpr int NT = 256, NW = NT / 64;
constexpr int CTH = 2 * PTH + 1;                     // 9 conv rows: row 0 is the one kept from the tile above
constexpr int ITH = 2 * CTH + 5;                     // 23 input rows per channel
constexpr int NROW = CIN * ITH;                      // 69 patch rows
constexpr int NPC = 32;                              // fetched column pairs per row (columns 0..63; 64, 65 stay zero)
constexpr int ROWD = 80, LO_D = 40;                  // dwords per patch row (16 mod 64 banks); offset of the lo pairs
constexpr int RSTEP = NT / NPC;                      // 8 rows per sweep
constexpr int PER_T = (NROW + RSTEP - 1) / RSTEP;    // 9 column pairs per thread
constexpr int PATCH_D = NROW * ROWD;                 // dwords per patch buffer
constexpr int OFF_BITS = 2 * PATCH_D * 4;            // sign words of the tile: [2][56 pixels][2 halves] u32
constexpr int OFF_BN = OFF_BITS + 2 * PTH * PTW * 8;  // BNN_ROWS_LEAN: BatchNorm constants [a|b][64 channels] fp32
#ifdef BNN_ROWS_TIMING  // per-phase shader-clock sums of every wave (debug builds): [wave][16 phases][64 lanes] u32 in LDS
constexpr int OFF_TIME = OFF_BN + 2 * COUT * 4;
constexpr int LDS_BYTES = OFF_TIME + NW * 16 * 64 * 4;
#else
constexpr int LDS_BYTES = OFF_BN + 2 * COUT * 4;
#endif
constexpr int CONV_ROW_D = 2 * ROWD;                 // one conv row further = two input rows further
static_assert(CONV_ROW_D + LO_D + 3 < 256, "two conv rows must stay inside ds_read2_b32's offsets");
constexpr int NSTEP = KSTEPS * PTH;                  // (pooled row, k-step) steps per tile
}  // namespace stemr

using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

#if defined(__HIP_DEVICE_COMPILE__)
using RowsRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ RowsRsrc rows_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
}
__device__ __forceinline__ float rows_ld(RowsRsrc r, unsigned voff, unsigned soff) {  // out of range reads as 0
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// Two floats per lane.  NOT __builtin_amdgcn_raw_buffer_load_b64: hipcc 7.2 narrows that load to ONE dword when its two
// elements are extracted (both read the first; tools/experiments/README.md 55) — the float-vector form of the same
// intrinsic is selected as buffer_load_dwordx2.
using f32x2v = __attribute__((ext_vector_type(2))) float;
__device__ f32x2v rows_ld2_intrinsic(RowsRsrc r, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.v2f32");
__device__ __forceinline__ f32x2v rows_ld2(RowsRsrc r, unsigned voff, unsigned soff) {  // out of range reads as 0
  return rows_ld2_intrinsic(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void rows_st(RowsRsrc r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void rows_st2(RowsRsrc r, unsigned voff, unsigned soff, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}
// One pooled value per lane from the three conv rows a, b, c of its column: v_max3 (vertical), ReLU (max and relu
// commute; values are >= 0 from here on), then the maximum with the two neighbouring conv columns as two v_max_f32
// with a DPP operand.  Lanes 0..7 of a 16-lane row hold the EVEN conv columns 0, 2, .. 14 of the wave's strip, lanes
// 8..15 the odd ones 1, 3, .. 15: pooled column j (centre column 2j + 1) ends up on lane 8 + j, its neighbours sit 8
// and 7 lanes below (row_shr:8, row_shr:7; a missing source reads as 0) — the seven pooled values of a row leave from
// seven ADJACENT lanes.  Hand-written: from the builtins hipcc emits v_mov_dpp + a canonicalising v_max + the v_max
// per neighbour and canonicalises a, b, c.  The eight values of a pooled row (2 channel tiles x 4 registers) go through
// each step TOGETHER, one asm block per step: neighbours in the stream are independent (no dependent-issue stalls) and
// the two wait states between a VALU write and a DPP read of the same register are covered by the other values'
// instructions instead of an s_nop (round 5; one value at a time: +90 cycles per pooled row).
__device__ __forceinline__ void pool_rows4(float (&t)[4], const float (&a)[4], const float (&b)[4], const float (&c)[4]) {
  asm("v_max3_f32 %0, %4, %8, %12\n\tv_max3_f32 %1, %5, %9, %13\n\tv_max3_f32 %2, %6, %10, %14\n\tv_max3_f32 %3, %7, %11, %15"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
        "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
}
__device__ __forceinline__ void pool_cols8(float (&v)[8], float (&t)[8]) {
  asm("v_max_f32_e32 %8, 0, %8\n\tv_max_f32_e32 %9, 0, %9\n\tv_max_f32_e32 %10, 0, %10\n\tv_max_f32_e32 %11, 0, %11\n\t"
      "v_max_f32_e32 %12, 0, %12\n\tv_max_f32_e32 %13, 0, %13\n\tv_max_f32_e32 %14, 0, %14\n\tv_max_f32_e32 %15, 0, %15\n\t"
      "v_max_f32_dpp %0, %8, %8 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %1, %9, %9 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %2, %10, %10 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %3, %11, %11 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %4, %12, %12 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %5, %13, %13 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %6, %14, %14 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %7, %15, %15 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %0, %8, %0 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %1, %9, %1 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %2, %10, %2 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %3, %11, %3 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %4, %12, %4 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %5, %13, %5 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %6, %14, %6 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_max_f32_dpp %7, %15, %7 row_shr:7 row_mask:0xf bank_mask:0xf bound_ctrl:0"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
        "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
}
// bit r of z0 = (v[r] is a positive number), bit r of z1 = (v[4 + r] is): word = 2 * word + bit as v_cmp_class + v_addc
// (the carry-in IS the bit), two independent carry chains (vcc and a scalar pair), the first bit of each placed by a select
__device__ __forceinline__ void sign_bits8(uint32_t& z0, uint32_t& z1, const float (&v)[8]) {
  uint64_t c1;
  asm("v_cmp_class_f32 vcc, %6, %11\n\t"
      "v_cmp_class_f32 %2, %10, %11\n\t"
      "v_cndmask_b32_e64 %0, 0, 1, vcc\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, %2\n\t"
      "v_cmp_class_f32 vcc, %5, %11\n\t"
      "v_cmp_class_f32 %2, %9, %11\n\t"
      "v_addc_co_u32_e64 %0, vcc, %0, %0, vcc\n\t"
      "v_addc_co_u32_e64 %1, %2, %1, %1, %2\n\t"
      "v_cmp_class_f32 vcc, %4, %11\n\t"
      "v_cmp_class_f32 %2, %8, %11\n\t"
      "v_addc_co_u32_e64 %0, vcc, %0, %0, vcc\n\t"
      "v_addc_co_u32_e64 %1, %2, %1, %1, %2\n\t"
      "v_cmp_class_f32 vcc, %3, %11\n\t"
      "v_cmp_class_f32 %2, %7, %11\n\t"
      "v_addc_co_u32_e64 %0, vcc, %0, %0, vcc\n\t"
      "v_addc_co_u32_e64 %1, %2, %1, %1, %2"
      : "=&v"(z0), "=&v"(z1), "=&s"(c1)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "s"(kClassPos)
      : "vcc");
}
// OR of the four 16-lane rows of a wave, in every lane: two lane-swap instructions (no LDS round trip)
__device__ __forceinline__ uint32_t or_rows(uint32_t w) {
  const auto a = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  w = a[0] | a[1];
  const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);
  return b[0] | b[1];
}
#else  // the host pass of hipcc only parses these
struct RowsRsrc {};
__device__ __forceinline__ RowsRsrc rows_rsrc(const void*, unsigned) { return {}; }
__device__ __forceinline__ float rows_ld(RowsRsrc, unsigned, unsigned) { return 0.0f; }
using f32x2v = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2v rows_ld2(RowsRsrc, unsigned, unsigned) { return f32x2v{0.0f, 0.0f}; }
__device__ __forceinline__ void rows_st(RowsRsrc, unsigned, unsigned, float) {}
__device__ __forceinline__ void rows_st2(RowsRsrc, unsigned, unsigned, u32x2) {}
__device__ __forceinline__ void pool_rows4(float (&t)[4], const float (&a)[4], const float (&)[4], const float (&)[4]) { for (int i = 0; i < 4; ++i) t[i] = a[i]; }
__device__ __forceinline__ void pool_cols8(float (&v)[8], float (&t)[8]) { for (int i = 0; i < 8; ++i) v[i] = t[i]; }
__device__ __forceinline__ void sign_bits8(uint32_t& z0, uint32_t& z1, const float (&)[8]) { z0 = z1 = 0; }
__device__ __forceinline__ uint32_t or_rows(uint32_t w) { return w; }
#endif
constexpr unsigned kRowsOOB = 0xFFFFFFF0u;  // beyond every descriptor's num_records

// HALF: plain fp16 operands, one MFMA per product (BNN_HIP_STEM_FP16).
// RAW (round 5, the training forward: bnn_hip_stem7x7_conv_f32): the convolution alone — every conv row leaves as it is
// finished, fp32 [N, 64, Hc, Wc]; no BatchNorm, no pooling, no sign planes (bn_a / bn_b / P / M unused).  The same
// MFMA stream, so the values are the ones the fused kernel normalises and pools.
// AFF (round 6): the sign planes are those of fmaf(y, pk_a[c], pk_b[c]) instead of y — the first binary layer of a
// pre-activation network (a hierarchical block: bn1 -> relu -> conv1, hierarchical_block.py:39) binarises the stem's output
// behind its own BatchNorm; with it the separate packing pass over the fp32 tensor disappears.  The fp32 output is unchanged.
template <bool HALF, bool RAW = false, bool AFF = false>
__global__ __launch_bounds__(stemr::NT, 2) BNN_ROWS_VGPR_ATTR void stem_rows_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bn_a,
    const float* __restrict__ bn_b, int N, int H, int W, int Hc, int Wc, int Hp, int Wp, int tiles_y,
    int tiles_x, int seg_len, int nseg, unsigned x_bytes, float* __restrict__ out, uint64_t* __restrict__ P,
    uint64_t* __restrict__ M, unsigned out_bytes, unsigned plane_bytes, const float* __restrict__ pk_a = nullptr,
    const float* __restrict__ pk_b = nullptr) {
  static_assert(!(AFF && RAW), "the raw convolution writes no planes");
  using namespace stemr;
#ifndef BNN_ROWS_AHEAD  // fragment sets requested ahead in the split mode (each set: 32 VGPRs of the ring)
#define BNN_ROWS_AHEAD 1
#endif
  constexpr int AHEAD = HALF ? 3 : BNN_ROWS_AHEAD, RING = AHEAD + 1;  // (a HALF step is 4 MFMAs long)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_rows[];
  uint32_t* patch = reinterpret_cast<uint32_t*>(lds_rows);
  uint32_t* bits = reinterpret_cast<uint32_t*>(lds_rows + OFF_BITS);  // double-buffered: tile t's words leave during t + 1

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: what depends on it is uniform
  const int li = lane & 15, lg = lane >> 4;
  const int lcol = li < 8 ? 2 * li : 2 * li - 15;  // conv column of the strip held by this lane (see pool_cols8)
  const int mg = wave & 1, nh = wave >> 1;  // pooled columns 7*mg .. 7*mg + 6, channels 32*nh .. 32*nh + 31

  // ---- once: zero both patch buffers (columns 64, 65 and the row padding are read by the idle pixel column and
  // by the zero-weight kx slot: they must stay finite)
  for (int i = tid; i < 2 * PATCH_D; i += NT) patch[i] = 0u;

  // ---- once: A fragments (hi, lo) of this wave's 2 channel tiles x 6 k-steps.
  // MFMA 16x16x32 A operand: lane holds A[i = li][k = 8*lg + e]  ->  channel 16*tt + li, k-row 4*ks + lg, kx = e - 1.
  constexpr bool K16 = BNN_ROWS_K16TAIL != 0 && !HALF;
  half8 wh[KSTEPS][2], wl[KSTEPS][2];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int krow = 4 * ks + lg;
    const int c = krow / KS, ky = krow - c * KS;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int o = 32 * nh + 16 * tt + li;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (krow < CIN * KS && e >= 1) ? w[((size_t)(o * CIN + c) * KS + ky) * KS + e - 1] : 0.0f;
        if (K16 && ks == KSTEPS - 1) {
          // 16x16x16 A operand: lane holds A[i = li][k = 4*lg + e], e < 4: k-row 20 + (lg >> 1), kx slot 4*(lg & 1) + e
          const int slot = 4 * (lg & 1) + e;
          v = (lg < 2 && e < 4 && slot >= 1) ? w[((size_t)(o * CIN + CIN - 1) * KS + KS - 1) * KS + slot - 1] : 0.0f;
        }
        const _Float16 h = (_Float16)v;
        wh[ks][tt][e] = h;
        wl[ks][tt][e] = (_Float16)(v - (float)h);
      }
    }
  }
  // B operand: lane holds B[k = 8*lg + e][j = li] = patch[c][2*t + ky][2*(14*mg + lcol) + e] of conv pixel (t, 14*mg + lcol).
  // kb[ks]: byte offset of (k-row 4*ks + lg, conv row 0, column pair 14*mg + lcol); zero-weight k-rows read k-row 20.
  // (LDS ADDRESSES, the base of the dynamic-LDS symbol included: added per read it is a `v_add_u32 v, 0, v` in front of
  // every fragment request — the symbol's offset is only known to the assembler)
  using lds_byte = __attribute__((address_space(3))) unsigned char;
  using lds_word = __attribute__((address_space(3))) const uint32_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte*)lds_rows;
  uint32_t kb[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    int krow = 4 * ks + lg;
    if (krow >= CIN * KS) krow = CIN * KS - 1;
    const int c = krow / KS, ky = krow - c * KS;
    kb[ks] = lds0 + (uint32_t)((c * ITH + ky) * ROWD + 2 * WPW * mg + lcol) * 4u;
    if (K16 && ks == KSTEPS - 1)  // k-row 20 for every lane group, its kx slots 4*(lg & 1) .. + 3 (two column pairs)
      kb[ks] = lds0 + (uint32_t)(((CIN - 1) * ITH + KS - 1) * ROWD + 2 * WPW * mg + lcol + 2 * (lg & 1)) * 4u;
  }
  // BN constants of the accumulator layout (register r of tile tt -> channel 32*nh + 16*tt + 4*lg + r)
  constexpr bool LEAN = BNN_ROWS_LEAN != 0 && !RAW;
  [[maybe_unused]] float ba[2][4], bb[2][4];
  using lds_f4 = __attribute__((address_space(3))) const f32x4;
  [[maybe_unused]] const uint32_t bn_lane = lds0 + (uint32_t)(OFF_BN + (32 * nh + 4 * lg) * 4);  // + 64 * tt, + 256 for b
  if constexpr (LEAN) {
    if (tid < COUT) {  // visible after the barrier below ("the zero fill is complete")
      float* bnl = reinterpret_cast<float*>(lds_rows + OFF_BN);
      bnl[tid] = bn_a[tid];
      bnl[COUT + tid] = bn_b[tid];
    }
  } else if constexpr (!RAW) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ba[tt][r] = bn_a[32 * nh + 16 * tt + 4 * lg + r];
        bb[tt][r] = bn_b[32 * nh + 16 * tt + 4 * lg + r];
      }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // waited for HERE: inside the loop the wait would also cover the next patch's loads
        asm volatile("" : "+v"(ba[tt][r]));
        asm volatile("" : "+v"(bb[tt][r]));
      }
  }
  [[maybe_unused]] float pka[2][4], pkb[2][4];
  if constexpr (AFF) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pka[tt][r] = pk_a[32 * nh + 16 * tt + 4 * lg + r];
        pkb[tt][r] = pk_b[32 * nh + 16 * tt + 4 * lg + r];
      }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        asm volatile("" : "+v"(pka[tt][r]));
        asm volatile("" : "+v"(pkb[tt][r]));
      }
  }
#if BNN_ROWS_PRIO
  __builtin_amdgcn_s_setprio(BNN_ROWS_PRIO);
#endif
  // fetch role: column pair `fpc` of patch rows frow0 + 8*u.  Buffer loads: zero padding = an offset beyond the
  // descriptor, one tile-invariant per-lane offset per load + a wave-uniform offset per tile (no address arithmetic
  // and no exec-mask regions in the tile loop)
  const int fpc = tid % NPC, frow0 = tid / NPC;
  unsigned rowoff[PER_T];  // bytes from (channel 0, first patch row, first patch column) of the tile's image
#pragma unroll
  for (int u = 0; u < PER_T; ++u) {
    const int R = frow0 + RSTEP * u;
    const int c = (R >= 2 * ITH) + (R >= ITH), r = R - c * ITH;
    rowoff[u] = R < NROW ? (unsigned)(((c * H + r) * W + 2 * fpc) * 4) : kRowsOOB;
  }
  const RowsRsrc r_x = rows_rsrc(x, x_bytes);
  const RowsRsrc r_out = rows_rsrc(out, out_bytes), r_P = rows_rsrc(P, plane_bytes), r_M = rows_rsrc(M, plane_bytes);
  // output role: lanes 8 .. 14 of a row hold pooled columns 0 .. 6 after the horizontal maximum
  const int plx = li - 8;
  const bool pool_lane = li >= 8 && li < 8 + WPW;
  const unsigned out_lane = (unsigned)((4 * lg * Hp) * Wp + plx) * 4u;      // channel 4*lg (+ r), pooled column plx
  const unsigned flush_lane = (unsigned)((tid / PTW) * Wp + tid % PTW) * 8u;  // pixel (tid / PTW, tid % PTW) of a tile
  // RAW: a wave stores conv columns 1 .. 14 of its strip (column 0 is its left neighbour's column 14, column 15 idles):
  // channel 4*lg (+ r), conv column 14*mg + lcol - 1 of the tile's 28
  [[maybe_unused]] const unsigned raw_lane = (unsigned)((4 * lg * Hc) * Wc + 2 * WPW * mg + lcol - 1) * 4u;
  [[maybe_unused]] const bool raw_col = lcol >= 1 && lcol <= 2 * WPW;

  // Tile list: strip-major (image, column strip, row).  A workgroup works through `nseg` segments of `seg_len`
  // consecutive tiles; segment j of logical workgroup wg starts at tile (wg + j * grid) * seg_len.  With at least
  // one strip per workgroup a segment is a whole strip: the four strips of an image are then walked by four
  // workgroups side by side and in step, so that the 56-byte pieces they write into shared cache lines meet in the
  // L2 (workgroup b sits on XCD b % 8 — observed placement, used for speed only — and logical neighbours share it).
  const int ntiles = N * tiles_y * tiles_x;
  const int wg = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int seg_stride = (int)gridDim.x * seg_len;
  const int niter = nseg * seg_len;
  float nx0[PER_T], nx1[PER_T];
  // even width and an 8-byte aligned tensor: every column pair of the patch (it starts at an even column) is one aligned
  // 8-byte word that lies inside or outside the image as a whole
  const bool pair_loads = !(W & 1) && !((uintptr_t)x & 7u);
  auto fetch = [&](int n, int tx, int ty) {
    const int iy0 = 4 * ty * PTH - 5, ixe = 4 * tx * PTW - 6;  // first input row; first (even) input column
    const unsigned img = (unsigned)(n * CIN * H * W) * 4u;
    const int toff = (iy0 * W + ixe) * 4;
    const bool rows_in = iy0 >= 0 && iy0 + ITH <= H;
    if (rows_in && pair_loads && ixe >= 0 && ixe + 2 * NPC <= W) {  // the patch lies inside the image
#pragma unroll
      for (int u = 0; u < PER_T; ++u) {  // a column pair per 8-byte load (half the requests of the texture addresser)
        const f32x2v v = rows_ld2(r_x, rowoff[u], img + (unsigned)toff);
        nx0[u] = v.x;
        nx1[u] = v.y;
      }
    } else if (rows_in && pair_loads) {
      // left / right edge of an even-width image: a column pair is inside or outside as a whole (the patch starts at
      // an even column) — one select per load
      const bool okc = (unsigned)(ixe + 2 * fpc) < (unsigned)W;
      const unsigned rowpart = img + (unsigned)(iy0 * W * 4);
#pragma unroll
      for (int u = 0; u < PER_T; ++u) {
        const unsigned off = (okc && frow0 + RSTEP * u < NROW) ? rowoff[u] + (unsigned)(ixe * 4) : kRowsOOB;
        const f32x2v v = rows_ld2(r_x, off, rowpart);
        nx0[u] = v.x;
        nx1[u] = v.y;
      }
    } else {
      const int ix = ixe + 2 * fpc;
      const bool okc0 = (unsigned)ix < (unsigned)W;
      const bool okc1 = (unsigned)(ix + 1) < (unsigned)W;
#pragma unroll
      for (int u = 0; u < PER_T; ++u) {
        const int R = frow0 + RSTEP * u;
        const int c = (R >= 2 * ITH) + (R >= ITH), r = R - c * ITH;
        const bool okr = R < NROW && (unsigned)(iy0 + r) < (unsigned)H;
        const unsigned off = rowoff[u] + (unsigned)toff;
        nx0[u] = rows_ld(r_x, (okr && okc0) ? off : kRowsOOB, img);
        nx1[u] = rows_ld(r_x, (okr && okc1) ? off + 4u : kRowsOOB, img);
      }
    }
  };
  auto commit = [&](int pb) {
    uint32_t* dst = patch + pb * PATCH_D + fpc;
#pragma unroll
    for (int u = 0; u < PER_T; ++u) {
      const int R = frow0 + RSTEP * u;
      if (R < NROW) {
        half2v h, l;
        h[0] = (_Float16)nx0[u];
        h[1] = (_Float16)nx1[u];
        l[0] = (_Float16)(nx0[u] - (float)h[0]);
        l[1] = (_Float16)(nx1[u] - (float)h[1]);
        dst[R * ROWD] = __builtin_bit_cast(uint32_t, h);
        if constexpr (!HALF) dst[R * ROWD + LO_D] = __builtin_bit_cast(uint32_t, l);
      }
    }
  };

  // sign words of the PREVIOUS tile: its waves left two 32-bit halves per pixel in LDS; PTH*PTW threads send them as
  // whole 64-bit words
  int prev_n = -1, prev_py0 = 0, prev_px0 = 0, buf = 0, pb = 0;
  auto flush_bits = [&](int b) {
    if (!RAW && prev_n >= 0 && tid < PTH * PTW) {  // wave 0 only
      const int ply = tid / PTW, px = tid - ply * PTW;
      const bool ok = prev_py0 + ply < Hp && prev_px0 + px < Wp;
      const unsigned soff = (unsigned)((prev_n * Hp + prev_py0) * Wp + prev_px0) * 8u;
      const u32x2 word = *reinterpret_cast<const u32x2*>(bits + (b * PTH * PTW + tid) * 2);
      rows_st2(r_P, ok ? flush_lane : kRowsOOB, soff, word);
      rows_st2(r_M, ok ? flush_lane : kRowsOOB, soff, u32x2{0u, 0u});  // nothing is negative after ReLU
    }
  };

#ifdef BNN_ROWS_TIMING
  uint32_t* tcount = reinterpret_cast<uint32_t*>(lds_rows + OFF_TIME) + wave * 16 * 64 + lane;
  for (int k = 0; k < 16; ++k) tcount[k * 64] = 0u;
  uint32_t tlast = (uint32_t)__builtin_readcyclecounter();
#define ROWS_T(k) { const uint32_t tn = (uint32_t)__builtin_readcyclecounter(); tcount[(k) * 64] += tn - tlast; tlast = tn; }
#else
#define ROWS_T(k) {}
#endif
  // position of a tile of the list (divisions: once per segment; inside a segment the walk is ty + 1, tx + 1, n + 1)
  auto locate = [&](int g, int& n, int& tx, int& ty) {
    n = g / (tiles_x * tiles_y);
    const int rem = g - n * tiles_x * tiles_y;
    tx = rem / tiles_y;
    ty = rem - tx * tiles_y;
  };
  int g = wg * seg_len, n = 0, tx = 0, ty = 0;
  bool valid = g < ntiles;  // workgroup-uniform
  if (valid) {
    locate(g, n, tx, ty);
    if (!(BNN_ROWS_ABL & 8)) fetch(n, tx, ty);
  }
  __syncthreads();  // the zero fill is complete
  if (valid) commit(0);
  float carry[2][4];  // BN'd conv row above the next pooled row (row 2*py - 1), kept across tiles of a strip
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) carry[tt][r] = 0.0f;
  [[maybe_unused]] uint32_t kq_run[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) kq_run[ks] = kb[ks];
  for (int it = 0; it < niter && valid; ++it) {
    ROWS_T(13)
    __syncthreads();                   // this tile's patch (buffer pb) is in LDS; nobody reads buffer pb ^ 1 any more
    ROWS_T(0)
    flush_bits(buf ^ 1);
    ROWS_T(14)
    // the next tile: one step down the list, or the first tile of the next segment
    const bool seg_first = it % seg_len == 0, seg_last = (it + 1) % seg_len == 0;
    int ng = g + 1, nn = n, ntx = tx, nty = ty + 1;
    if (seg_last) {
      ng = g + 1 + seg_stride - seg_len;
      if (ng < ntiles) locate(ng, nn, ntx, nty);
    } else if (nty == tiles_y) {
      nty = 0;
      if (++ntx == tiles_x) { ntx = 0; ++nn; }
    }
    const bool more = it + 1 < niter && ng < ntiles;
    if (more && !(BNN_ROWS_ABL & 8)) fetch(nn, ntx, nty);  // global loads fly during the first half of the tile
    ROWS_T(15)
    const int py0 = ty * PTH, px0 = tx * PTW;        // pooled origin
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // conv origin (pool pad 1): conv row t of the tile is row cy0 + t
    // most tiles lie entirely inside the conv output: no range tests there
    const bool interior = cy0 >= 0 && cx0 >= 0 && cy0 + CTH <= Hc && cx0 + 2 * PTW + 1 <= Wc;  // workgroup-uniform
    const bool col_in = (unsigned)(cx0 + 2 * WPW * mg + lcol) < (unsigned)Wc;
    const bool out_live = pool_lane && px0 + WPW * mg + plx < Wp;
    const unsigned out_tile = (unsigned)(((n * COUT + 32 * nh) * Hp + py0) * Wp + px0 + WPW * mg) * 4u;  // wave-uniform
    const unsigned chw4 = (unsigned)(Hp * Wp) * 4u;

    // walks down the strip: advanced by the rows a step has read.  LEAN: ONE register per k-step for the whole kernel
    // (the tile's start = the previous tile's end minus its nine conv rows, in the other patch buffer)
    uint32_t kq_tile[KSTEPS];
    uint32_t (&kq)[KSTEPS] = LEAN ? kq_run : kq_tile;
    if constexpr (!LEAN) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) kq[ks] = kb[ks] + (uint32_t)(pb * PATCH_D * 4);
    }

    half8 fh[RING][2], fl[RING][2];
    // fragments of `rows` conv rows at the walking base of k-step ks into ring slot `slot`
    auto request = [&](int slot, int ks, int rows) {  // constants after unrolling
      lds_word* p = (lds_word*)(uintptr_t)kq[ks];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        if (d >= rows) continue;
        if (BNN_ROWS_ABL & 4) {
          fh[slot][d] = wh[ks][d];
          fl[slot][d] = wl[ks][d];
          continue;
        }
        const bool k16 = K16 && ks == KSTEPS - 1;  // two column pairs per lane
        u32x4 v{0u, 0u, 0u, 0u};
        v[0] = p[d * CONV_ROW_D + 0]; v[1] = p[d * CONV_ROW_D + 1];
        if (!k16) { v[2] = p[d * CONV_ROW_D + 2]; v[3] = p[d * CONV_ROW_D + 3]; }
        fh[slot][d] = __builtin_bit_cast(half8, v);
        if constexpr (!HALF) {
          u32x4 l{0u, 0u, 0u, 0u};
          l[0] = p[d * CONV_ROW_D + LO_D + 0]; l[1] = p[d * CONV_ROW_D + LO_D + 1];
          if (!k16) { l[2] = p[d * CONV_ROW_D + LO_D + 2]; l[3] = p[d * CONV_ROW_D + LO_D + 3]; }
          fl[slot][d] = __builtin_bit_cast(half8, l);
        }
      }
      kq[ks] += (uint32_t)(rows * CONV_ROW_D * 4);
      asm volatile("" : "+v"(kq[ks]));  // keep the walking base: folded into constants, the offsets leave the immediates' range
    };
    f32x4 acc[2][2];
    // the MFMAs of one k-step on `rows` conv rows: product-type major, so that the accumulator chains of a step are
    // independent between two MFMAs on the same one
    auto lo4 = [](half8 v) { return half4{v[0], v[1], v[2], v[3]}; };
    auto multiply = [&](int slot, int ks, int rows) {
      if (K16 && ks == KSTEPS - 1) {  // (constant after unrolling) the same three products, 16 k per instruction
#if defined(__HIP_DEVICE_COMPILE__)
        // hipcc 7.2 may schedule an 8-pass v_mfma_f32_16x16x32_f16 and, in the very next slot, a 4-pass
        // v_mfma_f32_16x16x16_f16 whose SrcC is that result and whose vDst is another register quad, with no wait states
        // between them; gfx950 then delivers two of the four result registers wrong (seen in the last pooled row of a
        // tile: conv rows 6 mod 8 / pooled rows 3 mod 4, channels 4*lg + {0, 1}; tools/experiments/README.md 54).
        // Whether the pair ends up back to back depends on the schedule of the surrounding code, so the window is closed
        // explicitly: every 32-k product is issued before the first 16-k one, 16 slots apart (4 x 16 cycles per tile).
        // tools/mfma_pairs.py + tests/test_isa_cpu.py scan the assembly for the pair.
        asm volatile("s_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
#endif
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            if (d < rows) acc[d][tt] = __builtin_amdgcn_mfma_f32_16x16x16f16(lo4(wh[ks][tt]), lo4(fl[slot][d]), acc[d][tt], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            if (d < rows) acc[d][tt] = __builtin_amdgcn_mfma_f32_16x16x16f16(lo4(wl[ks][tt]), lo4(fh[slot][d]), acc[d][tt], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            if (d < rows) acc[d][tt] = __builtin_amdgcn_mfma_f32_16x16x16f16(lo4(wh[ks][tt]), lo4(fh[slot][d]), acc[d][tt], 0, 0, 0);
        return;
      }
      if constexpr (!HALF && !(BNN_ROWS_ABL & 1)) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            if (d < rows) acc[d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks][tt], fl[slot][d], acc[d][tt], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            if (d < rows) acc[d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ks][tt], fh[slot][d], acc[d][tt], 0, 0, 0);
      }
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          if (d < rows && !(BNN_ROWS_ABL & 1))
            acc[d][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ks][tt], fh[slot][d], acc[d][tt], 0, 0, 0);
    };
    auto clear = [&]() {
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[d][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // BN of accumulator row d = conv row t of the tile (the ReLU is applied after the max-pool: max and relu commute
    // exactly); positions outside the conv output are MaxPool padding: 0 never beats a ReLU output
    auto bn_row = [&](float (&y)[2][4], int d, int t) {
      using f32x2 = __attribute__((ext_vector_type(2))) float;
      if constexpr (LEAN) {  // (two ds_read_b128 per channel tile: 16 registers that are not alive across the MFMA stream)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const f32x4 a4 = *(lds_f4*)(uintptr_t)(bn_lane + 64u * tt), b4 = *(lds_f4*)(uintptr_t)(bn_lane + 64u * tt + 256u);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ba[tt][r] = a4[r];
            bb[tt][r] = b4[r];
          }
        }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {  // two channels per v_pk_fma_f32: the same fma per element
          const f32x2 v = __builtin_elementwise_fma(f32x2{acc[d][tt][r], acc[d][tt][r + 1]}, f32x2{ba[tt][r], ba[tt][r + 1]},
                                                    f32x2{bb[tt][r], bb[tt][r + 1]});
          y[tt][r] = v.x;
          y[tt][r + 1] = v.y;
        }
      if (!interior) {
        const bool in = col_in && (unsigned)(cy0 + t) < (unsigned)Hc;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) y[tt][r] = in ? y[tt][r] : 0.0f;
      }
    };
    // pooled row q of the tile from the kept row and the two rows in the accumulators: pooling, stores, sign bits
    auto finish = [&](int q) {
      if constexpr (RAW) {
        // conv rows 2q + 1, 2q + 2 of the tile = rows 8*ty + 2q, + 1 of the image (row 0 of a tile is the previous
        // tile's row 8: stored there), columns 28*tx + 14*mg .. + 13
        const bool col_ok = raw_col && 2 * px0 + 2 * WPW * mg + lcol - 1 < Wc;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int row = 2 * py0 + 2 * q + d;
          if (col_ok && row < Hc) {
            unsigned soff = (unsigned)(((n * COUT + 32 * nh) * Hc + row) * Wc + 2 * px0) * 4u;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                rows_st(r_out, raw_lane, soff, acc[d][tt][r]);
                soff += (r == 3 ? 13u : 1u) * (unsigned)(Hc * Wc) * 4u;
                asm volatile("" : "+s"(soff));
              }
          }
        }
        return;
      }
      float y0[2][4], y1[2][4];
      bn_row(y0, 0, 2 * q + 1);
      bn_row(y1, 1, 2 * q + 2);
      float t[8], v8[8];
      {
        float t0[4], t1[4];
        pool_rows4(t0, carry[0], y0[0], y1[0]);
        pool_rows4(t1, carry[1], y0[1], y1[1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[r] = t0[r];
          t[4 + r] = t1[r];
          carry[0][r] = y1[0][r];
          carry[1][r] = y1[1][r];
        }
      }
      pool_cols8(v8, t);  // v8[4*tt + r]: the pooled value of channel 16*tt + r (+ 4*lg) on lanes 8 .. 14
      // fp32 stores under an EXEC mask of the 28 lanes that hold a pooled value (rows below / columns right of the
      // image dropped): the texture addresser walks the active lanes of a store, not all 64
      if (out_live && py0 + q < Hp && !(BNN_ROWS_ABL & 16)) {
        // wave-uniform store offset of channel 16*tt + r, walked upwards (kept as ONE running SGPR: as 32 loop
        // invariants they are spilled to lanes and read back in front of every store)
        unsigned soff = out_tile + (unsigned)(q * Wp) * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          rows_st(r_out, out_lane, soff, v8[j]);
          soff += ((j & 3) == 3 ? 13u : 1u) * chw4;
          asm volatile("" : "+s"(soff));
        }
      }
      uint32_t z0, z1;
      if constexpr (AFF) {   // (v8[4 * tt + r]: channel 32 * nh + 16 * tt + 4 * lg + r)
        float u8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) u8[j] = fmaf(v8[j], pka[j >> 2][j & 3], pkb[j >> 2][j & 3]);
        sign_bits8(z0, z1, u8);
      } else {
        sign_bits8(z0, z1, v8);
      }
      if (P != nullptr && !(BNN_ROWS_ABL & 32)) {
        // channel 16*tt + 4*lg + r is bit 16*tt + 4*lg + r of this wave's half of the pixel's word
        const uint32_t wd = or_rows((z0 | (z1 << 16)) << (4 * lg));
        if (lg == 0 && pool_lane) bits[((buf * PTH + q) * PTW + WPW * mg + plx) * 2 + nh] = wd;
      }
    };

    // The conv row above the tile: kept in registers from the tile above (same strip, previous iteration); MaxPool
    // padding above the image; computed here (2 accumulator chains only) at the start of a chunk inside a strip.
    if (RAW) {
      // (no pooling: the row above the tile is not needed)
    } else if (cy0 < 0) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) carry[tt][r] = 0.0f;
    } else if (seg_first) {
      clear();
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const uint32_t keep = kq[ks];
        request(0, ks, 1);
        kq[ks] = keep;
        multiply(0, ks, 1);
      }
      bn_row(carry, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {  // the stream starts at conv row 1
      kq[ks] += (uint32_t)(CONV_ROW_D * 4);
      // opaque: folded into the reads' offsets instead, this row moves every fragment of the SECOND conv row of a step
      // beyond ds_read2_b32's 8-bit offsets (four v_add_u32 per k-step and pooled row for their addresses)
      asm volatile("" : "+v"(kq[ks]));
    }
    ROWS_T(1)
    // The tile is a stream of 24 steps (pooled row, k-step) of 12 MFMAs on two conv rows.  The B fragments of step
    // s + AHEAD are requested BEFORE the MFMAs of step s are issued (ring of AHEAD + 1 fragment sets): with two waves
    // per SIMD nobody else covers the LDS latency.
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) request(s % RING, s % KSTEPS, 2);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + AHEAD < NSTEP) request((s + AHEAD) % RING, (s + AHEAD) % KSTEPS, 2);
      __builtin_amdgcn_sched_barrier(0);  // requests first: sunk next to their use they are exposed again
      const int ks = s % KSTEPS;
      if (ks == 0) clear();
      multiply(s % RING, ks, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == KSTEPS - 1) ROWS_T(2 + s / KSTEPS)
      // next patch: registers -> the other LDS buffer.  BEFORE this row's stores: the loads' counted wait then has the
      // eight stores of the first pooled row behind them, not sixteen (the compiler counts as if none were issued)
      if (s == 2 * KSTEPS - 1 && more && !(BNN_ROWS_ABL & 8)) commit(pb ^ 1);
      if (ks == KSTEPS - 1 && !(BNN_ROWS_ABL & 2)) finish(s / KSTEPS);
      if (ks == KSTEPS - 1) ROWS_T(7 + s / KSTEPS)
      if (s == 2 * KSTEPS - 1) ROWS_T(12)
    }
    if constexpr (LEAN) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        kq_run[ks] += (uint32_t)((pb ? -1 : 1) * PATCH_D * 4 - CTH * CONV_ROW_D * 4);
    }
    prev_n = n;
    prev_py0 = py0;
    prev_px0 = px0;
    buf ^= 1;
    pb ^= 1;
    g = ng;
    n = nn;
    tx = ntx;
    ty = nty;
    valid = more;
  }
  __syncthreads();
  flush_bits(buf ^ 1);
#ifdef BNN_ROWS_TIMING
  __syncthreads();
  if (lane == 0 && M)
    for (int k = 0; k < 16; ++k) M[((size_t)blockIdx.x * NW + wave) * 16 + k] = tcount[k * 64];
#endif
}

template <bool HALF, bool RAW = false, bool AFF = false>
static int launch_stem_rows_t(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H,
                              int W, float* out, uint64_t* P, uint64_t* M, hipStream_t stream,
                              const float* pk_a = nullptr, const float* pk_b = nullptr) {
  using namespace stemr;
  const int Hc = (H + 6 - KS) / 2 + 1, Wc = (W + 6 - KS) / 2 + 1;
  const int Hp = (Hc + 2 - 3) / 2 + 1, Wp = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Hp + PTH - 1) / PTH, tiles_x = (Wp + PTW - 1) / PTW;
  const long long ntiles = (long long)N * tiles_y * tiles_x;
  const int cus = current_device_cus();
#ifndef BNN_ROWS_WG_PER_CU  // workgroups per CU (2: two waves per SIMD cover each other's LDS latency)
#define BNN_ROWS_WG_PER_CU (8 / NW)
#endif
  const long long want = (long long)cus * BNN_ROWS_WG_PER_CU;  // 8 waves (two workgroups) per CU
  const unsigned grid = (unsigned)(ntiles < want ? ((ntiles + 7) / 8 * 8) : want);
  // segments (see the kernel): whole strips round-robin when every workgroup gets at least one, else one chunk each
  const long long strips = (long long)N * tiles_x;
  const int seg_len = strips >= grid ? tiles_y : (int)((ntiles + grid - 1) / grid);
  const int nseg = strips >= grid ? (int)((strips + grid - 1) / grid) : 1;
  // per device and per kernel, so it is set on every launch (no mutable global state in a re-entrant API)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_rows_kernel<HALF, RAW, AFF>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  // byte sizes of the streams (the C-ABI caps every tensor below 2^32 bytes)
  const unsigned out_bytes = (unsigned)((long long)N * COUT * (RAW ? (long long)Hc * Wc : (long long)Hp * Wp) * 4);
  const unsigned plane_bytes = (unsigned)((long long)N * Hp * Wp * 8);
  const unsigned x_bytes = (unsigned)((long long)N * CIN * H * W * 4);
  hipLaunchKernelGGL((stem_rows_kernel<HALF, RAW, AFF>), dim3(grid), dim3(NT), LDS_BYTES, stream, x, w, bn_a, bn_b, N, H, W,
                     Hc, Wc, Hp, Wp, tiles_y, tiles_x, seg_len, nseg, x_bytes, out, P, M, out_bytes, plane_bytes, pk_a, pk_b);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_stem_rows(const float* x, const float* w, const float* bn_a, const float* bn_b, int N, int H, int W,
                     int half, float* out, uint64_t* P, uint64_t* M, hipStream_t stream) {
  return half ? launch_stem_rows_t<true>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream)
              : launch_stem_rows_t<false>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream);
}

// The sign planes behind a per-channel affine of the output (see AFF above).
int launch_stem_rows_aff(const float* x, const float* w, const float* bn_a, const float* bn_b, const float* pk_a,
                         const float* pk_b, int N, int H, int W, int half, float* out, uint64_t* P, uint64_t* M,
                         hipStream_t stream) {
  return half ? launch_stem_rows_t<true, false, true>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream, pk_a, pk_b)
              : launch_stem_rows_t<false, false, true>(x, w, bn_a, bn_b, N, H, W, out, P, M, stream, pk_a, pk_b);
}

// The convolution alone (training forward): fp32 [N, 64, Hc, Wc].
int launch_stem_conv(const float* x, const float* w, int N, int H, int W, int half, float* out, hipStream_t stream) {
  return half ? launch_stem_rows_t<true, true>(x, w, nullptr, nullptr, N, H, W, out, nullptr, nullptr, stream)
              : launch_stem_rows_t<false, true>(x, w, nullptr, nullptr, N, H, W, out, nullptr, nullptr, stream);
}

}  // namespace bnn
