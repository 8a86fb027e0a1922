// stem_wgrad.hip — weight gradient of the real-valued stem convolution (training backward of bnn/models/resnet.py:150,
// `x = self.conv1(x)`: conv 7x7 / stride 2 / pad 3, 3 -> 64, no bias):
//
//     dW[o][c][ky][kx] = sum over n, y, x of  dy[n][o][y][x] * in[n][c][2y + ky - 3][2x + kx - 3]      (zero padding)
//
// The input is data (no input gradient), so this is the whole backward of the layer.  The library evaluates it as an
// implicit GEMM on NHWC copies (two transposes of the 822 MB gradient + the GEMM: 1.46 ms at batch 256).  Here: a GEMM
// M = 64 channels, N = 21 (c, ky) rows x 8 kx slots (kx = 7 and row 21 are padding: computed, never written out),
// K = every conv pixel, on v_mfma_f32_16x16x4_f32 — fp32 products, fp32 accumulation, operands ONE float per lane, so
// neither operand needs a layout: no transposes, no fp16 split, no alignment rules for the stride-2 patch reads.
//
//  * A workgroup (4 waves) owns a band of RB conv rows of one image at a time (grid-stride over bands); wave w owns
//    channels 16w .. 16w + 15 and all 11 column tiles of the result (44 accumulator registers, kept for the whole launch).
//  * B operand = the input patch of the band in LDS as plain fp32 rows [c][2 RB + 5 rows][3 + W + pad] (zero padding
//    staged as zeros); lane (j = lane & 15, k = lane >> 4) reads patch[c][2 yl + ky][2 p + kx] for its (c, ky, kx) and
//    pixel p.  Row stride = 32 mod 64 dwords: the 64 addresses of a read (32 consecutive columns of two patch rows) fall
//    into 64 different banks.
//  * A operand = dy straight from global memory: a chunk of 16 conv pixels of a row is one float4 per lane (channel =
//    lane & 15, pixels 4 (lane >> 4) .. + 3) = 64 contiguous bytes per channel; element t of the float4 feeds MFMA step t
//    of the chunk, i.e. pixel(k, t) = 16 q + 4 k + t — any assignment of a chunk's pixels to (k, step) is a valid
//    reduction order as long as B reads the same pixel.  The next chunk's float4 is requested before this chunk's MFMAs.
//  * Every workgroup writes its partial [64][176] sums once; a second kernel adds the partials in index order in fp64
//    (deterministic: no atomics) and drops the padding columns.
#include <algorithm>

#include "bnn_dev.h"

namespace bnn {
namespace swg {
constexpr int NT = 256, NW = NT / 64;
constexpr int COUT = 64, CIN = 3, KS = 7;
constexpr int RB = 4;                    // conv rows per band
constexpr int PR = 2 * RB + 5;           // patch rows per channel
constexpr int KT = 11;                   // 16-wide column tiles: 21 (c, ky) rows x 8 kx slots = 168 -> 176
constexpr int KPAD = 16 * KT;
constexpr int MAX_LDS = 64 * 1024;       // patch bytes a workgroup may take (three workgroups per CU at 224 x 224)
}  // namespace swg

using swg_f32x4 = __attribute__((ext_vector_type(4))) float;

// dwords per patch row: columns -3 .. 2 * (16 * ceil(Wc / 16)) + 4 of the input, rounded up to 32 mod 64
static inline int swg_row_stride(int Wc) {
  const int need = 2 * ((Wc + 15) / 16 * 16) + 8;
  int s = (need + 63) / 64 * 64 + 32;
  if (s - 64 >= need) s -= 64;
  return s;
}

template <bool VEC>
__global__ __launch_bounds__(swg::NT) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             int N, int H, int W, int Hc, int Wc, int bands_y,
                                                             int row_stride, float* __restrict__ work) {
  using namespace swg;
  extern __shared__ __attribute__((aligned(16))) float patch[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  // column tile T of the result: (c, ky) rows 2T and 2T + 1 (row 21 reads row 20 again; dropped by the reduction)
  int base[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    int R = 2 * t + (li >> 3);
    if (R > CIN * KS - 1) R = CIN * KS - 1;
    const int c = R / KS, ky = R - c * KS;
    base[t] = (c * PR + ky) * row_stride + (li & 7) + 8 * lg;  // + 2 * (4 lg) columns of this lane's pixels
  }
  swg_f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = swg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int chunks = (Wc + 15) / 16;

  auto load_a = [&](const float* row, int q) -> swg_f32x4 {  // dy[channel li of this wave][y][16 q + 4 lg .. + 3]
    const int p0 = 16 * q + 4 * lg;
    if constexpr (VEC) {
      return *reinterpret_cast<const swg_f32x4*>(row + p0);
    } else {
      swg_f32x4 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = p0 + e < Wc ? row[p0 + e] : 0.0f;
      return a;
    }
  };

  for (int b = blockIdx.x; b < N * bands_y; b += gridDim.x) {
    const int n = b / bands_y, y0 = (b - n * bands_y) * RB;
    __syncthreads();  // the previous band's reads are done
    // ---- the band's patch: input rows 2 y0 - 3 .. 2 y0 + 2 RB + 1, columns -3 .. row_stride - 4 (zeros outside)
    // (a wave takes rows wave, wave + 4, ..; the loads of TWO rows are issued before the first LDS write: one element
    // at a time the staging is ~50 dependent global-memory latencies per band, several times the band's matrix work)
    for (int r0 = wave; r0 < CIN * PR; r0 += 2 * NW) {
      float v[2][8];
      for (int cb = 0; cb < row_stride; cb += 8 * 64) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = r0 + h * NW;
          const int c = r / PR, pr = r - c * PR;
          const int iy = 2 * y0 - 3 + pr;
          const bool row_in = r < CIN * PR && (unsigned)iy < (unsigned)H;
          const float* src = x + ((size_t)(n * CIN + (row_in ? c : 0)) * H + (row_in ? iy : 0)) * W;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ix = cb + lane + 64 * j - 3;
            v[h][j] = (row_in && (unsigned)ix < (unsigned)W) ? src[ix] : 0.0f;
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = r0 + h * NW;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int col = cb + lane + 64 * j;
            if (r < CIN * PR && col < row_stride) patch[r * row_stride + col] = v[h][j];
          }
        }
      }
    }
    __syncthreads();
    const int rows = min(RB, Hc - y0);
    const float* drow = dy + ((size_t)(n * COUT + 16 * wave + li) * Hc + y0) * Wc;
    // Chunks of the band in (row, 16-pixel chunk) order, two per iteration with two float4 registers in turn: the
    // float4 of chunk i + 1 is requested before the MFMAs of chunk i (a single register pair carried around the loop
    // ends up loaded at the top of its own iteration and waited for at once).  Inside a chunk all 22 LDS reads are
    // issued before the first MFMA (counted waits), not one in front of each.
    const int total = rows * chunks;
    int yl = 0, q = 0;  // position of the chunk the NEXT request is for
    auto request = [&]() -> swg_f32x4 {
      const swg_f32x4 a = load_a(drow + (size_t)yl * Wc, q);
      return a;
    };
    auto advance = [&]() {
      if (++q == chunks) { q = 0; ++yl; }
    };
    auto chunk = [&](const swg_f32x4& a, int cyl, int cq) {
      const int off = 2 * cyl * row_stride + 32 * cq;
      float bv[4][KT];
#pragma unroll
      for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t][k] = patch[base[k] + off + 2 * t];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bv[t][k], acc[k], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // (branch-free pair body: a request under a condition is sunk into the block of its use, i.e. behind the MFMAs it
    // was meant to run under; a request past the end of the band reads the last chunk again)
    auto advance_clamped = [&](int done) {
      if (done + 1 < total) advance();
    };
    swg_f32x4 a0 = request(), a1;
    int i = 0;
    for (; i + 1 < total; i += 2) {
      const int yl0 = yl, q0 = q;
      advance();                         // chunk i + 1 exists
      a1 = request();
      const int yl1 = yl, q1 = q;
      __builtin_amdgcn_sched_barrier(0);
      chunk(a0, yl0, q0);
      advance_clamped(i + 1);
      a0 = request();                    // chunk i + 2 (or i + 1 again)
      __builtin_amdgcn_sched_barrier(0);
      chunk(a1, yl1, q1);
    }
    if (i < total) chunk(a0, yl, q);     // odd count: the last request was for this chunk
  }
  // ---- partial sums of this workgroup: D[i = 4 lg + r][j = li] of wave w, tile k -> work[block][16 w + i][16 k + j]
  float* out = work + ((size_t)blockIdx.x * COUT + 16 * wave + 4 * lg) * KPAD + li;
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)r * KPAD + 16 * k] = acc[k][r];
}

// dW[o][c][ky][kx] = sum over the workgroups' partials in fp64: four slices of the list in parallel (each in index order),
// the four slice sums added in slice order — a fixed order, so the same bits on every run
__global__ __launch_bounds__(4 * 192) void stem_wgrad_reduce_kernel(const float* __restrict__ work, int G, float* __restrict__ dw) {
  using namespace swg;
  __shared__ double part[4][192];
  const int o = blockIdx.x, j = threadIdx.x % 192, sl = threadIdx.x / 192;
  const int per = (G + 3) / 4, g0 = sl * per, g1 = min(G, g0 + per);
  double s = 0.0;
  if (j < KPAD) {
    const float* p = work + (size_t)o * KPAD + j;
#pragma unroll 8
    for (int g = g0; g < g1; ++g) s += (double)p[(size_t)g * COUT * KPAD];
  }
  part[sl][j] = s;
  __syncthreads();
  const int R = j >> 3, kx = j & 7;
  if (sl == 0 && R < CIN * KS && kx < KS)
    dw[(size_t)o * CIN * KS * KS + R * KS + kx] = (float)(((part[0][j] + part[1][j]) + part[2][j]) + part[3][j]);
}

size_t stem_wgrad_lds_bytes(int H, int W);

// workgroups of a launch: one per band, at most what fits beside each other (three per CU by LDS at 224 x 224)
static int swg_grid(int N, int H, int W) {
  using namespace swg;
  const int Hc = (H - 1) / 2 + 1;
  const long long bands = (long long)N * ((Hc + RB - 1) / RB);
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(3, (size_t)160 * 1024 / stem_wgrad_lds_bytes(H, W)));
  return (int)std::min<long long>(bands, (long long)current_device_cus() * per_cu);
}

size_t stem_wgrad_lds_bytes(int H, int W) {
  using namespace swg;
  (void)H;
  const int Wc = (W - 1) / 2 + 1;
  return (size_t)CIN * PR * swg_row_stride(Wc) * sizeof(float);
}

bool stem_wgrad_supported(int H, int W) { return H > 0 && W > 0 && stem_wgrad_lds_bytes(H, W) <= (size_t)swg::MAX_LDS; }

// floats of workspace: one [64][176] slab per workgroup the launch will use
size_t stem_wgrad_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || !stem_wgrad_supported(H, W)) return 0;
  return (size_t)swg_grid(N, H, W) * swg::COUT * swg::KPAD * sizeof(float);
}

int launch_stem_wgrad(const float* x, const float* dy, int N, int H, int W, float* work, float* dw, hipStream_t stream) {
  using namespace swg;
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
  const int bands_y = (Hc + RB - 1) / RB;
  const int rs = swg_row_stride(Wc);
  const int G = swg_grid(N, H, W);
  const size_t lds = stem_wgrad_lds_bytes(H, W);
  const bool vec = Wc % 16 == 0 && ((uintptr_t)dy & 15u) == 0;
  auto kern = vec ? stem_wgrad_kernel<true> : stem_wgrad_kernel<false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess)
    return BNN_HIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(G), dim3(NT), lds, stream, x, dy, N, H, W, Hc, Wc, bands_y, rs, work);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(COUT), dim3(4 * 192), 0, stream, work, G, dw);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
