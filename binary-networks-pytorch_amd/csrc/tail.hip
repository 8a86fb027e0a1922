// tail.hip — the real-valued LAST layer of the reference's ResNets as one kernel:
//     x = avgpool(x); x = torch.flatten(x, 1); x = fc(x)          (bnn/models/resnet.py:160-164)
// i.e. AdaptiveAvgPool2d((1,1)) over the HW positions of every channel, then Linear(C -> O) with bias.
// At ResNet-18 size (C = 512, HW = 49, O = 1000, batch 256) this is 25.7 MB of input and 0.13 G MAC:
// launch- and latency-bound work that three library kernels (mean-reduce, GEMM, bias/copies) spent
// ~100 us on; fused it is one pass with the means kept in LDS.
//
// Work decomposition: a workgroup owns IMG = 4 images x 256 output features (ResNet-18, batch 256: 64 x 4 = 256
// workgroups, one per CU).  The kernel is latency-bound, so both phases are written for loads in flight:
//   phase 1: thread (i, c) sums the HW contiguous values of channel c of image i — all HW loads of an item are
//            issued before the first add when HW <= 64 (the 7x7 = 49 of every ResNet: one memory round trip per
//            item instead of seven), fp32, index order, one division by HW (the order of ATen's CPU
//            adaptive_avg_pool2d) -> m[c][i] in LDS;
//   phase 2: thread o walks k = 0..C-1, 32 iterations in flight (16: 29 us at batch 256, 32: 23 us; computing the means
//            once in a separate wide launch and starting from them: 23.5 us — not adopted): one coalesced load of Wt[k][o] (the weight is
//            passed transposed, [C][O], so that the 64 lanes of a wave read 256 contiguous bytes), one broadcast
//            ds_read_b128 of m[k][0..3], four fmaf.  The weight (2 MB at ResNet-18 size) comes from L2.
#include "bnn_dev.h"

namespace bnn {

namespace tail {
constexpr int IMG = 4;
constexpr int NT = 256;
}  // namespace tail

// HWC > 0: HW is this compile-time constant (49 for every 224x224 ResNet); 0: run-time HW, rolled loop.
template <int HWC>
__global__ __launch_bounds__(tail::NT) void avgpool_fc_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ wt,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out, int N, int C,
                                                               int HW_rt, int O) {
  using namespace tail;
  const int HW = HWC > 0 ? HWC : HW_rt;
  extern __shared__ __attribute__((aligned(16))) float m[];  // [C][IMG]
  const int n0 = blockIdx.x * IMG;
  const int tid = threadIdx.x;
  const float hw = (float)HW;
  for (int idx = tid; idx < C * IMG; idx += NT) {
    const int i = idx / C, c = idx - i * C;
    float s = 0.0f;
    if (n0 + i < N) {
      const float* p = x + ((size_t)(n0 + i) * C + c) * HW;
      if constexpr (HWC > 0) {
        float v[HWC];
#pragma unroll
        for (int q = 0; q < HWC; ++q) v[q] = p[q];
#pragma unroll
        for (int q = 0; q < HWC; ++q) s += v[q];
      } else {
#pragma unroll 8
        for (int q = 0; q < HW; ++q) s += p[q];
      }
      s = s / hw;
    }
    m[c * IMG + i] = s;
  }
  __syncthreads();
  const int o = blockIdx.y * NT + tid;
  if (o >= O) return;
  float acc[IMG];
#pragma unroll
  for (int i = 0; i < IMG; ++i) acc[i] = 0.0f;
  const float* wp = wt + o;
#pragma unroll 32
  for (int k = 0; k < C; ++k) {
    const float w = wp[(size_t)k * O];
    const float4 a = *reinterpret_cast<const float4*>(&m[k * IMG]);
    acc[0] = fmaf(a.x, w, acc[0]);
    acc[1] = fmaf(a.y, w, acc[1]);
    acc[2] = fmaf(a.z, w, acc[2]);
    acc[3] = fmaf(a.w, w, acc[3]);
  }
  const float bv = bias ? bias[o] : 0.0f;
#pragma unroll
  for (int i = 0; i < IMG; ++i)
    if (n0 + i < N) out[(size_t)(n0 + i) * O + o] = acc[i] + bv;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the head as TWO launches through a workspace (bnn_hip_avgpool_fc_ws_f32).  The one-kernel form above is
// latency-bound by construction: a workgroup needs the means of its images before its first product, every workgroup
// of an image group recomputes them (4x the input traffic), and 512 dependent weight loads per thread follow with one
// wave per SIMD.  Split, each half is a plain streaming kernel:
//   avgpool_rows_kernel: 256 (image, channel) rows per workgroup; their HW * 256 contiguous floats are fetched with
//       coalesced loads (all in flight at once), staged in LDS, and every thread sums ITS row from LDS in index order
//       (row stride HW = 49 words: odd, conflict-free) -> the same fp32 value as above.  Means are written
//       group-major, mt[n / 16][c][n % 16], so that an fc workgroup's operand is one contiguous block.
//   fc_ws_kernel: workgroup = 16 images x 64 outputs, 8 waves = 8 k-segments; a lane owns one output and 16
//       accumulators (images), per k one coalesced weight load (the weight is read once per workgroup, 32 MB of L2
//       traffic per call instead of 128), four broadcast ds_read_b128 of the 16 means, eight v_pk_fma_f32 (measured:
//       12 us inside a forward — the broadcast reads are 16 k cycles of LDS pipe per workgroup; the means through the
//       scalar cache instead, one s_load_dwordx16 per k feeding v_pk_fma_f32 as its scalar pair: 36 us, the loads
//       serialise on lgkmcnt(0); the means in vector registers, lane l holding those of k = kb + l, broadcast per step
//       with v_readlane_b32 into the scalar pair: 12 us again — the reads are not what binds; neither kept).  The
//       partial sums of the segments meet in LDS and are added in segment order, then the bias: the summation tree of
//       an output depends on C only — not on the batch size nor on the image's position in the batch.
namespace tail2 {
constexpr int ROWS = 256;        // rows (= threads) per avgpool workgroup
constexpr int IMG = 16, OT = 64, KSEG = 8, NT = KSEG * 64;
constexpr int KU = 64;           // weight loads in flight per wave (ResNet-18: a wave's whole k-segment)
}  // namespace tail2

template <int HWC>
__global__ __launch_bounds__(tail2::ROWS) void avgpool_rows_kernel(const float* __restrict__ x, float* __restrict__ mt,
                                                                    int N, int C, int HW_rt) {
  using namespace tail2;
  const int HW = HWC > 0 ? HWC : HW_rt;
  extern __shared__ __attribute__((aligned(16))) float stage[];  // [ROWS * HW]
  const long long rows = (long long)N * C;
  const long long r0 = (long long)blockIdx.x * ROWS;
  const int tid = threadIdx.x;
  const long long e0 = r0 * HW, etot = rows * HW;
  float s = 0.0f;
  if constexpr (HWC > 0) {
    float v[HWC];
#pragma unroll
    for (int j = 0; j < HWC; ++j) {  // element e0 + tid + ROWS * j: a wave reads 256 contiguous bytes per load
      const long long e = e0 + tid + (long long)ROWS * j;
      v[j] = e < etot ? x[e] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < HWC; ++j) stage[tid + ROWS * j] = v[j];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < HWC; ++q) s += stage[tid * HWC + q];
  } else {
    if (r0 + tid < rows) {
      const float* p = x + (r0 + tid) * HW;
#pragma unroll 8
      for (int q = 0; q < HW; ++q) s += p[q];
    }
  }
  // rows of the images that pad the last group of IMG (the grid covers them): zeros — fc_ws_kernel stages whole groups
  const long long r = r0 + tid;
  const long long rows_pad = (((long long)N + IMG - 1) / IMG) * IMG * C;
  if (r < rows_pad) {
    const int n = (int)(r / C), c = (int)(r - (long long)n * C);
    mt[((size_t)(n / IMG) * C + c) * IMG + (n % IMG)] = r < rows ? s / (float)HW : 0.0f;
  }
}

__global__ __launch_bounds__(tail2::NT) void fc_ws_kernel(const float* __restrict__ mt, const float* __restrict__ wt,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int N, int C, int O) {
  using namespace tail2;
  using f2 = __attribute__((ext_vector_type(2))) float;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // means [C][IMG]; afterwards partial sums [KSEG][IMG][OT]
  const int g = blockIdx.x, o0 = blockIdx.y * OT;
  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int o = o0 + lane;
  const float* wp = wt + (o < O ? o : O - 1);  // lanes past the last output recompute it and store nothing
  const int klen = (C + KSEG - 1) / KSEG;
  const int k0 = seg * klen, k1 = min(C, k0 + klen);
  // the first KU weights of the segment are requested BEFORE the means are staged: they depend on nothing in LDS, and
  // at ResNet-18 size (C = 512: 64 values per segment) they are the whole segment — one memory round trip
  float w[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) w[u] = k0 < k1 ? wp[(size_t)min(k0 + u, k1 - 1) * O] : 0.0f;
  {  // the group's means: one contiguous block of C * IMG floats
    const float4* src = reinterpret_cast<const float4*>(mt + (size_t)g * C * IMG);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < C * IMG / 4; i += NT) dst[i] = src[i];
  }
  __syncthreads();
  f2 acc[IMG / 2];
#pragma unroll
  for (int i = 0; i < IMG / 2; ++i) acc[i] = f2{0.0f, 0.0f};
  for (int kb = k0; kb < k1; kb += KU) {
    if (kb > k0) {
#pragma unroll
      for (int u = 0; u < KU; ++u) w[u] = wp[(size_t)min(kb + u, k1 - 1) * O];
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (kb + u < k1) {  // wave-uniform
        const float4* m4 = reinterpret_cast<const float4*>(&lds[(kb + u) * IMG]);
        const f2 ww = f2{w[u], w[u]};
#pragma unroll
        for (int i = 0; i < IMG / 4; ++i) {
          const float4 a = m4[i];
          acc[2 * i] = __builtin_elementwise_fma(f2{a.x, a.y}, ww, acc[2 * i]);
          acc[2 * i + 1] = __builtin_elementwise_fma(f2{a.z, a.w}, ww, acc[2 * i + 1]);
        }
      }
    }
  }
  __syncthreads();  // everybody is done with the means
#pragma unroll
  for (int i = 0; i < IMG / 2; ++i) {
    lds[(seg * IMG + 2 * i) * OT + lane] = acc[i].x;
    lds[(seg * IMG + 2 * i + 1) * OT + lane] = acc[i].y;
  }
  __syncthreads();
  const float bv = (bias && o < O) ? bias[o] : 0.0f;
  for (int i = seg; i < IMG; i += KSEG) {  // wave `seg` finishes images seg, seg + 8
    float y = lds[i * OT + lane];
#pragma unroll
    for (int sg = 1; sg < KSEG; ++sg) y += lds[(sg * IMG + i) * OT + lane];
    const int n = g * IMG + i;
    if (n < N && o < O) out[(size_t)n * O + o] = y + bv;
  }
}


size_t avgpool_fc_workspace_bytes(int N, int C) {  // (64-bit, saturating: N and C are unchecked caller values here)
  const unsigned long long groups = ((unsigned long long)N + tail2::IMG - 1) / tail2::IMG;
  const unsigned long long elems = groups * (unsigned long long)C;  // < 2^58
  return elems > (1ull << 56) ? ~(size_t)0 : (size_t)(elems * tail2::IMG * sizeof(float));
}

// Whether the two-launch head covers the shape (else the caller runs the one-kernel form).
bool avgpool_fc_ws_supported(int C, int HW) {
  const size_t fc_lds = (size_t)C * tail2::IMG * sizeof(float);
  return HW >= 1 && fc_lds <= (size_t)kMaxDynamicLds - 1024;
}

int launch_avgpool_fc_ws(const float* x, const float* wt, const float* bias, float* out, float* ws, int N, int C,
                         int HW, int O, hipStream_t stream) {
  using namespace tail2;
  const long long rows = (((long long)N + IMG - 1) / IMG) * IMG * C;  // incl. the pad images of the last group (zeros)
  const unsigned gridA = (unsigned)((rows + ROWS - 1) / ROWS);
  if (HW == 49) {
    hipLaunchKernelGGL(avgpool_rows_kernel<49>, dim3(gridA), dim3(ROWS), ROWS * 49 * sizeof(float), stream, x, ws, N, C,
                       HW);
  } else {
    hipLaunchKernelGGL(avgpool_rows_kernel<0>, dim3(gridA), dim3(ROWS), 0, stream, x, ws, N, C, HW);
  }
  if (hipGetLastError() != hipSuccess) return BNN_HIP_ERR_LAUNCH;
  const size_t part = (size_t)KSEG * IMG * OT * sizeof(float);
  size_t lds = (size_t)C * IMG * sizeof(float);
  if (lds < part) lds = part;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(fc_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          kMaxDynamicLds) != hipSuccess)
    return BNN_HIP_ERR_UNSUPPORTED;
  const dim3 gridB((N + IMG - 1) / IMG, (O + OT - 1) / OT);
  hipLaunchKernelGGL(fc_ws_kernel, gridB, dim3(NT), lds, stream, ws, wt, bias, out, N, C, O);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_avgpool_fc(const float* x, const float* wt, const float* bias, float* out, int N, int C, int HW,
                      int O, hipStream_t stream) {
  using namespace tail;
  const size_t lds = (size_t)C * IMG * sizeof(float);
  if (lds > 160 * 1024) return BNN_HIP_ERR_UNSUPPORTED;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(avgpool_fc_kernel<49>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(avgpool_fc_kernel<0>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return BNN_HIP_ERR_UNSUPPORTED;
  }
  const dim3 grid((N + IMG - 1) / IMG, (O + NT - 1) / NT);
  if (HW == 49)
    hipLaunchKernelGGL(avgpool_fc_kernel<49>, grid, dim3(NT), lds, stream, x, wt, bias, out, N, C, HW, O);
  else
    hipLaunchKernelGGL(avgpool_fc_kernel<0>, grid, dim3(NT), lds, stream, x, wt, bias, out, N, C, HW, O);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
