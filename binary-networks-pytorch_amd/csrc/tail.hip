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
