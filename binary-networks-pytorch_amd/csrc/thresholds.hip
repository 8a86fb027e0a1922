// thresholds.hip — BatchNorm + ReLU + sign() of a binary conv folded into an integer test on the popcount result.
//
// For the conv1-type layer of a residual block (bnn/models/layers/res_block.py:40-56: conv -> bn -> relu, the next
// binary conv binarises the result) the only thing that leaves the epilogue is the bit
//     P[o] = ( fmaf( fmaf(alpha[o], dot, bias[o]) [* scale[o]], bn_a[o], bn_b[o] ) > 0 ),      dot in [-K, K] integer.
// Every step is a monotone function of `dot` (rounding is monotone), so the set of dots whose bit is 1 is an INTERVAL.
// One thread per channel finds its ends by bisection, evaluating exactly the float operations of the conv epilogue
// (csrc/bconv.hip: epilogue()); the conv kernel then needs one subtract and one unsigned compare per channel and pixel
// instead of int->float, two fmas and a float compare — same bits by construction (tests/test_gpu_fused.py).
//     thr[2o] = lo, thr[2o+1] = span:   P  <=>  (unsigned)(dot - lo) <= (unsigned)span
#include "bnn_dev.h"

namespace bnn {

__global__ __launch_bounds__(256) void sign_threshold_kernel(const float* __restrict__ alpha,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ bn_a,
                                                             const float* __restrict__ bn_b, int O, int kmax,
                                                             int32_t* __restrict__ thr) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= O) return;
  const float al = alpha[o], bi = bias ? bias[o] : 0.0f, sc = scale ? scale[o] : 1.0f;
  const float a = bn_a ? bn_a[o] : 1.0f, b = bn_b ? bn_b[o] : 0.0f;
  auto pos = [&](int d) {
    float y = fmaf(al, (float)d, bi);
    if (scale) y *= sc;
    if (bn_a) y = fmaf(y, a, b);
    return is_pos(y);
  };
  const bool p0 = pos(-kmax), p1 = pos(kmax);
  int lo, span;
  if (p0 && p1) {          // always 1
    lo = -kmax; span = 2 * kmax;
  } else if (!p0 && !p1) {  // never 1 (a NaN constant lands here too): an interval no dot can reach
    lo = 0x40000000; span = 0;
  } else if (p1) {          // increasing: [first d with pos(d), kmax]
    int l = -kmax, h = kmax;  // pos(l) == false, pos(h) == true
    while (h - l > 1) {
      const int m = l + (h - l) / 2;
      if (pos(m)) h = m; else l = m;
    }
    lo = h; span = kmax - h;
  } else {                  // decreasing: [-kmax, last d with pos(d)]
    int l = -kmax, h = kmax;  // pos(l) == true, pos(h) == false
    while (h - l > 1) {
      const int m = l + (h - l) / 2;
      if (pos(m)) l = m; else h = m;
    }
    lo = -kmax; span = l + kmax;
  }
  thr[2 * o] = lo;
  thr[2 * o + 1] = span;
}

int launch_sign_thresholds(const float* alpha, const float* bias, const float* scale, const float* bn_a,
                           const float* bn_b, int O, int kmax, int32_t* thr, hipStream_t stream) {
  hipLaunchKernelGGL(sign_threshold_kernel, dim3((O + 255) / 256), dim3(256), 0, stream, alpha, bias, scale, bn_a, bn_b,
                     O, kmax, thr);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
