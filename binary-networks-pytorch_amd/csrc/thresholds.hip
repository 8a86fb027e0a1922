// thresholds.hip — BatchNorm + ReLU + sign() of a binary conv folded into an integer test on the popcount result.
//
// For the conv1-type layer of a residual block (bnn/models/layers/res_block.py:40-56: conv -> bn -> relu, the next
// binary conv binarises the result) the only thing that leaves the epilogue is the bit
//     P[o] = ( fmaf( fmaf(alpha[o], dot, bias[o]) [* scale[o]], bn_a[o], bn_b[o] ) > 0 ),      dot in [-K, K] integer.
// Every step is a monotone function of `dot` (rounding is monotone), so the set of dots whose bit is 1 is an INTERVAL.
// One thread per channel finds its ends by bisection, evaluating exactly the float operations of the conv epilogue
// (csrc/bconv.hip: epilogue()); the conv kernel then needs one integer compare per channel and pixel instead of
// int->float, two fmas and a float compare — same bits by construction (tests/test_gpu_fused.py).
// A monotone function crosses zero once, so the interval always touches an end of [-K, K] and ONE one-sided compare
// is enough:   thr[4o] = T,   P  <=>  (dot >= T) XOR flip_o ,
// with the flip bits of a 32-channel block gathered into one word that thr[4o+1] of the block's EVEN channels repeats
// (bit k = flip of channel 32*(o/32) + k; the odd channels' word 1 holds the parities of the block's T instead): increasing channels have flip 0 and T = first dot whose bit is 1,
// decreasing ones flip 1 and T = last such dot + 1; "always" is T = -K, "never" (a NaN constant too) T = 2^30.
#include "bnn_dev.h"

namespace bnn {

__global__ __launch_bounds__(256) void sign_threshold_kernel(const float* __restrict__ alpha,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ bn_a,
                                                             const float* __restrict__ bn_b, int O, int kmax,
                                                             int32_t* __restrict__ thr) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  const bool in = o < O;
  const int o_pad = (O + 31) / 32 * 32;  // the table covers whole 32-channel blocks: pad channels are "never"
  const int oc = in ? o : O - 1;  // lanes past the last channel recompute it (they take part in the ballots below)
  const float al = alpha[oc], bi = bias ? bias[oc] : 0.0f, sc = scale ? scale[oc] : 1.0f;
  const float a = bn_a ? bn_a[oc] : 1.0f, b = bn_b ? bn_b[oc] : 0.0f;
  auto pos = [&](int d) {
    float y = fmaf(al, (float)d, bi);
    if (scale) y *= sc;
    if (bn_a) y = fmaf(y, a, b);
    return is_pos(y);
  };
  const bool p0 = pos(-kmax), p1 = pos(kmax);
  int T;
  bool flip = false;
  if (p0 && p1) {           // always 1
    T = -kmax;
  } else if (!p0 && !p1) {  // never 1 (a NaN constant lands here too): a bound no dot can reach
    T = 0x40000000;
  } else if (p1) {          // increasing: 1 on [first d with pos(d), kmax]
    int l = -kmax, h = kmax;  // pos(l) == false, pos(h) == true
    while (h - l > 1) {
      const int m = l + (h - l) / 2;
      if (pos(m)) h = m; else l = m;
    }
    T = h;
  } else {                  // decreasing: 1 on [-kmax, last d with pos(d)] = NOT (d >= that + 1)
    int l = -kmax, h = kmax;  // pos(l) == true, pos(h) == false
    while (h - l > 1) {
      const int m = l + (h - l) / 2;
      if (pos(m)) l = m; else h = m;
    }
    T = h;
    flip = true;
  }
  if (!in) {  // pad channel of the last block
    T = 0x40000000;
    flip = false;
  }
  // flip word and PARITY word (bit k = T of channel k is odd) of the 32-channel block: a wave holds two consecutive
  // blocks (256 threads = consecutive channels).  Word 1 of an even channel is the block's flip word, of an odd channel
  // its parity word (the two-instruction form of the test picks its carry mask by T's parity: one scalar bit test on
  // this word instead of a scalar load of T per channel)
  const unsigned long long fm = __ballot(in && flip), pm = __ballot((T & 1) != 0);
  const uint32_t fword = (threadIdx.x & 32) ? (uint32_t)(fm >> 32) : (uint32_t)fm;
  const uint32_t pword = (threadIdx.x & 32) ? (uint32_t)(pm >> 32) : (uint32_t)pm;
  const uint32_t word = (o & 1) ? pword : fword;
  if (o < o_pad) {
    // words 2, 3: the comparand of the two-instruction form of the test (bconv_core.h: midt2_shift_in), for the
    // agreement count (non-negative activations) and for the disagreement count, both with the bias of the chains
    constexpr int kBias = 1 << 20;
    const int a_nn = (int)(((long long)T + 1 + 2 * kBias) >> 1);                    // ceil(T / 2) + bias
    const long long tp = ((long long)2 + 2 * kBias - T) >> 1;                       // floor(-T / 2) + 1 + bias
    thr[4 * o] = T;
    thr[4 * o + 1] = (int32_t)word;
    thr[4 * o + 2] = a_nn;
    thr[4 * o + 3] = (int)(tp < 0 ? 0 : tp);
  }
}

int launch_sign_thresholds(const float* alpha, const float* bias, const float* scale, const float* bn_a,
                           const float* bn_b, int O, int kmax, int32_t* thr, hipStream_t stream) {
  hipLaunchKernelGGL(sign_threshold_kernel, dim3(((O + 31) / 32 * 32 + 255) / 256), dim3(256), 0, stream, alpha, bias, scale, bn_a, bn_b,
                     O, kmax, thr);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
