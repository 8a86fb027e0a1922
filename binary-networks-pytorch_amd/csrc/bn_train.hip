// bn_train.hip — training-mode BatchNorm2d fused with what surrounds it in the reference's residual blocks.
//
// The step "after" the binary-conv path: in training mode every binary convolution of the
// reference's ResNets is followed by   bnN(...)  [+ identity]  [-> ReLU]   (bnn/models/layers/res_block.py:40-56,
// bnn/models/resnet.py:150-153), evaluated by the library as separate passes over the fp32 tensor (BatchNorm statistics,
// normalisation, add, ReLU: 5-8 HBM passes forward, 8 backward).  Of the 29.5 ms ResNet-18 training step at batch 256,
// 9.1 ms were the library's BatchNorm kernels (4x off the HBM roofline) and 6 ms its element-wise kernels.  Here:
//
//   forward :  stats  (1 read)     per-channel sum / sum of squares, fp64 accumulation, deterministic two-stage reduction
//              apply  (1-2 reads, 1 write)   y = relu( x * scale[c] + shift[c] (+ residual) ), running statistics updated
//   backward:  reduce (3 reads)    dbeta = sum g, dgamma = sum g * xhat,   g = gy * 1[y > 0]
//              dx     (3 reads, 1-2 writes)  dx = gamma * invstd * (g - dbeta / m - xhat * dgamma / m),  dres = g
//
// Semantics of torch.nn.BatchNorm2d in training mode (biased variance for the normalisation, unbiased for
// running_var, momentum update), of ReLU and of `out += identity`; fp32 tensors, NCHW.  The statistics are accumulated
// in double precision (the library: fp32 Welford), so results agree with the library to fp32 rounding, not bit for bit.
// HBM-bound kernels: float4 accesses along the contiguous HW axis of a (image, channel) row.
#include "bnn_dev.h"

namespace bnn {

namespace bnt {
constexpr int NT = 256;
constexpr int MAX_SPLITS = 64;
}  // namespace bnt

// sum over a block of (a, b) in double: wave shuffles, then LDS
__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double sa[bnt::NT / 64], sb[bnt::NT / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off);
    b += __shfl_xor(b, off);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { sa[wave] = a; sb[wave] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
#pragma unroll
  for (int w = 0; w < bnt::NT / 64; ++w) { a += sa[w]; b += sb[w]; }   // fixed order: deterministic
}

// partial[c][s] = (sum, sum of squares) of channel c over the images of split s.  The block walks the flattened
// (image, unit) index space of its images (unit = float4 or float of the contiguous HW axis), so that 7x7 images keep
// all 256 threads busy too.
template <int VEC>
__global__ __launch_bounds__(bnt::NT) void bn_stats_kernel(const float* __restrict__ x, int N, int C, int HW, int per,
                                                           double* __restrict__ partial) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int n0 = s * per, n1 = min(N, n0 + per);
  const int upr = HW / VEC;                       // units per (image, channel) row
  const int total = (n1 - n0) * upr;
  double a = 0.0, b = 0.0;
  for (int idx = threadIdx.x; idx < total; idx += bnt::NT) {
    const int dn = idx / upr, u = idx - dn * upr;
    const float* p = x + ((size_t)(n0 + dn) * C + c) * HW + (size_t)u * VEC;
    if constexpr (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      a += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      b += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      const float v = *p;
      a += (double)v;
      b += (double)v * v;
    }
  }
  block_sum2(a, b);
  if (threadIdx.x == 0) {
    partial[((size_t)c * S + s) * 2 + 0] = a;
    partial[((size_t)c * S + s) * 2 + 1] = b;
  }
}

// one thread per channel: partials -> mean, invstd, scale = gamma * invstd, shift = beta - mean * scale; running statistics
__global__ __launch_bounds__(64) void bn_finalize_kernel(const double* __restrict__ partial, int S, int C, double count,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ mean_out,
                                                         float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                                         float* __restrict__ shift_out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < S; ++s) {               // fixed order: deterministic
    a += partial[((size_t)c * S + s) * 2 + 0];
    b += partial[((size_t)c * S + s) * 2 + 1];
  }
  const double mean = a / count;
  double var = b / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
  const float scale = g * invstd;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  scale_out[c] = scale;
  shift_out[c] = (float)((double)bt - mean * (double)scale);
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * running_mean[c] + (double)momentum * mean);
    running_var[c] = (float)((1.0 - (double)momentum) * running_var[c] + (double)momentum * unbiased);
  }
}

// y = relu( x * scale[c] + shift[c] (+ res) ): flat over the tensor in units of VEC floats (a unit never straddles a row)
template <int VEC, bool RELU, bool RES>
__global__ __launch_bounds__(bnt::NT) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale_c,
                                                           const float* __restrict__ shift_c, const float* __restrict__ res,
                                                           float* __restrict__ y, long long units, int C, int upr) {
  const long long u = (long long)blockIdx.x * bnt::NT + threadIdx.x;
  if (u >= units) return;
  const int c = (int)((u / upr) % C);
  const float scale = scale_c[c], shift = shift_c[c];
  const size_t i0 = (size_t)u * VEC;
  if constexpr (VEC == 4) {
    float4 v = *reinterpret_cast<const float4*>(x + i0);
    v.x = fmaf(v.x, scale, shift); v.y = fmaf(v.y, scale, shift); v.z = fmaf(v.z, scale, shift); v.w = fmaf(v.w, scale, shift);
    if constexpr (RES) {
      const float4 r = *reinterpret_cast<const float4*>(res + i0);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if constexpr (RELU) {  // (x < 0 ? 0 : x keeps NaN, like torch.relu; fmaxf would return 0)
      v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
    }
    *reinterpret_cast<float4*>(y + i0) = v;
  } else {
    float v = fmaf(x[i0], scale, shift);
    if constexpr (RES) v += res[i0];
    if constexpr (RELU) v = v < 0.f ? 0.f : v;
    y[i0] = v;
  }
}

// partial[c][s] = (sum g, sum g * xhat) over the images of split s;  g = gy * 1[y > 0] when RELU
template <int VEC, bool RELU>
__global__ __launch_bounds__(bnt::NT) void bn_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int N, int C, int HW,
                                                                int per, double* __restrict__ partial) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int n0 = s * per, n1 = min(N, n0 + per);
  const int upr = HW / VEC;
  const int total = (n1 - n0) * upr;
  const float mu = mean[c], is = invstd[c];
  double a = 0.0, b = 0.0;
  for (int idx = threadIdx.x; idx < total; idx += bnt::NT) {
    const int dn = idx / upr, u = idx - dn * upr;
    const size_t i0 = ((size_t)(n0 + dn) * C + c) * HW + (size_t)u * VEC;
    if constexpr (VEC == 4) {
      float4 g = *reinterpret_cast<const float4*>(gy + i0);
      const float4 xv = *reinterpret_cast<const float4*>(x + i0);
      if constexpr (RELU) {
        const float4 yv = *reinterpret_cast<const float4*>(y + i0);
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      }
      a += (double)g.x + (double)g.y + (double)g.z + (double)g.w;
      b += (double)g.x * ((xv.x - mu) * is) + (double)g.y * ((xv.y - mu) * is) + (double)g.z * ((xv.z - mu) * is) +
           (double)g.w * ((xv.w - mu) * is);
    } else {
      float g = gy[i0];
      if constexpr (RELU) g = y[i0] > 0.f ? g : 0.f;
      a += (double)g;
      b += (double)g * ((x[i0] - mu) * is);
    }
  }
  block_sum2(a, b);
  if (threadIdx.x == 0) {
    partial[((size_t)c * S + s) * 2 + 0] = a;
    partial[((size_t)c * S + s) * 2 + 1] = b;
  }
}

// one thread per channel: dbeta, dgamma and the three coefficients of dx = k * (g - mb - (x - mean) * kg)
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int S, int C, double count,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double sb = 0.0, sg = 0.0;
  for (int s = 0; s < S; ++s) {
    sb += partial[((size_t)c * S + s) * 2 + 0];
    sg += partial[((size_t)c * S + s) * 2 + 1];
  }
  if (dbeta) dbeta[c] = (float)sb;
  if (dgamma) dgamma[c] = (float)sg;
  const float is = invstd[c];
  coef[3 * c + 0] = (gamma ? gamma[c] : 1.0f) * is;      // k
  coef[3 * c + 1] = (float)(sb / count);                  // mb
  coef[3 * c + 2] = is * (float)(sg / count);             // kg: xhat * dgamma / m = (x - mean) * invstd * (sg / m)
}

// dx = k * (g - mb - (x - mean) * kg);  dres = g (the gradient of the residual branch)
template <int VEC, bool RELU, bool RES>
__global__ __launch_bounds__(bnt::NT) void bn_bwd_dx_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ coef, float* __restrict__ dx,
                                                            float* __restrict__ dres, long long units, int C, int upr) {
  const long long u = (long long)blockIdx.x * bnt::NT + threadIdx.x;
  if (u >= units) return;
  const int c = (int)((u / upr) % C);
  const float mu = mean[c], k = coef[3 * c], mb = coef[3 * c + 1], kg = coef[3 * c + 2];
  const size_t i0 = (size_t)u * VEC;
  if constexpr (VEC == 4) {
    float4 g = *reinterpret_cast<const float4*>(gy + i0);
    const float4 xv = *reinterpret_cast<const float4*>(x + i0);
    if constexpr (RELU) {
      const float4 yv = *reinterpret_cast<const float4*>(y + i0);
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    if constexpr (RES) *reinterpret_cast<float4*>(dres + i0) = g;
    float4 d;
    d.x = k * (g.x - mb - (xv.x - mu) * kg);
    d.y = k * (g.y - mb - (xv.y - mu) * kg);
    d.z = k * (g.z - mb - (xv.z - mu) * kg);
    d.w = k * (g.w - mb - (xv.w - mu) * kg);
    *reinterpret_cast<float4*>(dx + i0) = d;
  } else {
    float g = gy[i0];
    if constexpr (RELU) g = y[i0] > 0.f ? g : 0.f;
    if constexpr (RES) dres[i0] = g;
    dx[i0] = k * (g - mb - (x[i0] - mu) * kg);
  }
}

// ------------------------------------------------------------------------------------------------- stem tail
// bn1 -> relu -> maxpool(3, stride 2, pad 1) of the reference's ResNet stem (bnn/models/resnet.py:150-153) in training
// mode.  The library evaluates it as BatchNorm (3 passes over the 822 MB conv output at batch 256), ReLU, max-pool
// (+ 410 MB of int64 indices), and backward as max-pool backward (1.56 ms), ReLU backward, BatchNorm backward — 3.5 ms
// of a 22.8 ms step.  Here the normalised tensor is never written: the forward pools relu(x * scale + shift) straight
// from x and keeps ONE BYTE per pooled output (which of the 9 window positions won); the backward routes gy through
// those codes inside the BatchNorm reductions.
//   forward : p[w] = max over the window of relu(fma(x, scale, shift)),  code[w] = 3 * dy + dx of the first maximum
//   backward: g[pix] = sum of gy[w] over the windows w whose winner is pix and whose p[w] > 0   (ReLU: a winner of value
//             0 passes nothing — so ties at 0 need no tie-breaking rule),  then BatchNorm's dbeta / dgamma / dx on g.
__global__ __launch_bounds__(bnt::NT) void bn_relu_pool_fwd_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ scale_c,
                                                                   const float* __restrict__ shift_c,
                                                                   float* __restrict__ p, unsigned char* __restrict__ code,
                                                                   long long total, int C, int H, int W, int Hp, int Wp) {
  const long long o = (long long)blockIdx.x * bnt::NT + threadIdx.x;
  if (o >= total) return;
  const int px = (int)(o % Wp);
  const long long t = o / Wp;
  const int py = (int)(t % Hp);
  const long long row = t / Hp;                 // n * C + c
  const int c = (int)(row % C);
  const float scale = scale_c[c], shift = shift_c[c];
  const float* xb = x + (size_t)row * H * W;
  float best = -1.0f;                           // below every ReLU output: the first in-range position always wins first
  int bc = 0;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int r = 2 * py - 1 + dy;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int q = 2 * px - 1 + dx;
      if ((unsigned)r < (unsigned)H && (unsigned)q < (unsigned)W) {
        // NaN survives the ReLU and wins the maximum, as in torch's relu + max_pool2d (a diverged run must not turn
        // into all-zero activations with a finite loss)
        const float t = fmaf(xb[(size_t)r * W + q], scale, shift);
        const float v = t < 0.0f ? 0.0f : t;
        if (v > best || v != v) { best = v; bc = 3 * dy + dx; }
      }
    }
  }
  p[o] = best;
  code[o] = (unsigned char)bc;
}

// The same forward with the plane staged in LDS (round 5; planes of at most 64 KB): one workgroup per (image, channel)
// plane reads it once as float4s, keeps relu(fma(x, scale, shift)) in LDS and takes every window from there — the same
// nine comparisons in the same order as above, so the same pooled values and codes.
__global__ __launch_bounds__(bnt::NT) void bn_relu_pool_fwd_lds_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ scale_c,
                                                                       const float* __restrict__ shift_c,
                                                                       float* __restrict__ p, unsigned char* __restrict__ code,
                                                                       int C, int H, int W, int Hp, int Wp) {
  extern __shared__ __attribute__((aligned(16))) float vpl[];
  const int tid = threadIdx.x;
  const size_t plane = blockIdx.x;
  const int c = (int)(plane % C);
  const float scale = scale_c[c], shift = shift_c[c];
  const int hw = H * W, hwp = Hp * Wp;
  const float* xp = x + plane * hw;
  auto act = [&](float xv) {
    const float t = fmaf(xv, scale, shift);
    return t < 0.0f ? 0.0f : t;  // NaN stays NaN
  };
  for (int i = tid * 4; i < hw; i += bnt::NT * 4) {
    const float4 xv = *reinterpret_cast<const float4*>(xp + i);
    *reinterpret_cast<float4*>(vpl + i) = float4{act(xv.x), act(xv.y), act(xv.z), act(xv.w)};
  }
  __syncthreads();
  for (int idx = tid; idx < hwp; idx += bnt::NT) {
    const int py = idx / Wp, px = idx - py * Wp;
    float best = -1.0f;
    int bc = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int r = 2 * py - 1 + dy;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int q = 2 * px - 1 + dx;
        if ((unsigned)r < (unsigned)H && (unsigned)q < (unsigned)W) {
          const float v = vpl[r * W + q];
          if (v > best || v != v) { best = v; bc = 3 * dy + dx; }
        }
      }
    }
    p[plane * hwp + idx] = best;
    code[plane * hwp + idx] = (unsigned char)bc;
  }
}

// partial[c][s] = (sum g, sum g * xhat) over the images of split s, from the pooled outputs: g is non-zero only at winners
__global__ __launch_bounds__(bnt::NT) void bn_pool_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ p,
                                                                     const unsigned char* __restrict__ code,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd, int N, int C, int H,
                                                                     int W, int Hp, int Wp, int per,
                                                                     double* __restrict__ partial) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int n0 = s * per, n1 = min(N, n0 + per);
  const int hwp = Hp * Wp;
  const int total = (n1 - n0) * hwp;
  const float mu = mean[c], is = invstd[c];
  double a = 0.0, b = 0.0;
  for (int idx = threadIdx.x; idx < total; idx += bnt::NT) {
    const int dn = idx / hwp, w = idx - dn * hwp;
    const size_t row = (size_t)(n0 + dn) * C + c;
    const size_t o = row * hwp + w;
    const float g = p[o] > 0.0f ? gy[o] : 0.0f;
    const int py = w / Wp, px = w - py * Wp, cd = code[o];
    const int r = 2 * py - 1 + cd / 3, q = 2 * px - 1 + cd % 3;
    const float xv = x[row * H * W + (size_t)r * W + q];
    a += (double)g;
    b += (double)g * ((xv - mu) * is);
  }
  block_sum2(a, b);
  if (threadIdx.x == 0) {
    partial[((size_t)c * S + s) * 2 + 0] = a;
    partial[((size_t)c * S + s) * 2 + 1] = b;
  }
}

// dx over the conv pixels.  A wave owns one image row of one (image, channel) plane, a lane up to four consecutive pixels
// of it (no per-element divisions: the row index is wave-uniform); the gradient arrives from the <= 2 x 3 windows that
// can have their winner among those pixels — each window's code is decoded once and its gy added to the pixel it names.
template <int VEC>
__global__ __launch_bounds__(bnt::NT) void bn_pool_bwd_dx_kernel(const float* __restrict__ gy, const float* __restrict__ p,
                                                                 const unsigned char* __restrict__ code,
                                                                 const float* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ coef, float* __restrict__ dx,
                                                                 long long rows, int C, int H, int W, int Hp, int Wp,
                                                                 int lpr_shift) {
  // lanes per row = 2^lpr_shift (the smallest power of two >= W / VEC, at most 64): a wave covers 64 >> lpr_shift
  // image rows at a time, so that narrow images (112 pixels: 28 float4s) keep the lanes busy
  const int wave = (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int rpw = 64 >> lpr_shift, lpr = 1 << lpr_shift;
  const long long gr = ((long long)blockIdx.x * (bnt::NT / 64) + wave) * rpw + (lane >> lpr_shift);
  if (gr >= rows) return;
  const long long plane = gr / H;                                         // n * C + c
  const int r = (int)(gr - plane * H), c = (int)(plane % C);
  const float mu = mean[c], k = coef[3 * c], mb = coef[3 * c + 1], kg = coef[3 * c + 2];
  const float* xr = x + (size_t)gr * W;
  float* dr = dx + (size_t)gr * W;
  const size_t pb = (size_t)plane * Hp * Wp;
  const int py0 = r >> 1, py1 = (r + 1) >> 1;                             // pooled rows whose windows contain row r
  for (int q = (lane & (lpr - 1)) * VEC; q < W; q += lpr * VEC) {
    float g[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = 0.0f;
    // pooled columns whose windows reach pixels q .. q + VEC - 1: floor(q / 2) .. floor((q + VEC) / 2)
    const int pxa = q >> 1, pxb = min((q + VEC) >> 1, Wp - 1);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int py = a ? py1 : py0;
      if ((a && py1 == py0) || py >= Hp) continue;
      const int dy = r - (2 * py - 1);                                    // the window row that IS image row r
      for (int px = pxa; px <= pxb; ++px) {
        const size_t o = pb + (size_t)py * Wp + px;
        const int cd = code[o];
        const int wy = cd / 3, wx = cd - 3 * wy;
        const int e = 2 * px - 1 + wx - q;                                // the winner's position among this lane's pixels
        if (wy == dy && (unsigned)e < (unsigned)VEC && p[o] > 0.0f) {
          const float gv = gy[o];
#pragma unroll
          for (int j = 0; j < VEC; ++j) g[j] += (j == e) ? gv : 0.0f;
        }
      }
    }
    if constexpr (VEC == 4) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + q);
      float4 d;
      d.x = k * (g[0] - mb - (xv.x - mu) * kg);
      d.y = k * (g[1] - mb - (xv.y - mu) * kg);
      d.z = k * (g[2] - mb - (xv.z - mu) * kg);
      d.w = k * (g[3] - mb - (xv.w - mu) * kg);
      *reinterpret_cast<float4*>(dr + q) = d;
    } else {
      dr[q] = k * (g[0] - mb - (xr[q] - mu) * kg);
    }
  }
}

// ------------------------------------------------------------------------------------------------- host side
int bn_train_splits(int N, int C, int HW) {
  // enough blocks to fill the chip (~4 per CU), at most one split per image
  int s = (1024 + C - 1) / C;
  if (s > bnt::MAX_SPLITS) s = bnt::MAX_SPLITS;
  if (s > N) s = N;
  if (s < 1) s = 1;
  const int per = (N + s - 1) / s;
  (void)HW;
  return (N + per - 1) / per;
}

// ------------------------------------------------------------------------------------------------- shortcut pooling
// Backward of AvgPool2d(2, 2) on an even-sized map (the shortcut branch of a down-sampling block, bnn/models/resnet.py:
// 128-133, in a training step): gx[n, c, 2y + a, 2x + b] = gy[n, c, y, x] / 4.  A streaming kernel (the library's generic
// avg_pool2d backward: 0.53 ms for the three shortcuts of a ResNet-18 step at batch 256 — this: 0.09).  VEC: two pooled
// values per thread in, two float4 rows out.
template <bool VEC>
__global__ __launch_bounds__(bnt::NT) void avgpool2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                               long long units, int Ho, int Wo) {
  const long long u = (long long)blockIdx.x * bnt::NT + threadIdx.x;
  if (u >= units) return;
  if constexpr (VEC) {
    const int wp = Wo / 2;
    const long long row = u / wp;              // n * C * Ho + y
    const int xp = (int)(u - row * wp);
    const float2 g = *reinterpret_cast<const float2*>(gy + row * Wo + 2 * xp);
    const float a = 0.25f * g.x, b = 0.25f * g.y;
    const long long plane_row = row / Ho;      // n * C
    const int y = (int)(row - plane_row * Ho);
    float* o = gx + ((plane_row * 2 * Ho + 2 * y) * (2LL * Wo)) + 4 * xp;
    const float4 v{a, a, b, b};
    *reinterpret_cast<float4*>(o) = v;
    *reinterpret_cast<float4*>(o + 2 * Wo) = v;
  } else {
    const long long row = u / Wo;
    const int x = (int)(u - row * Wo);
    const float a = 0.25f * gy[u];
    const long long plane_row = row / Ho;
    const int y = (int)(row - plane_row * Ho);
    float* o = gx + ((plane_row * 2 * Ho + 2 * y) * (2LL * Wo)) + 2 * x;
    o[0] = a; o[1] = a; o[2 * Wo] = a; o[2 * Wo + 1] = a;
  }
}

int launch_avgpool2x2_bwd(const float* gy, int N, int C, int Ho, int Wo, float* gx, hipStream_t s) {
  const bool vec = Wo % 2 == 0 && (reinterpret_cast<uintptr_t>(gy) & 7) == 0 && (reinterpret_cast<uintptr_t>(gx) & 15) == 0;
  const long long units = (long long)N * C * Ho * (vec ? Wo / 2 : Wo);
  const dim3 grid((unsigned)((units + bnt::NT - 1) / bnt::NT));
  if (vec)
    hipLaunchKernelGGL(avgpool2_bwd_kernel<true>, grid, dim3(bnt::NT), 0, s, gy, gx, units, Ho, Wo);
  else
    hipLaunchKernelGGL(avgpool2_bwd_kernel<false>, grid, dim3(bnt::NT), 0, s, gy, gx, units, Ho, Wo);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

static bool vec4(const void* a, const void* b, const void* c, const void* d, int HW) {
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return HW % 4 == 0 && al(a) && al(b) && al(c) && al(d);
}

int launch_bn_stats(const float* x, int N, int C, int HW, int splits, double* partial, hipStream_t s) {
  const int per = (N + splits - 1) / splits;
  const dim3 grid((unsigned)C, (unsigned)splits);
  if (vec4(x, nullptr, nullptr, nullptr, HW))
    hipLaunchKernelGGL(bn_stats_kernel<4>, grid, dim3(bnt::NT), 0, s, x, N, C, HW, per, partial);
  else
    hipLaunchKernelGGL(bn_stats_kernel<1>, grid, dim3(bnt::NT), 0, s, x, N, C, HW, per, partial);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// the apply pass: y = relu?( fma(x, scale[c], shift[c]) (+ res) )
static int apply_pass(const float* x, const float* scale, const float* shift, const float* res, int relu, float* y, int N,
                      int C, int HW, hipStream_t s) {
  const bool v4 = vec4(x, res, y, nullptr, HW);
  const int vec = v4 ? 4 : 1;
  const long long units = (long long)N * C * HW / vec;
  const dim3 grid((unsigned)((units + bnt::NT - 1) / bnt::NT));
  const int upr = HW / vec;
#define BNN_APPLY(V_, R_, S_)   hipLaunchKernelGGL((bn_apply_kernel<V_, R_, S_>), grid, dim3(bnt::NT), 0, s, x, scale, shift, res, y, units, C, upr)
  if (v4) {
    if (relu && res) BNN_APPLY(4, true, true); else if (relu) BNN_APPLY(4, true, false);
    else if (res) BNN_APPLY(4, false, true); else BNN_APPLY(4, false, false);
  } else {
    if (relu && res) BNN_APPLY(1, true, true); else if (relu) BNN_APPLY(1, true, false);
    else if (res) BNN_APPLY(1, false, true); else BNN_APPLY(1, false, false);
  }
#undef BNN_APPLY
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// finalize (statistics, running statistics, scale / shift into `work` = [scale C | shift C]) + the apply pass
int launch_bn_apply(const float* x, const double* partial, int splits, const float* gamma, const float* beta,
                    const float* res, int relu, float* y, int N, int C, int HW, float eps, float momentum, float* rm,
                    float* rv, float* mean_out, float* invstd_out, float* work, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, partial, splits, C,
                     (double)N * HW, gamma, beta, eps, momentum, rm, rv, mean_out, invstd_out, work, work + C);
  return apply_pass(x, work, work + C, res, relu, y, N, C, HW, s);
}

// eval-mode BatchNorm with folded constants (+ residual) (+ ReLU): the apply pass alone
int launch_bn_act(const float* x, const float* scale, const float* shift, const float* res, int relu, float* y, int N,
                  int C, int HW, hipStream_t s) {
  return apply_pass(x, scale, shift, res, relu, y, N, C, HW, s);
}

int launch_bn_bwd_reduce(const float* gy, const float* y, const float* x, const float* mean, const float* invstd, int N,
                         int C, int HW, int splits, double* partial, hipStream_t s) {
  const int per = (N + splits - 1) / splits;
  const dim3 grid((unsigned)C, (unsigned)splits);
  const bool v4 = vec4(gy, y, x, nullptr, HW);
  if (v4 && y) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true>), grid, dim3(bnt::NT), 0, s, gy, y, x, mean, invstd, N, C, HW, per, partial);
  else if (v4) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, false>), grid, dim3(bnt::NT), 0, s, gy, y, x, mean, invstd, N, C, HW, per, partial);
  else if (y) hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, true>), grid, dim3(bnt::NT), 0, s, gy, y, x, mean, invstd, N, C, HW, per, partial);
  else hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, false>), grid, dim3(bnt::NT), 0, s, gy, y, x, mean, invstd, N, C, HW, per, partial);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// finalize (dgamma, dbeta, coefficients into `work` = [3 C]) + the dx pass
int launch_bn_bwd_dx(const float* gy, const float* y, const float* x, const float* mean, const float* invstd,
                     const float* gamma, const double* partial, int splits, float* dx, float* dres, float* dgamma,
                     float* dbeta, int N, int C, int HW, float* work, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, partial, splits, C,
                     (double)N * HW, gamma, invstd, dgamma, dbeta, work);
  const bool v4 = vec4(gy, y, x, dx, HW) && vec4(dres, nullptr, nullptr, nullptr, HW);
  const int vec = v4 ? 4 : 1;
  const long long units = (long long)N * C * HW / vec;
  const dim3 grid((unsigned)((units + bnt::NT - 1) / bnt::NT));
  const int upr = HW / vec;
#define BNN_DX(V_, R_, S_)   hipLaunchKernelGGL((bn_bwd_dx_kernel<V_, R_, S_>), grid, dim3(bnt::NT), 0, s, gy, y, x, mean, work, dx, dres, units, C, upr)
  if (v4) {
    if (y && dres) BNN_DX(4, true, true); else if (y) BNN_DX(4, true, false);
    else if (dres) BNN_DX(4, false, true); else BNN_DX(4, false, false);
  } else {
    if (y && dres) BNN_DX(1, true, true); else if (y) BNN_DX(1, true, false);
    else if (dres) BNN_DX(1, false, true); else BNN_DX(1, false, false);
  }
#undef BNN_DX
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

// The same dx with the routing done in LDS (round 5; planes of at most 64 KB: 112 x 112 at ImageNet size).  One workgroup
// per (image, channel) plane: every pooled output is read ONCE, coalesced (the gather form above reads each of them from
// the lanes of three conv rows: 18 small loads per float4 of dx), its gy is added to the plane's LDS copy at the pixel
// its code names, and the plane leaves as float4s.  Windows of equal row and column parity never share a pixel (their
// 3 x 3 fields are two pixels apart), so the four parity classes are added one after the other without atomics: a fixed
// order, the same bits on every run.
__global__ __launch_bounds__(bnt::NT) void bn_pool_bwd_dx_lds_kernel(const float* __restrict__ gy, const float* __restrict__ p,
                                                                     const unsigned char* __restrict__ code,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ coef, float* __restrict__ dx,
                                                                     int C, int H, int W, int Hp, int Wp) {
  extern __shared__ __attribute__((aligned(16))) float gpl[];
  constexpr int PER = 16;  // pooled outputs per thread: Hp * Wp <= 16 * 256 (H * W <= 16384)
  const int tid = threadIdx.x;
  const size_t plane = blockIdx.x;
  const int c = (int)(plane % C);
  const int hw = H * W, hwp = Hp * Wp;
  const size_t pb = plane * hwp;
  // this thread's pooled outputs: value (0 if the ReLU closed the window: p == 0) and target pixel, parity class
  float val[PER];
  int tgt[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = tid + u * bnt::NT;
    val[u] = 0.0f;
    tgt[u] = -1;
    if (idx < hwp) {
      const int py = idx / Wp, px = idx - py * Wp;
      const int cd = code[pb + idx];
      const float g = gy[pb + idx];
      const bool open = p[pb + idx] > 0.0f;
      const int wy = cd / 3, wx = cd - 3 * wy;
      val[u] = open ? g : 0.0f;
      tgt[u] = open ? (((2 * py - 1 + wy) * W + 2 * px - 1 + wx) << 2) | ((py & 1) << 1) | (px & 1) : -1;
    }
  }
  for (int i = tid * 4; i < hw; i += bnt::NT * 4) *reinterpret_cast<float4*>(gpl + i) = float4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (tgt[u] >= 0 && (tgt[u] & 3) == ph) gpl[tgt[u] >> 2] += val[u];
    __syncthreads();
  }
  const float mu = mean[c], k = coef[3 * c], mb = coef[3 * c + 1], kg = coef[3 * c + 2];
  const float* xp = x + plane * hw;
  float* dp = dx + plane * hw;
  for (int i = tid * 4; i < hw; i += bnt::NT * 4) {
    const float4 xv = *reinterpret_cast<const float4*>(xp + i);
    const float4 g = *reinterpret_cast<const float4*>(gpl + i);
    float4 d;
    d.x = k * (g.x - mb - (xv.x - mu) * kg);
    d.y = k * (g.y - mb - (xv.y - mu) * kg);
    d.z = k * (g.z - mb - (xv.z - mu) * kg);
    d.w = k * (g.w - mb - (xv.w - mu) * kg);
    *reinterpret_cast<float4*>(dp + i) = d;
  }
}

// stem tail forward: statistics + finalize as launch_bn_apply, then the pooling pass instead of the apply pass
int launch_bn_relu_pool_fwd(const float* x, const double* partial, int splits, const float* gamma, const float* beta,
                            float* p, unsigned char* code, int N, int C, int H, int W, float eps, float momentum, float* rm,
                            float* rv, float* mean_out, float* invstd_out, float* work, hipStream_t s) {
  const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, partial, splits, C,
                     (double)N * H * W, gamma, beta, eps, momentum, rm, rv, mean_out, invstd_out, work, work + C);
  const long long total = (long long)N * C * Hp * Wp;
  const long long hw = (long long)H * W;
  if (hw % 4 == 0 && hw <= 16384 && vec4(x, nullptr, nullptr, nullptr, 4) &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(bn_relu_pool_fwd_lds_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(hw * 4)) == hipSuccess) {
    hipLaunchKernelGGL(bn_relu_pool_fwd_lds_kernel, dim3((unsigned)((long long)N * C)), dim3(bnt::NT), (size_t)hw * 4, s, x, work,
                       work + C, p, code, C, H, W, Hp, Wp);
    return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3((unsigned)((total + bnt::NT - 1) / bnt::NT)), dim3(bnt::NT), 0, s, x,
                     work, work + C, p, code, total, C, H, W, Hp, Wp);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

int launch_bn_relu_pool_bwd(const float* gy, const float* p, const unsigned char* code, const float* x, const float* mean,
                            const float* invstd, const float* gamma, int N, int C, int H, int W, int splits,
                            double* partial, float* work, float* dx, float* dgamma, float* dbeta, hipStream_t s) {
  const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
  const int per = (N + splits - 1) / splits;
  const long long hw = (long long)H * W;
  const bool lds_plane = hw % 4 == 0 && hw <= 16384 && vec4(x, dx, nullptr, nullptr, 4);
  // (the reduction keeps its gather form: with the plane staged in LDS — one block per channel and split walking its
  // images — it measured 0.77 ms against 0.43)
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3((unsigned)C, (unsigned)splits), dim3(bnt::NT), 0, s, gy, p, code, x, mean,
                     invstd, N, C, H, W, Hp, Wp, per, partial);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, partial, splits, C,
                     (double)N * H * W, gamma, invstd, dgamma, dbeta, work);
  const long long rows = (long long)N * C * H;
  const bool v4 = W % 4 == 0 && vec4(x, dx, nullptr, nullptr, 4);
  int lpr_shift = 0;
  while ((1 << lpr_shift) < (v4 ? W / 4 : W) && lpr_shift < 6) ++lpr_shift;
  const long long rows_per_block = (long long)(bnt::NT / 64) * (64 >> lpr_shift);
  const dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
  if (lds_plane && (long long)Hp * Wp <= 16 * bnt::NT &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(bn_pool_bwd_dx_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(hw * 4)) == hipSuccess)
    hipLaunchKernelGGL(bn_pool_bwd_dx_lds_kernel, dim3((unsigned)((long long)N * C)), dim3(bnt::NT), (size_t)hw * 4, s, gy, p, code,
                       x, mean, work, dx, C, H, W, Hp, Wp);
  else if (v4)
    hipLaunchKernelGGL(bn_pool_bwd_dx_kernel<4>, grid, dim3(bnt::NT), 0, s, gy, p, code, x, mean, work, dx, rows, C, H, W, Hp,
                       Wp, lpr_shift);
  else
    hipLaunchKernelGGL(bn_pool_bwd_dx_kernel<1>, grid, dim3(bnt::NT), 0, s, gy, p, code, x, mean, work, dx, rows, C, H, W, Hp,
                       Wp, lpr_shift);
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
