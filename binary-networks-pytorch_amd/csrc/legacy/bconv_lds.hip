// legacy/bconv_lds.hip — TEST-ONLY (libbnn_hip_legacy.so; never loaded by the product path).
//
// The weight path north_star prescribes literally — packed weight tiles staged in LDS, one tile per 4-wave workgroup —
// kept as an independent implementation of the same integers: tests/test_gpu_fused.py compares it with the product
// kernel (scalar-cache weight stream into SGPRs, csrc/bconv.hip), which measured 2x faster on every ResNet-18 shape
// (512->512 7x7 b256: 100 vs 216 us).  Until round 3 this kernel rode in libbnn_hip.so behind BNN_HIP_FLAG_WEIGHTS_LDS.
#include "../bconv_core.h"

namespace bnn {

// ---------------------------------------------------------------------------------
// Tiled kernel, weights staged in LDS.  A workgroup = 4 waves = 256 pixels sharing one
// (ob, chunk) weight tile; every wave reads the tile with wave-uniform (broadcast)
// ds_read_b128.  Best when many output-channel blocks are in flight at once (small images,
// wide layers: ResNet layer3/layer4), where the scalar cache thrashes and every s_load pays
// an L2 round trip.
// ---------------------------------------------------------------------------------
constexpr int kLdsWaves = 4;

template <int KH, int KW, int CWC>
__global__ __launch_bounds__(kLdsWaves* kWave) void bconv_lds_kernel(
    const uint32_t* __restrict__ P, const uint32_t* __restrict__ M, const uint32_t* __restrict__ W,
    BNN_EPI_PARAMS, const Geo g) {
  constexpr int T = KH * KW;
  constexpr int NW = T * CWC;
  constexpr int TILE = kOCB * NW;          // words per (ob, chunk) weight tile
  constexpr int NV = (TILE + 3) / 4;       // uint4 pieces
  constexpr int NT = kLdsWaves * kWave;    // threads
  constexpr int PER = (NV + NT - 1) / NT;  // pieces per thread
  __shared__ __attribute__((aligned(16))) uint32_t wl[2][NV * 4];
  BNN_EPI_INIT;
  const Pix px = decode_pixel(g, blockIdx.x * NT + threadIdx.x);
  const int ob = blockIdx.y;
  const bool active = ob * kOCB < g.O;

  int acc[kOCB];
#pragma unroll
  for (int j = 0; j < kOCB; ++j) acc[j] = 0;
  int nz = 0;

  if (active) {
    const uint4* wsrc = reinterpret_cast<const uint4*>(W + (size_t)ob * g.nchunk * TILE);
    uint4 stage[PER];
    auto fetch = [&](int ch) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int v = threadIdx.x + k * NT;
        if (v < NV) stage[k] = wsrc[(size_t)ch * NV + v];
      }
    };
    auto commit = [&](int buf) {
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int v = threadIdx.x + k * NT;
        if (v < NV) reinterpret_cast<uint4*>(wl[buf])[v] = stage[k];
      }
    };
    fetch(0);
    commit(0);
    for (int ch = 0; ch < g.nchunk; ++ch) {
      const int buf = ch & 1;
      __syncthreads();  // tile `ch` is in wl[buf]; everyone is done reading wl[buf^1]
      if (ch + 1 < g.nchunk) fetch(ch + 1);  // global -> registers, lands during the VALU work
      uint32_t pr[NW], mr[NW];
      load_field<KH, KW, CWC>(g, px, ch, P, M, pr, mr);
      nz = count_nonzero<NW>(pr, mr, nz);
      const uint32_t* wt = wl[buf];
#pragma unroll
      for (int j = 0; j < kOCB; ++j) {
        int t0 = 0, t1 = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
          const uint32_t d = disagree(wt[j * NW + i], mr[i], pr[i]);  // uniform address: broadcast
          if (i & 1) t1 = popc_acc(d, t1);
          else t0 = popc_acc(d, t0);
        }
        acc[j] += t0 + t1;
      }
      if (ch + 1 < g.nchunk) commit(buf ^ 1);
    }
  }
#pragma unroll
  for (int j = 0; j < kOCB; ++j) acc[j] = nz - 2 * acc[j];  // dot = non-zero count - 2 * disagreements
  uint32_t pbits = 0u, mbits = 0u;
  float resv[kOCB];
  prefetch_residual<kOCB, EP_RUNTIME>(g, px, ob * kOCB, epi, resv);
  epilogue<kOCB, EP_RUNTIME>(g, px, ob * kOCB, acc, resv, epi, pbits, mbits);
  store_packed(g, px, ob, pbits, mbits, epi);
}

template <int KH, int KW, int CWC>
static void launch_lds_t(const ConvP& p, hipStream_t s) {
  constexpr int NT = kLdsWaves * kWave;
  const dim3 grid((p.npix + NT - 1) / NT, oblocks(p));
  hipLaunchKernelGGL((bconv_lds_kernel<KH, KW, CWC>), grid, dim3(NT), 0, s, p.P, p.M, p.W,
                     BNN_EPI_ACTUALS, make_geo(p));
}


int launch_bconv_lds(const ConvP& p, hipStream_t s) {
  if (p.dh != 1 || p.dw != 1 || !small_indices(p)) return BNN_HIP_ERR_UNSUPPORTED;
  if (p.KH == 3 && p.KW == 3 && p.cwc == 4) launch_lds_t<3, 3, 4>(p, s);
  else if (p.KH == 3 && p.KW == 3 && p.cwc == 2) launch_lds_t<3, 3, 2>(p, s);
  else return BNN_HIP_ERR_UNSUPPORTED;
  return hipGetLastError() == hipSuccess ? BNN_HIP_OK : BNN_HIP_ERR_LAUNCH;
}

}  // namespace bnn
