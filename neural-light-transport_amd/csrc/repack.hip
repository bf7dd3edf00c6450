// One launch that refreshes EVERY packed weight buffer of a model after an optimizer step.
// The train step re-derives ~75 fragment arrays per step (forward fragments of every conv, adjoint-family fragments of
// every backward-data launch, LDS-tile fragments); one small launch (+ one allocation, + a slice copy for the adjoint
// ones) each cost more host time than the kernels they serve.  The buffers are allocated once; a device table of
// descriptors says how to refill each of them from the flat parameter bucket.
#include "pack_common.h"

namespace {

template <int MODE>
__device__ __forceinline__ float frag(const nlt_repack_desc& e, long idx) {
  const int N = MODE == NLT_DECONV_K2S2 ? 4 * e.cout : e.cout;
  return nlt_mfma_fragment<MODE>(e.src, idx, e.c0, e.c1, e.cout, N, (N + 15) >> 4, e.full, e.lo);
}

template <int MODE>
__device__ __forceinline__ f32x4 frag4(const nlt_repack_desc& e, long idx) {
  return (f32x4){frag<MODE>(e, idx), frag<MODE>(e, idx + 1), frag<MODE>(e, idx + 2), frag<MODE>(e, idx + 3)};
}

// One wave per 256 packed elements; a lane fills FOUR consecutive ones (in every layout they are the four channels of one
// (tile, lane) slot: the index arithmetic -- a binary search and several divisions by runtime values -- is done once per quad) with
// one 16-byte store.  r01-r05: one element and one 4-byte store per thread, 40 us per train step at config 4.
__global__ __launch_bounds__(64) void repack_all_kernel(const nlt_repack_desc* __restrict__ d, int nd) {
  int lo = 0, hi = nd;                                                 // last descriptor with first_block <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (d[mid].first_block <= (long)blockIdx.x) lo = mid; else hi = mid;
  }
  const nlt_repack_desc e = d[lo];
  const long idx = ((long)blockIdx.x - e.first_block) * 256 + threadIdx.x * 4;
  if (idx >= e.total) return;                                          // (totals are multiples of 256)
  f32x4 v;
  if (e.kind == NLT_REPACK_WINO) {
    const bool tr = e.mode == NLT_DECONV_K2S1;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = nlt_wino_fragment(e.src, idx + j, e.c0, e.cout, e.tn >> 4, e.full, e.lo, tr);
  } else if (e.kind == NLT_REPACK_TILE && e.mode == NLT_DECONV_K2S2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = nlt_tile_fragment_d2(e.src, idx + j, e.c0, e.cout, e.tn >> 4, e.full, e.lo);
  } else if (e.kind == NLT_REPACK_TILE) {
    const bool tr = e.mode == NLT_DECONV_K2S1 || e.mode == NLT_DECONV_K2S2;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = nlt_tile_fragment(e.src, idx + j, e.c0, e.cout, e.tn >> 4, e.full, e.lo, tr);
  }
  else if (e.mode == NLT_CONV1X1) v = frag4<NLT_CONV1X1>(e, idx);
  else if (e.mode == NLT_CONV_K2S2) v = frag4<NLT_CONV_K2S2>(e, idx);
  else if (e.mode == NLT_CONV_K2S1) v = frag4<NLT_CONV_K2S1>(e, idx);
  else if (e.mode == NLT_DECONV_K2S2) v = frag4<NLT_DECONV_K2S2>(e, idx);
  else v = frag4<NLT_DECONV_K2S1>(e, idx);
  *reinterpret_cast<f32x4*>(e.dst + idx) = v;
}

}  // namespace

extern "C" int nlt_repack_weights(const nlt_repack_desc* descs_device, int n_desc, long total_blocks, void* stream) {
  if (!descs_device || n_desc <= 0 || total_blocks <= 0 || total_blocks >= (1l << 31)) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(repack_all_kernel, dim3((unsigned)total_blocks), dim3(64), 0, static_cast<hipStream_t>(stream),
                     descs_device, n_desc);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
