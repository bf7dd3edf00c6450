// Direct (VALU) implicit-GEMM conv: one thread = one GEMM row (texel) x 4 GEMM columns.
// Handles every channel count; it is the path for the 3/4/8-channel full-resolution layers
// and the independent cross-check of the MFMA path.  Reads Keras-layout weights.
#include "nlt_common.h"

namespace {

template <int MODE>
__device__ __forceinline__ int keras_widx(const ConvP& p, int t, int c, int ncol) {
  const int cin = p.c0 + p.c1;
  if (MODE == NLT_CONV1X1 || MODE == NLT_CONV_K2S2 || MODE == NLT_CONV_K2S1)
    return (t * cin + c) * p.cout + ncol;                 // HWIO (kh,kw,Cin,Cout), t = a*2+b
  if (MODE == NLT_DECONV_K2S1)
    return (t * p.cout + ncol) * cin + c;                 // HWOI (kh,kw,Cout,Cin)
  return ncol * cin + c;                                  // DECONV_K2S2: ncol = (a*2+b)*Cout + o
}

template <int MODE>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvP p) {
  constexpr int TAPS = ConvTraits<MODE>::TAPS;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int n0 = blockIdx.y * 4;
  if (m >= p.M) return;
  const int x = m % p.gw;
  const int y = (m / p.gw) % p.gh;
  const int f = m / (p.gw * p.gh);

  float acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nc = n0 + j;
    acc[j] = (nc < p.N) ? p.bias[MODE == NLT_DECONV_K2S2 ? nc % p.cout : nc] : 0.f;
  }

  for (int t = 0; t < TAPS; ++t) {
    const int tex = conv_tap_texel<MODE>(p, f, y, x, t);
    if (tex < 0) continue;
    for (int s = 0; s < 2; ++s) {
      const int cs = s ? p.c1 : p.c0;
      if (cs == 0) continue;
      const float* src = (s ? p.src1 : p.src0) + (size_t)tex * (s ? p.ld1 : p.ld0);
      const int cbase = s ? p.c0 : 0;
      for (int c = 0; c < cs; ++c) {
        const float xv = src[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nc = n0 + j;
          if (nc < p.N) acc[j] = fmaf(xv, p.wgt[keras_widx<MODE>(p, t, cbase + c, nc)], acc[j]);
        }
      }
    }
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nc = n0 + j;
    if (nc >= p.N) continue;
    int otex, oc;
    if (MODE == NLT_DECONV_K2S2) {
      const int ab = nc / p.cout; oc = nc % p.cout;
      otex = (f * p.oh + 2 * y + (ab >> 1)) * p.ow + 2 * x + (ab & 1);
    } else { otex = m; oc = nc; }
    float v = acc[j];
    float* o = p.out + (size_t)otex * p.ldo + oc;
    if (p.accumulate) v += *o;
    if (p.mask_src) v *= (p.mask_src[(size_t)otex * p.ldm + oc] > 0.f) ? 1.f : p.alpha;
    else if (p.act) v = v > 0.f ? v : p.alpha * v;
    *o = v;
  }
}

template <int MODE>
int launch(const ConvP& p, hipStream_t s) {
  dim3 grid((p.M + 255) / 256, (p.N + 3) / 4);
  hipLaunchKernelGGL(conv_direct_kernel<MODE>, grid, dim3(256), 0, s, p);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

}  // namespace

int nlt_conv_direct_launch(int mode, const ConvP& p, hipStream_t s) {
  switch (mode) {
    case NLT_CONV1X1: return launch<NLT_CONV1X1>(p, s);
    case NLT_CONV_K2S2: return launch<NLT_CONV_K2S2>(p, s);
    case NLT_CONV_K2S1: return launch<NLT_CONV_K2S1>(p, s);
    case NLT_DECONV_K2S2: return launch<NLT_DECONV_K2S2>(p, s);
    case NLT_DECONV_K2S1: return launch<NLT_DECONV_K2S1>(p, s);
  }
  return NLT_ERR_BAD_ARG;
}
