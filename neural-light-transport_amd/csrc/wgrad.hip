// Weight / bias gradients of every conv family:  dW[k][n] = sum_rows X[row][k] * dP[row][n],
// db[n] = sum_rows dP[row][n], where X is the implicit-GEMM gather of the forward pass (taps x
// virtual-concat sources) and dP the gradient w.r.t. the layer's PRE-activation output.
//
// MFMA path (v_mfma_f32_16x16x4_f32, exact fp32): the reduction runs over texel rows, 4 per
// MFMA step.  A = X^T (lane (i = l&15, mm = l>>4) holds X[row m0+mm][k0+i]), B = dP (lane holds
// dP[row m0+mm][n0+(l&15)]); a wave owns a KT x NT block of 16x16 dW tiles for one slice of the
// rows and adds its partial sums into the Keras-layout gradient with fp32 atomics (the slices of
// one tile land on different XCDs; device-scope atomics at L2 make that placement-independent).
#include "nlt_common.h"

namespace {

__host__ __device__ inline int chunks16(int c) { return (c + 15) >> 4; }

template <int MODE>
__device__ __forceinline__ int keras_widx(int t, int c, int ncol, int cin, int cout) {
  if (MODE == NLT_CONV1X1 || MODE == NLT_CONV_K2S2 || MODE == NLT_CONV_K2S1) return (t * cin + c) * cout + ncol;
  if (MODE == NLT_DECONV_K2S1) return (t * cout + ncol) * cin + c;
  return ncol * cin + c;   // DECONV_K2S2: ncol = (a*2+b)*cout + o
}

struct WgradP {
  ConvP c;            // geometry + X sources (src0/src1); wgt/bias/out unused
  const float* dp;    // gradient w.r.t. pre-activation output, [n, oh, ow, *] stride ldp
  int ldp;
  float* dw;          // Keras-layout weight gradient (accumulated into)
  float* db;          // bias gradient (accumulated into), may be null
  int rows_per_split; // multiple of 4
};

template <int MODE, int KT, int NT>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradP w, int kgroups, int ngroups, int msplits) {
  constexpr int TAPS = ConvTraits<MODE>::TAPS;
  const ConvP& p = w.c;
  const int lane = threadIdx.x & 63;
  int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int total = kgroups * ngroups * msplits;
  if (wave >= total) return;
  const int ms = wave % msplits; wave /= msplits;
  const int ng = wave % ngroups;
  const int kg = wave / ngroups;
  const int li = lane & 15, mm = lane >> 4;
  const int ch0 = chunks16(p.c0), ch1 = chunks16(p.c1);
  const int cps = ch0 + ch1;
  const int cin = p.c0 + p.c1;

  // this wave's K tiles: (tap, source, channel offset) each
  int kt_tap[KT], kt_c[KT], kt_cs[KT];
  bool kt_src1[KT], kt_ok[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int tile = kg * KT + kt;
    kt_ok[kt] = tile < TAPS * cps;
    const int tl = kt_ok[kt] ? tile : 0;
    kt_tap[kt] = tl / cps;
    const int r = tl % cps;
    kt_src1[kt] = r >= ch0;
    kt_c[kt] = (kt_src1[kt] ? r - ch0 : r) * 16 + li;
    kt_cs[kt] = kt_src1[kt] ? p.c1 : p.c0;
    kt_ok[kt] = kt_ok[kt] && kt_c[kt] < kt_cs[kt];
  }
  // this wave's N tiles
  int nt_col[NT], nt_oc[NT], nt_ab[NT];
  bool nt_ok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    nt_col[nt] = (ng * NT + nt) * 16 + li;
    nt_ok[nt] = nt_col[nt] < p.N;
    const int col = nt_ok[nt] ? nt_col[nt] : 0;
    nt_ab[nt] = (MODE == NLT_DECONV_K2S2) ? col / p.cout : 0;
    nt_oc[nt] = (MODE == NLT_DECONV_K2S2) ? col - nt_ab[nt] * p.cout : col;
  }

  f32x4 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[kt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  const int m_begin = ms * w.rows_per_split;
  int m_end = m_begin + w.rows_per_split;
  if (m_end > p.M) m_end = p.M;
  for (int m0 = m_begin; m0 < m_end; m0 += 4) {
    const int m = m0 + mm;
    const bool rv = m < m_end;
    const int mc = rv ? m : m_begin;
    const int x = mc % p.gw;
    const int y = (mc / p.gw) % p.gh;
    const int f = mc / (p.gw * p.gh);
    float a[KT], b[NT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int tex = conv_tap_texel<MODE>(p, f, y, x, kt_tap[kt]);
      const bool ok = rv && kt_ok[kt] && tex >= 0;
      const size_t tx = tex >= 0 ? (size_t)tex : 0;
      const float* src = kt_src1[kt] ? p.src1 + tx * p.ld1 : p.src0 + tx * p.ld0;
      const float v = src[kt_ok[kt] ? kt_c[kt] : 0];
      a[kt] = ok ? v : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      size_t otex = (size_t)mc;
      if (MODE == NLT_DECONV_K2S2)
        otex = ((size_t)f * p.oh + 2 * y + (nt_ab[nt] >> 1)) * p.ow + 2 * x + (nt_ab[nt] & 1);
      const float v = w.dp[otex * w.ldp + nt_oc[nt]];
      b[nt] = (rv && nt_ok[nt]) ? v : 0.f;
      bsum[nt] += b[nt];
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kt], b[nt], acc[kt][nt], 0, 0, 0);
  }

  // D[row = mm*4 + r][col = li]: row = k index inside the tile, col = n index inside the tile
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int tile = kg * KT + kt;
    if (tile >= TAPS * cps) continue;
    const int t = tile / cps;
    const int r0 = tile % cps;
    const bool s1 = r0 >= ch0;
    const int cbase = (s1 ? r0 - ch0 : r0) * 16 + mm * 4;
    const int cs = s1 ? p.c1 : p.c0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (!nt_ok[nt]) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = cbase + r;
        if (cl >= cs) continue;
        atomicAdd(w.dw + keras_widx<MODE>(t, (s1 ? p.c0 : 0) + cl, nt_col[nt], cin, p.cout), acc[kt][nt][r]);
      }
    }
  }
  if (w.db && kg == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float s = bsum[nt];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (mm == 0 && nt_ok[nt]) atomicAdd(w.db + nt_oc[nt], s);
    }
  }
}

template <int MODE, int KT, int NT>
int launch_tile(WgradP& w, hipStream_t s) {
  const ConvP& p = w.c;
  const int ktiles = ConvTraits<MODE>::TAPS * (chunks16(p.c0) + chunks16(p.c1));
  const int ntiles = (p.N + 15) >> 4;
  const int kgroups = (ktiles + KT - 1) / KT;
  const int ngroups = (ntiles + NT - 1) / NT;
  // enough row slices for ~4 waves per SIMD, at least 64 rows (16 MFMA steps) per slice
  long want = 4096 / ((long)kgroups * ngroups);
  if (want < 1) want = 1;
  long rows = (p.M + want - 1) / want;
  if (rows < 64) rows = 64;
  rows = (rows + 3) & ~3L;
  const int msplits = (int)((p.M + rows - 1) / rows);
  w.rows_per_split = (int)rows;
  const long waves = (long)kgroups * ngroups * msplits;
  hipLaunchKernelGGL((wgrad_mfma_kernel<MODE, KT, NT>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                     w, kgroups, ngroups, msplits);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

template <int MODE>
int launch_mode(WgradP& w, hipStream_t s) {
  const ConvP& p = w.c;
  const int ktiles = ConvTraits<MODE>::TAPS * (chunks16(p.c0) + chunks16(p.c1));
  const int ntiles = (p.N + 15) >> 4;
  const int KT = ktiles >= 4 ? 4 : (ktiles >= 2 ? 2 : 1);
  const int NT = ntiles >= 4 ? 4 : (ntiles >= 2 ? 2 : 1);
#define NLT_WT(K, N) if (KT == K && NT == N) return launch_tile<MODE, K, N>(w, s);
  NLT_WT(4, 4) NLT_WT(4, 2) NLT_WT(4, 1) NLT_WT(2, 4) NLT_WT(2, 2) NLT_WT(2, 1) NLT_WT(1, 4) NLT_WT(1, 2) NLT_WT(1, 1)
#undef NLT_WT
  return NLT_ERR_UNSUPPORTED;
}

// Direct fallback for channel counts that are not multiples of 4 (e.g. a 5-channel input):
// one thread per row, per-block partial sums in LDS, then global atomics.  K*N <= 4096.
template <int MODE>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WgradP w, int KN) {
  constexpr int TAPS = ConvTraits<MODE>::TAPS;
  extern __shared__ __attribute__((aligned(16))) float part[];
  const ConvP& p = w.c;
  const int cin = p.c0 + p.c1;
  for (int i = threadIdx.x; i < KN + p.N; i += blockDim.x) part[i] = 0.f;
  __syncthreads();
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < p.M) {
    const int x = m % p.gw;
    const int y = (m / p.gw) % p.gh;
    const int f = m / (p.gw * p.gh);
    for (int ncol = 0; ncol < p.N; ++ncol) {
      size_t otex = (size_t)m;
      int oc = ncol;
      if (MODE == NLT_DECONV_K2S2) {
        const int ab = ncol / p.cout; oc = ncol - ab * p.cout;
        otex = ((size_t)f * p.oh + 2 * y + (ab >> 1)) * p.ow + 2 * x + (ab & 1);
      }
      const float g = w.dp[otex * w.ldp + oc];
      atomicAdd(&part[KN + ncol], g);
      for (int t = 0; t < TAPS; ++t) {
        const int tex = conv_tap_texel<MODE>(p, f, y, x, t);
        if (tex < 0) continue;
        for (int c = 0; c < cin; ++c) {
          const float xv = c < p.c0 ? p.src0[(size_t)tex * p.ld0 + c] : p.src1[(size_t)tex * p.ld1 + (c - p.c0)];
          atomicAdd(&part[keras_widx<MODE>(t, c, ncol, cin, p.cout)], xv * g);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < KN; i += blockDim.x) atomicAdd(w.dw + i, part[i]);
  if (w.db)
    for (int i = threadIdx.x; i < p.N; i += blockDim.x)
      atomicAdd(w.db + (MODE == NLT_DECONV_K2S2 ? i % p.cout : i), part[KN + i]);
}

template <int MODE>
int launch_direct(WgradP& w, hipStream_t s) {
  const ConvP& p = w.c;
  const int KN = ConvTraits<MODE>::TAPS * (p.c0 + p.c1) * p.N;
  if (KN + p.N > 8192) return NLT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(wgrad_direct_kernel<MODE>, dim3((unsigned)((p.M + 255) / 256)), dim3(256),
                     (size_t)(KN + p.N) * sizeof(float), s, w, KN);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

}  // namespace

extern "C" int nlt_conv_backward_weights(int mode, int algo,
                                         const float* src0, int ld0, int c0,
                                         const float* src1, int ld1, int c1,
                                         int n, int h, int w,
                                         const float* dpre, int ldp, int cout,
                                         float* dw_keras, float* dbias, void* stream) {
  if (!dpre || !dw_keras) return NLT_ERR_BAD_ARG;
  WgradP wp;
  // reuse the forward parameter validation (weights/bias/out pointers are placeholders here)
  const int st = nlt_fill_conv_params(wp.c, mode, src0, ld0, c0, src1, ld1, c1, n, h, w, dw_keras, dw_keras, cout,
                                      dw_keras, cout, 0, 0.f, nullptr, 0, 0);
  if (st != NLT_OK) return st;
  if (ldp < cout) return NLT_ERR_BAD_ARG;
  wp.dp = dpre; wp.ldp = ldp; wp.dw = dw_keras; wp.db = dbias; wp.rows_per_split = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool mfma_ok = !(c0 & 3) && !(c1 & 3) && !(cout & 3);
  if (algo == NLT_ALGO_AUTO) algo = mfma_ok ? NLT_ALGO_MFMA : NLT_ALGO_DIRECT;
  if (algo == NLT_ALGO_MFMA) {
    switch (mode) {
      case NLT_CONV1X1: return launch_mode<NLT_CONV1X1>(wp, s);
      case NLT_CONV_K2S2: return launch_mode<NLT_CONV_K2S2>(wp, s);
      case NLT_CONV_K2S1: return launch_mode<NLT_CONV_K2S1>(wp, s);
      case NLT_DECONV_K2S2: return launch_mode<NLT_DECONV_K2S2>(wp, s);
      case NLT_DECONV_K2S1: return launch_mode<NLT_DECONV_K2S1>(wp, s);
    }
  } else if (algo == NLT_ALGO_DIRECT) {
    switch (mode) {
      case NLT_CONV1X1: return launch_direct<NLT_CONV1X1>(wp, s);
      case NLT_CONV_K2S2: return launch_direct<NLT_CONV_K2S2>(wp, s);
      case NLT_CONV_K2S1: return launch_direct<NLT_CONV_K2S1>(wp, s);
      case NLT_DECONV_K2S2: return launch_direct<NLT_DECONV_K2S2>(wp, s);
      case NLT_DECONV_K2S1: return launch_direct<NLT_DECONV_K2S1>(wp, s);
    }
  }
  return NLT_ERR_BAD_ARG;
}
