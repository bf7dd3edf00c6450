// Fragment layouts of the packed conv weights, shared by the per-layer pack launches (conv_mfma.hip, conv_tile.hip)
// and the one-launch refresh of every packed buffer after an optimizer step (repack.hip).
#pragma once
#include "nlt_common.h"

__host__ __device__ inline int nlt_chunks16(int c) { return (c + 15) >> 4; }

// Keras array index of (tap t, input channel c, GEMM column ncol) for the conv family MODE, when the array handed in is
// the slice [lo, lo + cout) of a wider array along its COLUMN axis (`full` = that axis' extent in memory; full = cout,
// lo = 0 for an unsliced kernel).  Backward-data packs the ADJOINT family from the forward layer's own array, sliced
// along the forward-input axis (networks/elements.py packed_adjoint), without materialising the slice.
template <int MODE>
__device__ __forceinline__ long nlt_keras_widx(int t, int c, int ncol, int cin, int cout, int full, int lo) {
  if (MODE == NLT_CONV1X1 || MODE == NLT_CONV_K2S2 || MODE == NLT_CONV_K2S1) return ((long)t * cin + c) * full + lo + ncol;
  if (MODE == NLT_DECONV_K2S1) return ((long)t * full + lo + ncol) * cin + c;
  const int ab = ncol / cout, o = ncol - ab * cout;                    // DECONV_K2S2: ncol = (a*2+b)*cout + o
  return ((long)ab * full + lo + o) * cin + c;
}

// register-tiled MFMA kernel (conv_mfma.hip): [tap][16-channel chunk of (c0 | c1)][column tile][lane 64][s4]
template <int MODE>
__device__ __forceinline__ float nlt_mfma_fragment(const float* __restrict__ wk, long idx, int c0, int c1, int cout, int N,
                                                   int ntiles, int full, int lo) {
  const int s4 = idx & 3;
  const int lane = (idx >> 2) & 63;
  const long tile = idx >> 8;
  const int nt = tile % ntiles;
  const int kc = tile / ntiles;
  const int ch0 = nlt_chunks16(c0), ch1 = nlt_chunks16(c1);
  const int t = kc / (ch0 + ch1);
  const int r = kc % (ch0 + ch1);
  const int s = r >= ch0;
  const int cl = (s ? r - ch0 : r) * 16 + 4 * (lane >> 4) + s4;
  const int cs = s ? c1 : c0;
  const int ncol = nt * 16 + (lane & 15);
  if (cl < cs && ncol < N) return wk[nlt_keras_widx<MODE>(t, (s ? c0 : 0) + cl, ncol, c0 + c1, cout, full, lo)];
  return 0.f;
}

// LDS-tiled kernel (conv_tile.hip): [g = N / TN][cc = K / 16][tap 4][ct TNT][lane 64][s4], K = input channels of the executed
// conv, N = its output channels.  The Keras array handed in may be a slice [lo, lo + N) of the executed conv's OUTPUT-channel axis
// (`full` = that axis' extent in memory) and is indexed
//   normal     (kh,kw,K,N_full): a forward Conv2D, or backward-data of a Conv2DTranspose (its (kh,kw,Cout,Cin) array read as a conv
//                                from Cout to a slice of Cin);
//   transposed (kh,kw,N_full,K): a forward Conv2DTranspose, or backward-data of a Conv2D (its (kh,kw,Cin,Cout) array read as a
//                                transposed conv from Cout to a slice of Cin).
__device__ __forceinline__ float nlt_tile_fragment(const float* __restrict__ wk, long idx, int cin, int cout, int tnt,
                                                   int full, int lo, bool transposed) {
  const int s4 = idx & 3, lane = (idx >> 2) & 63;
  long r = idx >> 8;
  const int ct = r % tnt; r /= tnt;
  const int t = r & 3; r >>= 2;
  const int ncc = cin >> 4;
  const int cc = r % ncc;
  const int g = r / ncc;
  const int c = cc * 16 + 4 * (lane >> 4) + s4;
  const int o = (g * tnt + ct) * 16 + (lane & 15);
  return transposed ? wk[((long)t * full + lo + o) * cin + c] : wk[((long)t * cin + c) * full + lo + o];
}

// Transposed k2s2 mode of the LDS-tiled kernel (a 1x1-conv-shaped GEMM: K = input channels, N = 4 * cout columns (a, b, o)):
// [g = 4 cout / TN][cc = K / 16][ct TNT][lane 64][s4]; the Keras array is indexed (kh,kw,N_full,K) as for every transposed mode.
__device__ __forceinline__ float nlt_tile_fragment_d2(const float* __restrict__ wk, long idx, int cin, int cout, int tnt, int full, int lo) {
  const int s4 = idx & 3, lane = (idx >> 2) & 63;
  long r = idx >> 8;
  const int ct = r % tnt; r /= tnt;
  const int ncc = cin >> 4;
  const int cc = r % ncc;
  const int g = r / ncc;
  const int c = cc * 16 + 4 * (lane >> 4) + s4;
  const int col = (g * tnt + ct) * 16 + (lane & 15);
  const int ab = col / cout, o = col - ab * cout;
  return wk[((long)ab * full + lo + o) * cin + c];
}

// Winograd F(2x2, 2x2) kernel (conv_wino.hip): [g = N / TN][c8 = K / 8][position 9 = (xi, nu)][ct TNT][channel quad 2][column 16][e 4]
// = U = G g G^T of the executed correlation's 2 x 2 taps g, G = [1 0; 1 1; 0 1]: position (xi, nu) sums the taps (a, b) with
// a in {0} / {0, 1} / {1} for xi = 0 / 1 / 2 and b likewise for nu.  `transposed` (the Conv2DTranspose k2s1 family: a forward
// Conv2DTranspose or backward-data of a Conv2D) indexes the Keras array as (kh,kw,N_full,K) and FLIPS the taps:
// y[i,j] = sum_ab x[i-a,j-b] W[a,b] is the correlation of the window starting at (i-1, j-1) with g[a'][b'] = W[1-a'][1-b'].
__device__ __forceinline__ float nlt_wino_fragment(const float* __restrict__ wk, long idx, int cin, int cout, int tnt,
                                                   int full, int lo, bool transposed) {
  const int e = idx & 3, i = (idx >> 2) & 15, q = (idx >> 6) & 1;
  long r = idx >> 7;
  const int ct = r % tnt; r /= tnt;
  const int ps = r % 9; r /= 9;
  const int nc8 = cin >> 3;
  const int c8 = r % nc8;
  const int g = r / nc8;
  const int c = c8 * 8 + q * 4 + e;
  const int o = (g * tnt + ct) * 16 + i;
  const int xi = ps / 3, nu = ps - 3 * xi;
  float v = 0.f;
  for (int a = (xi == 2); a <= (xi != 0); ++a)
    for (int b = (nu == 2); b <= (nu != 0); ++b) {
      const int t = transposed ? (1 - a) * 2 + (1 - b) : a * 2 + b;
      v += transposed ? wk[((long)t * full + lo + o) * cin + c] : wk[((long)t * cin + c) * full + lo + o];
    }
  return v;
}
