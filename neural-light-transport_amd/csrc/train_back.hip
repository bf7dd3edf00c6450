// Backward of the last expanding block + head: the training counterpart of back_kernel (fused.hip).
//   forward:  u = lrelu(Conv2DTranspose k2s2 (40 -> 4)(x | fm1)),  v = lrelu(Conv2DTranspose k2s1 (4 -> 4)(u)),
//             pred = v . Wh[0:4] + skip3                       (convnet.py:67-76,85 as models/nlt.py:182-195 runs them)
// One workgroup walks 8 x 16 half-resolution tiles (16 x 32 full-resolution texels) and keeps every weight-gradient
// sum in registers across its tiles; per tile
//   phase 1 (VALU)  dv = lrelu'(v) . (Wh[0:4] dpred) on the tile + 1 texel bottom/right (the adjoint of the transposed
//                   stride-1 conv reads (y + a, x + b)) -> LDS; sums for dWh[0:4], dbh, db_s1
//   phase 2 (VALU)  du = lrelu'(u) . sum_{a,b} W_s1[a,b]^T dv(y + a, x + b) -> LDS; sums for dW_s1 (u at (y - a, x - b)), db_s2
//   phase 3 (MFMA)  dW_s2^T[(a,b,o), c] += sum_texels du[2i+a,2j+b,o] in[i,j,c]   (K = 4 half-resolution texels per step)
//                   d_in[i,j,c] = sum_{a,b,o} W_s2[a,b,o,c] du[2i+a,2j+b,o]         (permuted K: lane group = tap, one
//                   16-byte LDS read feeds the four k-steps) -> dx (8 channels, times lrelu'(x)) | dfm1 (32 channels)
// Replaces, of the unfused backward plan (engine.py): bwd.head, bwd.L12.q.s1.{act,wgrad,dgrad} and
// bwd.L12.q.s2.{wgrad,dgrad.x,dgrad.skip}: seven launches over 4-channel full-resolution tensors (the weight-gradient
// kernel alone took 0.43 ms on the 4 -> 4 layer at 4 x 1024^2).  Deterministic: per-workgroup partial sums, then a
// fixed-order reduction.
#include "nlt_common.h"

namespace {

constexpr int BTH = 8, BTW = 16;                                       // half-resolution tile
constexpr int BFH = 2 * BTH, BFW = 2 * BTW;                            // 16 x 32 full-resolution texels
constexpr int BB_DW2 = 0, BB_DW1 = 640, BB_DWH = 704, BB_DBH = 716, BB_DB1 = 719, BB_DB2 = 723, BB_TOT = 728;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// PARTS: 3 = everything, the only form launched.  (r05 also launched 1 = backward-data only on the chain + 2 = the weight / bias
// sums on the weight-gradient stream: 3.16 vs 3.06 ms per l2 step -- the second launch re-reads (v, dpred, u) beside the chain's
// full-resolution launches -- so the split entry point was removed in r06; profiles/README.md r05.)
template <int PARTS>
__global__ __launch_bounds__(256) void back_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ fm1, const float* __restrict__ u, const float* __restrict__ v,
    const float* __restrict__ dpred, int h2, int w2, int tiles_y, int tiles_x, long tiles,
    const float* __restrict__ w_s2, const float* __restrict__ w_s1, const float* __restrict__ w_head, float alpha,
    float* __restrict__ dx, float* __restrict__ dfm1, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) float dvp[(BFH + 1) * (BFW + 1) * 4];
  __shared__ __attribute__((aligned(16))) float dup[BFH * BFW * 4];
  __shared__ __attribute__((aligned(16))) float utile[(BFH + 1) * (BFW + 1) * 4];   // u with its top / left halo: u(Y0 - 1 + ly, X0 - 1 + lx)
  __shared__ float red[4][BB_TOT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int h = 2 * h2, w = 2 * w2;

  constexpr bool DG = (PARTS & 1) != 0, WG = (PARTS & 2) != 0;
  float wa[3][4];                                                      // dgrad A operands: W_s2[n = 4 kk + ks][c = 16 mt + j]
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 16 * mt + j;
      wa[mt][ks] = (DG && c < 40) ? w_s2[(4 * kk + ks) * 40 + c] : 0.f;
    }
  float wh[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) wh[e] = w_head[e];

  float acch[12], accbh[3], accb1[4], accb2[4];
  f32x4 acc1m = (f32x4){0.f, 0.f, 0.f, 0.f};                            // dW_s1 on the matrix pipe: rows (tap, c), columns o
#pragma unroll
  for (int e = 0; e < 12; ++e) acch[e] = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { accb1[e] = 0.f; accb2[e] = 0.f; }
  accbh[0] = accbh[1] = accbh[2] = 0.f;
  f32x4 accw2[3] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int tx0 = (int)(tile % tiles_x) * BTW;
    const long tr = tile / tiles_x;
    const int ty0 = (int)(tr % tiles_y) * BTH, f = (int)(tr / tiles_y);
    const int Y0 = 2 * ty0, X0 = 2 * tx0;

    // ---- every global load of the tile is issued HERE, before the first barrier: the three phases used to fetch their own
    // operands one after the other, and with two workgroups per CU the kernel spent 74 % of its wave-cycles waiting for
    // memory (PMC: SQ_WAIT_ANY) at 2 TB/s.  (v, dpred) -> phase 1; the u tile with its top / left halo -> LDS -> phase 2 and
    // the dW_s1 MFMAs; x | fm1 -> phase 3.
    f32x4 pv[3], pu[3];
    float pg[3][3];
    bool in_v[3], in_u[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int e = threadIdx.x + 256 * it;
      const int ec = e < (BFH + 1) * (BFW + 1) ? e : 0;
      const int ly = ec / (BFW + 1), lx = ec - ly * (BFW + 1);
      const int y = Y0 + ly, xg = X0 + lx;
      in_v[it] = e < (BFH + 1) * (BFW + 1) && y < h && xg < w;
      const long tex = in_v[it] ? ((long)f * h + y) * w + xg : 0;
      pv[it] = *reinterpret_cast<const f32x4*>(v + tex * 4);
      pg[it][0] = dpred[tex * 3]; pg[it][1] = dpred[tex * 3 + 1]; pg[it][2] = dpred[tex * 3 + 2];
      const int yu = y - 1, xu = xg - 1;
      in_u[it] = e < (BFH + 1) * (BFW + 1) && yu >= 0 && xu >= 0 && yu < h && xu < w;
      const long texu = in_u[it] ? ((long)f * h + yu) * w + xu : 0;
      pu[it] = *reinterpret_cast<const f32x4*>(u + texu * 4);
    }
    float pb[8][3];                                                    // phase 3a's B operands (x | fm1 channels 16 mt + j of texel (gi, kk))
    const int na = j >> 3, nb = (j >> 2) & 1, no = j & 3;
    if constexpr (WG) {
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) {
      const int gi = wave + 4 * g8;
      const int ii = gi >> 2, jj = (gi & 3) * 4 + kk;
      const int gy = ty0 + ii, gx = tx0 + jj;
      const bool inside = gy < h2 && gx < w2;
      const long tex2 = inside ? ((long)f * h2 + gy) * w2 + gx : 0;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        const int c = 16 * mt + j;
        const float vx = x[tex2 * 8 + (c < 8 ? c : 0)];
        const float vf = fm1[tex2 * 32 + ((c >= 8 && c < 40) ? c - 8 : 0)];
        pb[g8][mt] = !inside ? 0.f : (c < 8 ? vx : (c < 40 ? vf : 0.f));
      }
    }
    }
    f32x4 pxv[2];                                                      // phase 3b: x at channels 4 kk (kk < 2) of texel (ty0 + wave + 4 r, tx0 + j)
    if constexpr (DG) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int gy = ty0 + wave + 4 * r, gx = tx0 + j;
      const bool inside = gy < h2 && gx < w2 && kk < 2;
      const long tex2 = inside ? ((long)f * h2 + gy) * w2 + gx : 0;
      pxv[r] = *reinterpret_cast<const f32x4*>(x + tex2 * 8 + (kk < 2 ? 4 * kk : 0));
    }
    }

    // ---- phase 1
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int e = threadIdx.x + 256 * it;
      if (e >= (BFH + 1) * (BFW + 1)) break;
      const int ly = e / (BFW + 1), lx = e - ly * (BFW + 1);
      const int y = Y0 + ly, xg = X0 + lx;
      f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (in_v[it]) {
        const f32x4 vv = pv[it];
        float g[3] = {pg[it][0], pg[it][1], pg[it][2]};
        if ((y | xg) == 0) { g[0] = 0.f; g[1] = 0.f; g[2] = 0.f; }   // set_left_top_corner
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const float dl = wh[o * 3] * g[0] + wh[o * 3 + 1] * g[1] + wh[o * 3 + 2] * g[2];
          d[o] = vv[o] > 0.f ? dl : alpha * dl;
        }
        if (WG && ly < BFH && lx < BFW) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int o = 0; o < 3; ++o) acch[c * 3 + o] = fmaf(vv[c], g[o], acch[c * 3 + o]);
            accb1[c] += d[c];
          }
          accbh[0] += g[0]; accbh[1] += g[1]; accbh[2] += g[2];
        }
      }
      *reinterpret_cast<f32x4*>(dvp + e * 4) = d;
      *reinterpret_cast<f32x4*>(utile + e * 4) = in_u[it] ? pu[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    // ---- phase 2
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int e = threadIdx.x + 256 * rep;
      const int ly = e >> 5, lx = e & 31;
      const int y = Y0 + ly, xg = X0 + lx;
      f32x4 dp = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (y < h && xg < w) {
        const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(utile + ((ly + 1) * (BFW + 1) + lx + 1) * 4);
        f32x4 du = zero;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 dv = *reinterpret_cast<const f32x4*>(dvp + ((ly + (t >> 1)) * (BFW + 1) + lx + (t & 1)) * 4);
#pragma unroll
          for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) du[c] = fmaf(w_s1[(t * 4 + o) * 4 + c], dv[o], du[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          dp[c] = u0[c] > 0.f ? du[c] : alpha * du[c];
          if (WG) accb2[c] += dp[c];
        }
      }
      *reinterpret_cast<f32x4*>(dup + e * 4) = dp;
    }
    // dW_s1[t][o][c] += sum_texels u(y - a, x - b)[c] dv(y, x)[o]: rows (t, c) = lane & 15, columns o = lane & 15 (< 4), K = 4
    // texels per MFMA; the wave takes a quarter of the tile (dv is zero outside the image, so clipped tiles need no mask)
    if constexpr (WG) {
      const int ta = (j >> 3) & 1, tb = (j >> 2) & 1, cc = j & 3;
#pragma unroll 4
      for (int m = 0; m < 32; ++m) {
        const int k = wave * 128 + 4 * m + kk;
        const int ly = k >> 5, lx = k & 31;
        const float av = utile[((ly + 1 - ta) * (BFW + 1) + lx + 1 - tb) * 4 + cc];
        const float bv = j < 4 ? dvp[(ly * (BFW + 1) + lx) * 4 + j] : 0.f;
        acc1m = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc1m, 0, 0, 0);
      }
    }
    __syncthreads();

    // ---- phase 3a: dW_s2^T, rows n = (a, b, o), columns c, K = half-resolution texels
    if constexpr (WG) {
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) {
      const int gi = wave + 4 * g8;
      const int ii = gi >> 2, jj = (gi & 3) * 4 + kk;
      const float av = dup[((2 * ii + na) * BFW + 2 * jj + nb) * 4 + no];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) accw2[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, pb[g8][mt], accw2[mt], 0, 0, 0);
    }
    }
    // ---- phase 3b: d_in = W_s2^T du, 16 texels (one tile row) per MFMA column block
    if constexpr (DG) {
    for (int ii = wave; ii < BTH; ii += 4) {
      const int gy = ty0 + ii, gx = tx0 + j;
      const bool inside = gy < h2 && gx < w2;
      const f32x4 bvec = *reinterpret_cast<const f32x4*>(dup + ((2 * ii + (kk >> 1)) * BFW + 2 * j + (kk & 1)) * 4);
      const long tex2 = ((long)f * h2 + gy) * w2 + gx;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][ks], bvec[ks], acc, 0, 0, 0);
        const int c0 = 16 * mt + 4 * kk;
        if (inside) {
          if (c0 < 8) {                                                // x is the previous block's LeakyReLU output: hand back the
            const f32x4 xv = pxv[(ii - wave) >> 2];                       // gradient w.r.t. its PRE-activation
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = xv[e] > 0.f ? acc[e] : alpha * acc[e];
            *reinterpret_cast<f32x4*>(dx + tex2 * 8 + c0) = acc;
          } else if (c0 < 40) {
            *reinterpret_cast<f32x4*>(dfm1 + tex2 * 32 + c0 - 8) = acc;
          }
        }
      }
    }
    }
    __syncthreads();
  }
  if constexpr (!WG) return;

  // ---- block partial sums -> workspace
  float* r = red[wave];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      if (16 * mt + j < 40) r[BB_DW2 + (4 * kk + rr) * 40 + 16 * mt + j] = accw2[mt][rr];
  if (j < 4) {                                                         // D rows 4 kk + rr = (t, c), column j = o  ->  [(t * 4 + o) * 4 + c]
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int i = 4 * kk + rr;
      r[BB_DW1 + ((i >> 2) * 4 + j) * 4 + (i & 3)] = acc1m[rr];
    }
  }
#pragma unroll
  for (int e = 0; e < 12; ++e) { const float s = wave_sum(acch[e]); if (lane == 0) r[BB_DWH + e] = s; }
#pragma unroll
  for (int e = 0; e < 3; ++e) { const float s = wave_sum(accbh[e]); if (lane == 0) r[BB_DBH + e] = s; }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s1 = wave_sum(accb1[e]), s2 = wave_sum(accb2[e]);
    if (lane == 0) { r[BB_DB1 + e] = s1; r[BB_DB2 + e] = s2; }
  }
  if (lane == 0) r[BB_TOT - 1] = 0.f;
  __syncthreads();
  for (int idx = threadIdx.x; idx < BB_TOT; idx += 256)
    ws[(long)blockIdx.x * BB_TOT + idx] = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
}

struct BackBwdOut { float *dw_s2, *db_s2, *dw_s1, *db_s1, *dw_head, *db_head; };

__global__ __launch_bounds__(256) void back_bwd_reduce_kernel(const float* __restrict__ ws, int nblocks, BackBwdOut out) {
  __shared__ float part[16][17];                                       // 16 entries x 16 row groups (x 4 running sums), fixed order
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int bl = g;
  if (idx < BB_TOT - 1) {
    for (; bl + 48 < nblocks; bl += 64) {
      s0 += ws[(long)bl * BB_TOT + idx];
      s1 += ws[(long)(bl + 16) * BB_TOT + idx];
      s2 += ws[(long)(bl + 32) * BB_TOT + idx];
      s3 += ws[(long)(bl + 48) * BB_TOT + idx];
    }
    for (; bl < nblocks; bl += 16) s0 += ws[(long)bl * BB_TOT + idx];
  }
  part[g][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g || idx >= BB_TOT - 1) return;
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) t += part[q][e];
  if (idx < BB_DW1) out.dw_s2[idx] += t;
  else if (idx < BB_DWH) out.dw_s1[idx - BB_DW1] += t;
  else if (idx < BB_DBH) out.dw_head[idx - BB_DWH] += t;
  else if (idx < BB_DB1) out.db_head[idx - BB_DBH] += t;
  else if (idx < BB_DB2) out.db_s1[idx - BB_DB1] += t;
  else out.db_s2[idx - BB_DB2] += t;
}

long back_bwd_blocks(int n, int h2, int w2) {
  const long tiles = (long)n * ((h2 + BTH - 1) / BTH) * ((w2 + BTW - 1) / BTW);
  return tiles < 1024 ? tiles : 1024;
}

}  // namespace

extern "C" long nlt_back_backward_workspace_floats(int n, int h2, int w2) {
  if (n <= 0 || h2 <= 0 || w2 <= 0) return -1;
  return back_bwd_blocks(n, h2, w2) * BB_TOT;
}

namespace {
int back_backward_launch(const float* x, const float* fm1, const float* u, const float* v, const float* dpred,
                         int n, int h2, int w2, const float* w_s2, const float* w_s1, const float* w_head,
                         float alpha, float* dx, float* dfm1, float* dw_s2, float* db_s2, float* dw_s1,
                         float* db_s1, float* dw_head, float* db_head, float* workspace, void* stream) {
  constexpr int parts = 3;
  if (!x || !fm1 || !u || !v || !dpred || !w_s2 || !w_s1 || !w_head) return NLT_ERR_BAD_ARG;
  if ((parts & 1) && (!dx || !dfm1)) return NLT_ERR_BAD_ARG;
  if ((parts & 2) && (!dw_s2 || !db_s2 || !dw_s1 || !db_s1 || !dw_head || !db_head || !workspace)) return NLT_ERR_BAD_ARG;
  if (n <= 0 || h2 <= 0 || w2 <= 0) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(u) || !nlt_aligned16(v) || ((parts & 1) && (!nlt_aligned16(dx) || !nlt_aligned16(dfm1)))) return NLT_ERR_BAD_ARG;
  if ((long long)n * h2 * w2 * 4 * 8 >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ty = (h2 + BTH - 1) / BTH, tx = (w2 + BTW - 1) / BTW;
  const long tiles = (long)n * ty * tx;
  const int blocks = (int)back_bwd_blocks(n, h2, w2);
#define NLT_BB(P_) hipLaunchKernelGGL(back_bwd_kernel<P_>, dim3(blocks), dim3(256), 0, s, x, fm1, u, v, dpred, h2, w2, ty, tx, tiles, \
                                      w_s2, w_s1, w_head, alpha, dx, dfm1, workspace)
  NLT_BB(3);
#undef NLT_BB
  NLT_CHECK_LAUNCH();
  if (parts & 2) {
    BackBwdOut out = {dw_s2, db_s2, dw_s1, db_s1, dw_head, db_head};
    hipLaunchKernelGGL(back_bwd_reduce_kernel, dim3((BB_TOT + 15) / 16), dim3(256), 0, s, workspace, blocks, out);
    NLT_CHECK_LAUNCH();
  }
  return NLT_OK;
}
}  // namespace

extern "C" int nlt_back_backward(const float* x, const float* fm1, const float* u, const float* v, const float* dpred,
                                 int n, int h2, int w2, const float* w_s2, const float* w_s1, const float* w_head,
                                 float alpha, float* dx, float* dfm1, float* dw_s2, float* db_s2, float* dw_s1,
                                 float* db_s1, float* dw_head, float* db_head, float* workspace, void* stream) {
  return back_backward_launch(x, fm1, u, v, dpred, n, h2, w2, w_s2, w_s1, w_head, alpha, dx, dfm1, dw_s2, db_s2, dw_s1, db_s1, dw_head,
                              db_head, workspace, stream);
}
