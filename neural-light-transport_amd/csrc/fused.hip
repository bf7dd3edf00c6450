// Inference-only fused ends of the U-Net: everything that touches FULL-resolution texels.
//
//   front : L0 of both paths (1x1, linear) FOLDED into L1's stride-2 convs, L1's stride-1 convs and both
//           observation means, in one pass over the raw texel buffers (nlt/models/nlt.py:95-96,141-180 for
//           layers 0-1).  fm0 / obs0 / the L1 intermediates never exist in HBM; what leaves the kernel is
//           fm1 = [q1 | mean_k o1], the k per-observation maps o1, and skip3 = the output head's share of
//           the L0 features (+ base), 3 floats per texel.
//   back  : last expanding block (Conv2DTranspose k2s2 + k2s1), output head, + base, corner zero
//           (convnet.py:67-76,85; nlt.py:99-102,110), reading two half-resolution maps and skip3.
//
// Folding is exact algebra (L0 has no activation): W' = W_L0 * W_next, so results differ from the
// layer-by-layer kernels only by fp32 re-association (~1e-7 rel-L2, test tolerance 1e-4).  Training and
// the obs_override / obs_weights paths keep the unfused kernels.
//
// Both kernels tile the half-resolution grid 8 x 16 per 256-thread workgroup, flatten the haloed tile
// (9 x 17 texels) into 16-texel MFMA column tiles (v_mfma_f32_16x16x4_f32: weights = A, texels = B, so a
// lane ends up with 4 consecutive output channels of one texel), and hand the stride-2 results to the
// stride-1 stage through LDS.
#include "front_common.h"

namespace {

// ---------------------------------------------------------------------------------------
// plan-time weight folding + packing (one thread per blob float)
// Keras layouts: conv (kh,kw,Cin,Cout) -> ((tap*Cin)+c)*Cout+o.
// ---------------------------------------------------------------------------------------
struct FrontW {
  const float *wq0, *bq0, *wo0, *bo0;      // L0: (1,1,5,16), (1,1,3,16)
  const float *wqa, *bqa, *wqb, *bqb;      // query L1: (2,2,32,16) s2, (2,2,16,16) s1
  const float *woa, *boa, *wob, *bob;      // obs   L1: (2,2,16,16) s2, (2,2,16,16) s1
  const float *wh, *bh;                    // head (1,1,36,3): [dec 4 | q0 16 | mean o0 16]
};

__global__ void front_pack_kernel(FrontW w, float* __restrict__ blob) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= BLOB) return;
  float v = 0.f;
  if (idx < OFF_AO2) {                                          // AQ2[m][lane]
    const int m = idx >> 6, lane = idx & 63, tap = lane >> 4, o = lane & 15;
    for (int c = 0; c < 16; ++c)
      v = m < 5 ? fmaf(w.wq0[m * 16 + c], w.wqa[((tap * 32) + c) * 16 + o], v)
                : fmaf(w.wo0[(m - 5) * 16 + c], w.wqa[((tap * 32) + 16 + c) * 16 + o], v);
  } else if (idx < OFF_AQ1) {                                   // AO2[m][lane]
    const int r = idx - OFF_AO2, m = r >> 6, lane = r & 63, tap = lane >> 4, o = lane & 15;
    for (int c = 0; c < 16; ++c) v = fmaf(w.wo0[m * 16 + c], w.woa[((tap * 16) + c) * 16 + o], v);
  } else if (idx < OFF_BQ2) {                                   // AQ1 / AO1 [tap][lane][s4]
    const bool obs = idx >= OFF_AO1;
    const int r = idx - (obs ? OFF_AO1 : OFF_AQ1);
    const int s4 = r & 3, lane = (r >> 2) & 63, tap = r >> 8, kk = lane >> 4, o = lane & 15;
    v = (obs ? w.wob : w.wqb)[((tap * 16) + 4 * kk + s4) * 16 + o];
  } else if (idx < OFF_BO2) {                                   // BQ2
    const int o = idx - OFF_BQ2;
    v = w.bqa[o];
    for (int t = 0; t < 4; ++t)
      for (int c = 0; c < 16; ++c) {
        v = fmaf(w.bq0[c], w.wqa[((t * 32) + c) * 16 + o], v);
        v = fmaf(w.bo0[c], w.wqa[((t * 32) + 16 + c) * 16 + o], v);
      }
  } else if (idx < OFF_BQ1) {                                   // BO2
    const int o = idx - OFF_BO2;
    v = w.boa[o];
    for (int t = 0; t < 4; ++t)
      for (int c = 0; c < 16; ++c) v = fmaf(w.bo0[c], w.woa[((t * 16) + c) * 16 + o], v);
  } else if (idx < OFF_BO1) {
    v = w.bqb[idx - OFF_BQ1];
  } else if (idx < OFF_WSK) {
    v = w.bob[idx - OFF_BO1];
  } else if (idx < OFF_BSK) {                                   // WSKIP[r][o]
    const int r = (idx - OFF_WSK) / 3, o = (idx - OFF_WSK) % 3;
    for (int c = 0; c < 16; ++c)
      v = r < 5 ? fmaf(w.wq0[r * 16 + c], w.wh[(4 + c) * 3 + o], v)
                : fmaf(w.wo0[(r - 5) * 16 + c], w.wh[(20 + c) * 3 + o], v);
  } else if (idx < OFF_BSK + 3) {                               // BSKIP[o]
    const int o = idx - OFF_BSK;
    v = w.bh[o];
    for (int c = 0; c < 16; ++c) {
      v = fmaf(w.bq0[c], w.wh[(4 + c) * 3 + o], v);
      v = fmaf(w.bo0[c], w.wh[(20 + c) * 3 + o], v);
    }
  } else if (idx >= OFF_WSK8 && idx < OFF_WSK8 + 24) {          // WSKIP[r][o] / 255 (uint8-store front kernel)
    const int r = (idx - OFF_WSK8) / 3, o = (idx - OFF_WSK8) % 3;
    for (int c = 0; c < 16; ++c)
      v = r < 5 ? fmaf(w.wq0[r * 16 + c], w.wh[(4 + c) * 3 + o], v)
                : fmaf(w.wo0[(r - 5) * 16 + c], w.wh[(20 + c) * 3 + o], v);
    v *= 1.0f / 255.0f;
  }
  blob[idx] = v;
}

__global__ void front_pack_l2_kernel(const float* __restrict__ wq, const float* __restrict__ bq,
                                     const float* __restrict__ wo, const float* __restrict__ bo, float* __restrict__ blob) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= BLOB3) return;
  float v;
  if (idx < OFF3_AO) {
    const int e = idx & 3, lane = (idx >> 2) & 63, c8 = (idx >> 8) & 7, rt = idx >> 11;
    const int tap = lane >> 4, o = rt * 16 + (lane & 15), c = 16 * (c8 >> 2) + 4 * (c8 & 3) + e;
    v = wq[((tap * 32) + c) * 32 + o];
  } else if (idx < OFF3_BQ) {
    const int r = idx - OFF3_AO;
    const int e = r & 3, lane = (r >> 2) & 63, c4 = (r >> 8) & 3, rt = r >> 10;
    const int tap = lane >> 4, o = rt * 16 + (lane & 15);
    v = wo[((tap * 16) + 4 * c4 + e) * 32 + o];
  } else if (idx < OFF3_BO) v = bq[idx - OFF3_BQ];
  else v = bo[idx - OFF3_BO];
  blob[idx] = v;
}

// ---------------------------------------------------------------------------------------
// front kernel
// ---------------------------------------------------------------------------------------
// L2S2 = true (k <= 4): the workgroup also runs level 2's stride-2 convs on its 8 x 16 level-1 tile (a 4 x 8 tile of
// level 2; k2s2 needs no halo) and writes qtmp2 / otmp2 instead of the per-observation level-1 maps.
template <bool L2S2>
__global__ __launch_bounds__(256) void front_kernel(
    const float* __restrict__ base, const float* __restrict__ cvis, const float* __restrict__ lvis,
    const float* __restrict__ nn_rgb, const float* __restrict__ nn_base, int k, int h, int w,
    int tiles_y, int tiles_x, const float* __restrict__ blob, int add_base, float alpha,
    float* __restrict__ fm1, float* __restrict__ obs1, float* __restrict__ skip3,
    const float* __restrict__ blob3, float* __restrict__ qtmp2, float* __restrict__ otmp2,
    float* __restrict__ qsave, float* __restrict__ osave) {
  extern __shared__ __attribute__((aligned(16))) float lds[];          // [1 + k][4 kk][PLT][4]: q, then obs i
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int h2 = h >> 1, w2 = w >> 1;
  int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int tx0 = (tile % tiles_x) * TW; tile /= tiles_x;
  const int ty0 = (tile % tiles_y) * TH;
  const int f = tile / tiles_y;
  const long hw = (long)h * w;

  // ---- stage 1: folded stride-2 convs on the haloed tile, straight from the raw buffers
  float aq2[8], ao2[3];
#pragma unroll
  for (int m = 0; m < 8; ++m) aq2[m] = blob[OFF_AQ2 + m * 64 + lane];
#pragma unroll
  for (int m = 0; m < 3; ++m) ao2[m] = blob[OFF_AO2 + m * 64 + lane];
  const f32x4 bq2 = *reinterpret_cast<const f32x4*>(blob + OFF_BQ2 + 4 * kk);
  const f32x4 bo2 = *reinterpret_cast<const f32x4*>(blob + OFF_BO2 + 4 * kk);
  const float inv_k = 1.f / (float)k;
  // stage-2 fragments are requested now, so their L2 latency hides behind stage 1 (the compiler would
  // otherwise sink each load to its first use and serialise four round trips per workgroup)
  f32x4 aq1[4], ao1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    aq1[t] = *reinterpret_cast<const f32x4*>(blob + OFF_AQ1 + (t * 64 + lane) * 4);
    ao1[t] = *reinterpret_cast<const f32x4*>(blob + OFF_AO1 + (t * 64 + lane) * 4);
  }
  const f32x4 bq1 = *reinterpret_cast<const f32x4*>(blob + OFF_BQ1 + 4 * kk);
  const f32x4 bo1 = *reinterpret_cast<const f32x4*>(blob + OFF_BO1 + 4 * kk);

  constexpr int KU = 4;                                                // observations whose loads are issued together
  for (int mt = wave; mt < NT; mt += 4) {
    const int t = mt * 16 + j;
    const bool live = t < HT;
    const int hy = live ? t / HW : 0, hx = live ? t % HW : 0;
    const int gy = ty0 + hy, gx = tx0 + hx;
    const bool inside = live && gy < h2 && gx < w2;                    // beyond the image: the s1 conv's zero padding
    const bool owned = inside && hy < TH && hx < TW;
    const int fy = inside ? 2 * gy + (kk >> 1) : 0, fx = inside ? 2 * gx + (kk & 1) : 0;
    const long pix = (long)fy * w + fx;
    const long tex = (long)f * hw + pix;                               // this lane's full-resolution texel (tap kk)
    // Every load of the tile is issued before the first use (a wave keeps 2*KU + 3 requests in flight; waiting
    // per observation would leave the memory system idle).  The first observation group is peeled out of the
    // loop so that no loop-carried wait separates its loads from the query-input loads.
    float xs0 = 0.f, xs1 = 0.f, xs2 = 0.f;
    auto load_group = [&](int i0, float (&r)[KU][3], float (&b)[KU][3]) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const int i = i0 + u < k ? i0 + u : k - 1;
        const long ot = (((long)f * k + i) * hw + pix) * 3;
        r[u][0] = nn_rgb[ot]; r[u][1] = nn_rgb[ot + 1]; r[u][2] = nn_rgb[ot + 2];
        b[u][0] = nn_base[ot]; b[u][1] = nn_base[ot + 1]; b[u][2] = nn_base[ot + 2];
      }
    };
    auto compute_group = [&](int i0, const float (&r)[KU][3], const float (&b)[KU][3]) {
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        if (i0 + u < k) {                                              // wave-uniform
          const float d0 = r[u][0] - b[u][0], d1 = r[u][1] - b[u][1], d2 = r[u][2] - b[u][2];
          xs0 += d0; xs1 += d1; xs2 += d2;
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ao2[0], d0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ao2[1], d1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ao2[2], d2, acc, 0, 0, 0);
          acc = lrelu4(acc + bo2, alpha);
          if (!inside) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (live) *reinterpret_cast<f32x4*>(lds + (size_t)(1 + i0 + u) * PATH + (kk * PLT + t) * 4) = acc;
          if (osave && owned)                                          // training: backward needs the stride-2 outputs
            *reinterpret_cast<f32x4*>(osave + ((((long)f * k + i0 + u) * h2 + gy) * w2 + gx) * 16 + 4 * kk) = acc;
        }
      }
    };
    float raw[8];
    {
      float r[KU][3], b[KU][3];
      load_group(0, r, b);
      raw[0] = base[tex * 3]; raw[1] = base[tex * 3 + 1]; raw[2] = base[tex * 3 + 2];
      raw[3] = cvis[tex]; raw[4] = lvis[tex];
      compute_group(0, r, b);
    }
    for (int i0 = KU; i0 < k; i0 += KU) {
      float r[KU][3], b[KU][3];
      load_group(i0, r, b);
      compute_group(i0, r, b);
    }
    raw[5] = xs0 * inv_k; raw[6] = xs1 * inv_k; raw[7] = xs2 * inv_k;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq2[m], raw[m], acc, 0, 0, 0);
    acc = lrelu4(acc + bq2, alpha);
    if (!inside) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live) *reinterpret_cast<f32x4*>(lds + (kk * PLT + t) * 4) = acc;
    if (qsave && owned) *reinterpret_cast<f32x4*>(qsave + (((long)f * h2 + gy) * w2 + gx) * 16 + 4 * kk) = acc;
    if (owned) {                                                       // the head's share of the L0 features (+ base)
      float s0 = blob[OFF_BSK], s1 = blob[OFF_BSK + 1], s2 = blob[OFF_BSK + 2];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        s0 = fmaf(raw[rr], blob[OFF_WSK + rr * 3], s0);
        s1 = fmaf(raw[rr], blob[OFF_WSK + rr * 3 + 1], s1);
        s2 = fmaf(raw[rr], blob[OFF_WSK + rr * 3 + 2], s2);
      }
      if (add_base) { s0 += raw[0]; s1 += raw[1]; s2 += raw[2]; }
      skip3[tex * 3] = s0; skip3[tex * 3 + 1] = s1; skip3[tex * 3 + 2] = s2;
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(aq1[t]), "v"(ao1[t]));   // keep them resident (see above)
  asm volatile("" ::"v"(bq1), "v"(bo1));
  __syncthreads();

  // ---- stage 2: stride-1 convs (TF 'same': taps (y+a, x+b), zero beyond bottom/right) + observation mean.
  // A wave owns tile rows `wave` and `wave + 4` and runs them side by side: two independent accumulators keep
  // the matrix pipe issuing every 32 cycles (one accumulator alone pays the 40-cycle dependent latency).
  const long hw2 = (long)h2 * w2;
  static_assert(TH == 8, "stage 2 pairs tile rows wave and wave + 4");
  {
    const int gx = tx0 + j;
    const int gy[2] = {ty0 + wave, ty0 + wave + 4};
    const bool inside[2] = {gy[0] < h2 && gx < w2, gy[1] < h2 && gx < w2};
    const long otex[2] = {(long)gy[0] * w2 + gx, (long)gy[1] * w2 + gx};
    f32x4 mean[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    f32x4 qv[2] = {mean[0], mean[0]};
    f32x4 o1[L2S2 ? 4 : 1][2];                                         // L2S2: the observations' level-1 outputs stay in registers
    auto path = [&](int p, f32x4 (&out)[2]) {
      const float* tilep = lds + (size_t)p * PATH + kk * PLT * 4;
      f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(tilep + ((wave + (t >> 1)) * HW + j + (t & 1)) * 4);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(tilep + ((wave + 4 + (t >> 1)) * HW + j + (t & 1)) * 4);
        const f32x4 a = p ? ao1[t] : aq1[t];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], b0[s4], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], b1[s4], acc[1], 0, 0, 0);
        }
      }
      out[0] = lrelu4(acc[0] + (p ? bo1 : bq1), alpha);
      out[1] = lrelu4(acc[1] + (p ? bo1 : bq1), alpha);
    };
    path(0, qv);
    if (L2S2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < k) {                                                   // wave-uniform; static register indices
          path(1 + i, o1[i]);
          mean[0] += o1[i][0]; mean[1] += o1[i][1];
        }
    } else {
      for (int p = 1; p <= k; ++p) {
        f32x4 v[2];
        path(p, v);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          mean[e] += v[e];
          if (inside[e]) *reinterpret_cast<f32x4*>(obs1 + (((long)f * k + (p - 1)) * hw2 + otex[e]) * 16 + 4 * kk) = v[e];
        }
      }
    }
    mean[0] *= inv_k; mean[1] *= inv_k;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (inside[e]) {
        float* o = fm1 + ((long)f * hw2 + otex[e]) * 32 + 4 * kk;
        *reinterpret_cast<f32x4*>(o) = qv[e];
        *reinterpret_cast<f32x4*>(o + 16) = mean[e];
      }

    if (L2S2) {
      // ---- stage 3: level 2's stride-2 convs.  The level-1 tile goes back into LDS as [slab][channel quad][x parity]
      // [row 8][x/2 8][4] (slab 0 = q1, 1 = mean o1, 2 + i = o1 of observation i), planar per channel quad and split by
      // x parity so that the stride-2 reads of 16 lanes are 16 different 16-byte slots; rows whose pair index (row/2)
      // is odd, xor the x parity, sit in the other half of the 16 slots (the ^ 8), which keeps both the row-major
      // writes (both parities of one row) and the reads (rows 2Y + a and 2Y + 2 + a) free of bank conflicts.  Lane
      // group kk is tap (a, b) again, wave = (column tile of the 32 level-2 texels, row tile of the 32 outputs).
      __syncthreads();                                                 // every wave is done reading the stage-1 tiles
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = wave + 4 * e, par = j & 1;
        float* dst = lds + (kk * 128 + par * 64 + ((r * 8 + (j >> 1)) ^ ((((r >> 1) & 1) ^ par) * 8))) * 4;
        *reinterpret_cast<f32x4*>(dst) = qv[e];
        *reinterpret_cast<f32x4*>(dst + 2048) = mean[e];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < k) *reinterpret_cast<f32x4*>(dst + (2 + i) * 2048) = o1[i][e];
      }
      __syncthreads();
      const int ct = wave & 1, rt = wave >> 1;
      const int t2 = ct * 16 + j;                                      // level-2 texel of the 4 x 8 tile
      const int Y = t2 >> 3, X = t2 & 7;
      const float* src = lds + ((kk & 1) * 64 + (((2 * Y + (kk >> 1)) * 8 + X) ^ (((Y & 1) ^ (kk & 1)) * 8))) * 4;
      const int gy2 = (ty0 >> 1) + Y, gx2 = (tx0 >> 1) + X;
      const int h4 = h2 >> 1, w4 = w2 >> 1;
      const bool in2 = gy2 < h4 && gx2 < w4;
      const long tex2 = (long)gy2 * w4 + gx2;
      const int oc = rt * 16 + 4 * kk;
      {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + (c8 >> 2) * 2048 + 512 * (c8 & 3));
          const f32x4 a = *reinterpret_cast<const f32x4*>(blob3 + OFF3_AQ + ((rt * 8 + c8) * 64 + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], v[e], acc, 0, 0, 0);
        }
        acc = lrelu4(acc + *reinterpret_cast<const f32x4*>(blob3 + OFF3_BQ + oc), alpha);
        if (in2) *reinterpret_cast<f32x4*>(qtmp2 + ((long)f * h4 * w4 + tex2) * 32 + oc) = acc;
      }
      f32x4 ao3[4];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) ao3[c4] = *reinterpret_cast<const f32x4*>(blob3 + OFF3_AO + ((rt * 4 + c4) * 64 + lane) * 4);
      const f32x4 bo3 = *reinterpret_cast<const f32x4*>(blob3 + OFF3_BO + oc);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < k) {
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (2 + i) * 2048 + 512 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ao3[c4][e], v[e], acc, 0, 0, 0);
          }
          acc = lrelu4(acc + bo3, alpha);
          if (in2) *reinterpret_cast<f32x4*>(otmp2 + (((long)f * k + i) * h4 * w4 + tex2) * 32 + oc) = acc;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------
// back kernel.  Half-resolution tile with a 1-texel halo on the TOP/LEFT (the transposed stride-1
// conv reads (y-a, x-b)); stage 1 = Conv2DTranspose k2s2 on the MFMA (K = 8 + 32 input channels of the
// virtual concat [x | fm1], 16 columns = (a,b,o)), its 2x2x4 outputs go to a full-resolution LDS tile;
// stage 2 = Conv2DTranspose k2s1 (4 -> 4), head (4 -> 3) + skip3, per full-resolution texel on the VALU.
// ---------------------------------------------------------------------------------------
constexpr int FH = 2 * TH + 1, FW = 2 * TW + 1;        // 17 x 33 full-resolution texels incl. top/left halo

// OVR (nlt_back_forward_map: the reference's inference mode, engine_infer.py): `fm1` is the interleaved level-1 map of which
// only the 16 QUERY channels are read (per-texel stride ld1), w_s2 is the Keras (2,2,4,24) slice over [x 8 | query 16], and what
// the given half adds to the first conv's pre-activation (+ its bias) arrives as bmap [1,2 h2,2 w2,4], shared by all frames.
template <bool OVR = false>
__global__ __launch_bounds__(256) void back_kernel(
    const float* __restrict__ x, const float* __restrict__ fm1, const float* __restrict__ skip3,
    int h2, int w2, int tiles_y, int tiles_x,
    const float* __restrict__ w_s2, const float* __restrict__ b_s2, const float* __restrict__ w_s1,
    const float* __restrict__ b_s1, const float* __restrict__ w_head, float alpha, float* __restrict__ pred,
    float* __restrict__ usave, float* __restrict__ vsave, int ld1 = 32, const float* __restrict__ bmap = nullptr) {
  constexpr int KC = OVR ? 24 : 40, NCK = OVR ? 2 : 3;                 // channels / 16-channel chunks of the virtual concat
  __shared__ __attribute__((aligned(16))) float tile_lds[FH * FW * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, j = lane & 15;
  int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int tx0 = (tile % tiles_x) * TW; tile /= tiles_x;
  const int ty0 = (tile % tiles_y) * TH;
  const int f = tile / tiles_y;
  const long hw2 = (long)h2 * w2;

  // A operands: column i = (ab = i >> 2, o = i & 3) of the Keras (2,2,Cout=4,Cin=40) kernel = row i of [16][40]
  f32x4 a2[NCK];
#pragma unroll
  for (int c = 0; c < NCK; ++c) {
    const int c0 = 16 * c + 4 * kk;
    a2[c] = c0 < KC ? *reinterpret_cast<const f32x4*>(w_s2 + j * KC + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const f32x4 bs2 = OVR ? (f32x4){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(b_s2);   // (OVR: the bias is part of the map)

  // r04: every HBM request of the workgroup is issued before the first MFMA -- the texels of the wave's (up to) three column
  // tiles AND the skip3 rows stage 2 will add -- instead of one exposed round trip per column tile and one more after the barrier
  // (the kernel ran at 0.50 of the HBM peak with 8 workgroups per CU to cover for that).
  constexpr int NIT = (NT + 3) / 4;
  f32x4 bq[NIT][NCK];
  f32x4 mq[NIT];                                                       // OVR: the map at the lane's output sub-texel (a, b) = kk
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int mt = wave + 4 * it;
    const int t = mt * 16 + j;
    const bool live = mt < NT && t < HT;
    const int hy = live ? t / HW : 0, hx = live ? t % HW : 0;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool inside = live && gy >= 0 && gx >= 0 && gy < h2 && gx < w2;
    const long tex = (long)f * hw2 + (inside ? (long)gy * w2 + gx : 0);
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
      const int c0 = 16 * c + 4 * kk;                                  // channel of the virtual concat [x 8 | fm1 32 (OVR: its query 16)]
      f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (mt < NT) {                                                   // wave-uniform
        if (c0 < 8) b = *reinterpret_cast<const f32x4*>(x + tex * 8 + c0);
        else if (c0 < KC) b = *reinterpret_cast<const f32x4*>(fm1 + tex * (OVR ? ld1 : 32) + (c0 - 8));
      }
      bq[it][c] = b;
    }
    if constexpr (OVR) {
      mq[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (mt < NT)
        mq[it] = *reinterpret_cast<const f32x4*>(bmap + (inside ? ((long)(2 * gy + (kk >> 1)) * (2 * w2) + 2 * gx + (kk & 1)) * 4 : 0));
    }
  }
  const int h = 2 * h2, w = 2 * w2;
  float sk[2][3];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int ox = threadIdx.x & 31, oy = (threadIdx.x >> 5) + 8 * half;
    const int y = 2 * ty0 + oy, xg = 2 * tx0 + ox;
    const long tex = ((long)f * h + (y < h ? y : 0)) * w + (xg < w ? xg : 0);
    sk[half][0] = skip3[tex * 3]; sk[half][1] = skip3[tex * 3 + 1]; sk[half][2] = skip3[tex * 3 + 2];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int mt = wave + 4 * it;
    if (mt >= NT) break;
    const int t = mt * 16 + j;
    const bool live = t < HT;
    const int hy = live ? t / HW : 0, hx = live ? t % HW : 0;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool inside = live && gy >= 0 && gx >= 0 && gy < h2 && gx < w2;
    f32x4 acc = OVR ? mq[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCK; ++c) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[c][s4], bq[it][c][s4], acc, 0, 0, 0);
    }
    acc = lrelu4(acc + bs2, alpha);
    if (!inside) acc = (f32x4){0.f, 0.f, 0.f, 0.f};                    // zero padding above / left of the image
    const int ly = 2 * hy + (kk >> 1) - 1, lx = 2 * hx + (kk & 1) - 1;  // lane kk holds output sub-texel (a,b) = kk
    if (live && ly >= 0 && lx >= 0) *reinterpret_cast<f32x4*>(tile_lds + (ly * FW + lx) * 4) = acc;
    if (usave && inside && hy >= 1 && hx >= 1)                          // training: the block's own texels of the 4-channel map
      *reinterpret_cast<f32x4*>(usave + (((long)f * 2 * h2 + 2 * gy + (kk >> 1)) * 2 * w2 + 2 * gx + (kk & 1)) * 4) = acc;
  }
  __syncthreads();

#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int ox = threadIdx.x & 31, oy = (threadIdx.x >> 5) + 8 * half;
    const int y = 2 * ty0 + oy, xg = 2 * tx0 + ox;
    if (y >= h || xg >= w) continue;
    float d[4] = {b_s1[0], b_s1[1], b_s1[2], b_s1[3]};
#pragma unroll
    for (int t = 0; t < 4; ++t) {                                      // y[o] += x[y-a][x-b][c] * W[a][b][o][c]
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile_lds + ((oy + 1 - (t >> 1)) * FW + ox + 1 - (t & 1)) * 4);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[o] = fmaf(v[c], w_s1[(t * 4 + o) * 4 + c], d[o]);
    }
    const long tex = ((long)f * h + y) * w + xg;
    float p0 = sk[half][0], p1 = sk[half][1], p2 = sk[half][2];
    if (vsave)
      *reinterpret_cast<f32x4*>(vsave + tex * 4) = lrelu4((f32x4){d[0], d[1], d[2], d[3]}, alpha);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float dv = d[c] > 0.f ? d[c] : alpha * d[c];
      p0 = fmaf(dv, w_head[c * 3], p0); p1 = fmaf(dv, w_head[c * 3 + 1], p1); p2 = fmaf(dv, w_head[c * 3 + 2], p2);
    }
    if (y == 0 && xg == 0) { p0 = 0.f; p1 = 0.f; p2 = 0.f; }         // set_left_top_corner(pred, 0)
    pred[tex * 3] = p0; pred[tex * 3 + 1] = p1; pred[tex * 3 + 2] = p2;
  }
}

}  // namespace

extern "C" long nlt_front_packed_floats(void) { return BLOB; }

extern "C" int nlt_front_pack_weights(const float* wq0, const float* bq0, const float* wo0, const float* bo0,
                                      const float* wqa, const float* bqa, const float* wqb, const float* bqb,
                                      const float* woa, const float* boa, const float* wob, const float* bob,
                                      const float* wh, const float* bh, float* packed, void* stream) {
  if (!wq0 || !bq0 || !wo0 || !bo0 || !wqa || !bqa || !wqb || !bqb || !woa || !boa || !wob || !bob || !wh || !bh || !packed)
    return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(packed)) return NLT_ERR_BAD_ARG;
  FrontW w = {wq0, bq0, wo0, bo0, wqa, bqa, wqb, bqb, woa, boa, wob, bob, wh, bh};
  hipLaunchKernelGGL(front_pack_kernel, dim3((BLOB + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

static int front_launch(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                        const float* nn_base, int n, int k, int h, int w, const float* packed,
                        int add_base, float alpha, float* fm1, float* obs1, float* skip3, float* qtmp1, float* otmp1,
                        void* stream) {
  if (!base || !cvis || !lvis || !nn_rgb || !nn_base || !packed || !fm1 || !obs1 || !skip3) return NLT_ERR_BAD_ARG;
  if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  if ((h | w) & 1) return NLT_ERR_UNSUPPORTED;
  if (!nlt_aligned16(packed) || !nlt_aligned16(fm1) || !nlt_aligned16(obs1)) return NLT_ERR_BAD_ARG;
  const size_t lds_bytes = (size_t)(1 + k) * PATH * sizeof(float);
  if (k > 14) return NLT_ERR_UNSUPPORTED;                              // (1 + k) * 10 KB of LDS, 160 KB per CU
  if ((long long)n * k * h * w * 3 >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  const int ty = (h / 2 + TH - 1) / TH, tx = (w / 2 + TW - 1) / TW;
  const long blocks = (long)n * ty * tx;
  if (lds_bytes > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(front_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess) return NLT_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(front_kernel<false>, dim3((unsigned)blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream),
                     base, cvis, lvis, nn_rgb, nn_base, k, h, w, ty, tx, packed, add_base, alpha, fm1, obs1, skip3,
                     nullptr, nullptr, nullptr, qtmp1, otmp1);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_front_forward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                 const float* nn_base, int n, int k, int h, int w, const float* packed,
                                 int add_base, float alpha, float* fm1, float* obs1, float* skip3, void* stream) {
  return front_launch(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, packed, add_base, alpha, fm1, obs1, skip3,
                      nullptr, nullptr, stream);
}

extern "C" int nlt_front_forward_train(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                       const float* nn_base, int n, int k, int h, int w, const float* packed,
                                       int add_base, float alpha, float* fm1, float* obs1, float* skip3,
                                       float* qtmp1, float* otmp1, void* stream) {
  if (!qtmp1 || !otmp1 || !nlt_aligned16(qtmp1) || !nlt_aligned16(otmp1)) return NLT_ERR_BAD_ARG;
  return front_launch(base, cvis, lvis, nn_rgb, nn_base, n, k, h, w, packed, add_base, alpha, fm1, obs1, skip3,
                      qtmp1, otmp1, stream);
}

extern "C" long nlt_front_l2_packed_floats(void) { return BLOB3; }

extern "C" int nlt_front_pack_l2_weights(const float* wq, const float* bq, const float* wo, const float* bo, float* packed,
                                         void* stream) {
  if (!wq || !bq || !wo || !bo || !packed || !nlt_aligned16(packed)) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(front_pack_l2_kernel, dim3((BLOB3 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), wq, bq, wo, bo, packed);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_front2_forward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                  const float* nn_base, int n, int k, int h, int w, const float* packed,
                                  const float* packed_l2, int add_base, float alpha, float* fm1, float* skip3,
                                  float* qtmp2, float* otmp2, void* stream) {
  if (!base || !cvis || !lvis || !nn_rgb || !nn_base || !packed || !packed_l2 || !fm1 || !skip3 || !qtmp2 || !otmp2) return NLT_ERR_BAD_ARG;
  if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  if (((h | w) & 3) || k > 4) return NLT_ERR_UNSUPPORTED;              // level 2 halves the half-resolution grid again
  if (!nlt_aligned16(packed) || !nlt_aligned16(packed_l2) || !nlt_aligned16(fm1) || !nlt_aligned16(qtmp2) || !nlt_aligned16(otmp2))
    return NLT_ERR_BAD_ARG;
  if ((long long)n * k * h * w * 3 >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  size_t lds_floats = (size_t)(1 + k) * PATH;
  if ((size_t)(2 + k) * 2048 > lds_floats) lds_floats = (size_t)(2 + k) * 2048;
  const size_t lds_bytes = lds_floats * sizeof(float);
  const int ty = (h / 2 + TH - 1) / TH, tx = (w / 2 + TW - 1) / TW;
  const long blocks = (long)n * ty * tx;
  hipLaunchKernelGGL(front_kernel<true>, dim3((unsigned)blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream),
                     base, cvis, lvis, nn_rgb, nn_base, k, h, w, ty, tx, packed, add_base, alpha, fm1, nullptr, skip3,
                     packed_l2, qtmp2, otmp2, nullptr, nullptr);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

static int back_launch(const float* x, const float* fm1, const float* skip3, int n, int h2, int w2,
                       const float* w_s2, const float* b_s2, const float* w_s1, const float* b_s1,
                       const float* w_head, float alpha, float* pred, float* u, float* v, void* stream) {
  if (!x || !fm1 || !skip3 || !w_s2 || !b_s2 || !w_s1 || !b_s1 || !w_head || !pred) return NLT_ERR_BAD_ARG;
  if (n <= 0 || h2 <= 0 || w2 <= 0) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(x) || !nlt_aligned16(fm1) || !nlt_aligned16(w_s2) || !nlt_aligned16(b_s2)) return NLT_ERR_BAD_ARG;
  if ((long long)n * h2 * w2 * 4 * 8 >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  const int ty = (h2 + TH - 1) / TH, tx = (w2 + TW - 1) / TW;
  const long blocks = (long)n * ty * tx;
  hipLaunchKernelGGL(back_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, fm1, skip3, h2, w2, ty, tx, w_s2, b_s2, w_s1, b_s1, w_head, alpha, pred, u, v, 32, nullptr);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_back_forward_map(const float* x, const float* q1, int ldq, const float* skip3, int n, int h2, int w2,
                                    const float* w_s2q, const float* w_s1, const float* b_s1, const float* w_head, float alpha,
                                    const float* bias_map, float* pred, void* stream) {
  if (!x || !q1 || !skip3 || !w_s2q || !w_s1 || !b_s1 || !w_head || !bias_map || !pred) return NLT_ERR_BAD_ARG;
  if (n <= 0 || h2 <= 0 || w2 <= 0 || ldq < 16 || (ldq & 3)) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(x) || !nlt_aligned16(q1) || !nlt_aligned16(w_s2q) || !nlt_aligned16(bias_map)) return NLT_ERR_BAD_ARG;
  if ((long long)n * h2 * w2 * 4 * 8 >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  const int ty = (h2 + TH - 1) / TH, tx = (w2 + TW - 1) / TW;
  const long blocks = (long)n * ty * tx;
  hipLaunchKernelGGL(back_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     x, q1, skip3, h2, w2, ty, tx, w_s2q, b_s1, w_s1, b_s1, w_head, alpha, pred, nullptr, nullptr, ldq, bias_map);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_back_forward(const float* x, const float* fm1, const float* skip3, int n, int h2, int w2,
                                const float* w_s2, const float* b_s2, const float* w_s1, const float* b_s1,
                                const float* w_head, float alpha, float* pred, void* stream) {
  return back_launch(x, fm1, skip3, n, h2, w2, w_s2, b_s2, w_s1, b_s1, w_head, alpha, pred, nullptr, nullptr, stream);
}

extern "C" int nlt_back_forward_train(const float* x, const float* fm1, const float* skip3, int n, int h2, int w2,
                                      const float* w_s2, const float* b_s2, const float* w_s1, const float* b_s1,
                                      const float* w_head, float alpha, float* pred, float* u, float* v, void* stream) {
  if (!u || !v || !nlt_aligned16(u) || !nlt_aligned16(v)) return NLT_ERR_BAD_ARG;
  return back_launch(x, fm1, skip3, n, h2, w2, w_s2, b_s2, w_s1, b_s1, w_head, alpha, pred, u, v, stream);
}
