// Training counterparts of the fused ends (fused.hip): the backward pass of layers 0-1 without any full-resolution
// 16/32-channel tensor.
//
// L0 is a 1x1 conv with NO activation (networks/convnet.py:42, elements.py:26-31), so the features it feeds to level 1's
// stride-2 convs and to the head's skip connection are linear in the raw texel channels
//     r[t] = [base(3) cvis lvis | mean_k (nn_rgb - nn_base)(3)]        fm0[t] = r[t] . A0 + c0
// (A0 = blockdiag(W0q 5x16, W0o 3x16), c0 = [b0q b0o]).  With dy1q / dy1o the gradients w.r.t. the PRE-activations of
// level 1's stride-2 convs (what the unfused backward already has) and dpred the gradient of the output, every weight
// gradient of L0, of the two stride-2 convs and of the head's 32 skip rows is a small matrix product with these
// texel sums:
//     GQ[a,b,c,o] = sum_{i,j} r[2i+a,2j+b,c] dy1q[i,j,o]      (4 x 8 x 16)      SQ[o] = sum dy1q
//     GO[a,b,c,o] = sum_k sum_{i,j} y_k[2i+a,2j+b,c] dy1o_k[i,j,o]  (4 x 3 x 16) SO[o] = sum_k sum dy1o_k
//     H[c,o]      = sum_t r[t,c] dpred[t,o]                   (8 x 3)           P[o]  = sum_t dpred[t,o]
// e.g. dW1q[a,b,m,o] = sum_c A0[c,m] GQ[a,b,c,o] + c0[m] SQ[o],  dW0q[c,m] = sum_{a,b,o} GQ[a,b,c,o] W1q[a,b,m,o] + ...
// front_bwd_kernel forms the sums on the matrix cores (texels are the K dimension: 4 half-resolution texels per
// v_mfma_f32_16x16x4_f32), two deterministic passes reduce them over the grid, and front_bwd_epilogue_kernel applies
// the products.  Replaces, of the unfused backward plan (engine.py): bwd.L1.{q,o}.s2.{wgrad,dgrad}, the query half's
// accumulate into dfm0, bwd.L0.stem and the skip rows of bwd.head; reads 5 + 6k + 3 floats per texel instead of
// ~100 (the 32-channel full-resolution gradient alone is written and read once each in the unfused plan).
#include "nlt_common.h"

namespace {

// totals layout (floats)
constexpr int T_GQ = 0;        // [a 2][row b*8+c 16][o 16]
constexpr int T_R = 512;       // [a 2][row b*8+c 16][col tap'*3+o 16 (12 used)]   R[(a,b,c),(a',b',o)]; H = its tap = tap' part
constexpr int T_GO = 1024;     // [row tap*3+c 16 (12 used)][o 16]
constexpr int T_SQ = 1280, T_SO = 1296, T_P = 1312;   // [16] each (P: col tap'*3+o)
constexpr int T_TOT = 1328;

template <bool PIPE>
__global__ __launch_bounds__(256) void front_bwd_kernel(
    const float* __restrict__ base, const float* __restrict__ cvis, const float* __restrict__ lvis,
    const float* __restrict__ nn_rgb, const float* __restrict__ nn_base, const float* __restrict__ dy1q,
    const float* __restrict__ dy1o, const float* __restrict__ dpred, int k, int h, int w, long groups,
    float* __restrict__ ws) {
  __shared__ float part[4][T_TOT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int h2 = h >> 1, w2 = w >> 1, gpr = w2 >> 2;
  const long hw = (long)h * w, hw2 = (long)h2 * w2;
  const int c = i & 7, b = i >> 3;                                     // query A rows: (b, c) of M tile a
  const int otap = i / 3, oc = i - 3 * otap;                           // obs A rows / R columns: (tap, channel) for i < 12
  const float inv_k = 1.f / (float)k;
  f32x4 gq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 rr[2] = {gq[0], gq[0]};
  f32x4 go = gq[0];
  float sq = 0.f, so = 0.f, sp = 0.f;
  // One group = 4 half-resolution texels (the K index of the MFMAs).  The operands of the NEXT group are requested before the
  // current group's MFMAs: with every load of an iteration consumed in that iteration the waves spent 74 % of their cycles
  // waiting for memory (PMC: SQ_WAIT_ANY; 9 four-byte gathers per lane per group at k = 1).
  struct Ops { float bq, br, aq[2], ao0, bo0; };
  auto fetch = [&](long g, Ops& q) {
    const int x0 = (int)(g % gpr) * 4;
    const long row = g / gpr;
    const int y = (int)(row % h2), f = (int)(row / h2);
    const int xh = x0 + kk;                                            // this lane's half-resolution texel (K index kk)
    const long tq = (long)f * hw2 + (long)y * w2 + xh;
    q.bq = dy1q[tq * 16 + i];
    q.br = 0.f;
    if (i < 12) {
      const int fy = 2 * y + (otap >> 1), fx = 2 * xh + (otap & 1);
      q.br = (fy | fx) ? dpred[((long)f * hw + (long)fy * w + fx) * 3 + oc] : 0.f;   // texel (0,0): set_left_top_corner
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {                                      // query A operands: raw channel c at tap (a, b)
      const long pix = (long)(2 * y + a) * w + 2 * xh + b;
      const long tex = (long)f * hw + pix;
      float v;
      if (c < 3) v = base[tex * 3 + c];
      else if (c == 3) v = cvis[tex];
      else if (c == 4) v = lvis[tex];
      else {
        v = 0.f;
        for (int io = 0; io < k; ++io) {
          const long ot = (((long)f * k + io) * hw + pix) * 3 + (c - 5);
          v += nn_rgb[ot] - nn_base[ot];
        }
        v *= inv_k;
      }
      q.aq[a] = v;
    }
    const long opix = (long)(2 * y + (otap >> 1)) * w + 2 * xh + (otap & 1);
    const long fo = (long)f * k;
    q.ao0 = 0.f;
    if (i < 12) {
      const long ot = (fo * hw + opix) * 3 + oc;
      q.ao0 = nn_rgb[ot] - nn_base[ot];
    }
    q.bo0 = dy1o[(fo * hw2 + (long)y * w2 + xh) * 16 + i];
  };
  auto consume = [&](long g, const Ops& q) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      gq[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.aq[a], q.bq, gq[a], 0, 0, 0);
      rr[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.aq[a], q.br, rr[a], 0, 0, 0);
    }
    sq += q.bq; sp += q.br;
    go = __builtin_amdgcn_mfma_f32_16x16x4f32(q.ao0, q.bo0, go, 0, 0, 0);
    so += q.bo0;
    if (k > 1) {                                                       // further observations of this group
      const int x0 = (int)(g % gpr) * 4;
      const long row = g / gpr;
      const int y = (int)(row % h2), f = (int)(row / h2);
      const int xh = x0 + kk;
      const long opix = (long)(2 * y + (otap >> 1)) * w + 2 * xh + (otap & 1);
      for (int io = 1; io < k; ++io) {
        const long fo = (long)f * k + io;
        float ao = 0.f;
        if (i < 12) {
          const long ot = (fo * hw + opix) * 3 + oc;
          ao = nn_rgb[ot] - nn_base[ot];
        }
        const float bo = dy1o[(fo * hw2 + (long)y * w2 + xh) * 16 + i];
        go = __builtin_amdgcn_mfma_f32_16x16x4f32(ao, bo, go, 0, 0, 0);
        so += bo;
      }
    }
  };
  {
    const long stride = (long)gridDim.x * 4;
    long g = (long)blockIdx.x * 4 + wave;
    Ops cur, nxt;
    if (PIPE) {
      if (g < groups) fetch(g, cur);
      while (g < groups) {
        const long gn = g + stride;
        if (gn < groups) fetch(gn, nxt);
        consume(g, cur);
        cur = nxt;
        g = gn;
      }
    } else {
      for (; g < groups; g += stride) { fetch(g, cur); consume(g, cur); }
    }
  }
  // lane (kk, i) holds rows 4kk..4kk+3, column i of every accumulator
  float* p = part[wave];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      p[T_GQ + (a * 16 + 4 * kk + r) * 16 + i] = gq[a][r];
      p[T_R + (a * 16 + 4 * kk + r) * 16 + i] = rr[a][r];
    }
    p[T_GO + (4 * kk + r) * 16 + i] = go[r];
  }
  sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
  so += __shfl_xor(so, 16); so += __shfl_xor(so, 32);
  sp += __shfl_xor(sp, 16); sp += __shfl_xor(sp, 32);
  if (kk == 0) { p[T_SQ + i] = sq; p[T_SO + i] = so; p[T_P + i] = sp; }
  __syncthreads();
  for (int idx = threadIdx.x; idx < T_TOT; idx += 256)
    ws[(long)blockIdx.x * T_TOT + idx] = (part[0][idx] + part[1][idx]) + (part[2][idx] + part[3][idx]);
}

// ---- second generation: operands staged through wave-private LDS ------------------------------------------------------
// The kernel above is bound by load ISSUE: 15 four-byte gather instructions per group (each of the five raw arrays behind its
// own lane branch) for 5 MFMAs -- 3.9 M vector-memory instructions per launch, the texture addresser busy half the time
// (PMC, r02_c_pmc_train.json).  Here a wave takes a run of 16 half-resolution texels (4 groups) per iteration and fetches
// everything they need -- two full-resolution rows of 32 texels of base / cvis / lvis / dpred / nn_rgb[k] / nn_base[k], the 16
// texels of dy1q and dy1o[k] -- with 3 + 3 k SIXTEEN-byte loads per lane (every lane of an instruction active, each array
// a row-contiguous run), parks them in its own LDS rows, and forms the MFMA operands with 4-byte LDS reads.  The loads of the
// next run are in flight while the current one is multiplied.  No workgroup barrier in the loop.  Needs w % 32 == 0,
// 16-byte aligned arrays and k <= 4 (LDS and registers grow with k); the kernel above stays for everything else.
__device__ __forceinline__ void fb_wave_sync() {       // orders this wave's LDS traffic for the compiler; no instruction
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int FB_QB = 0, FB_QC = 192, FB_QL = 256, FB_QD = 320, FB_DYQ = 512, FB_OBS = 768;   // wave-private LDS map (floats)
constexpr int FB_O_RGB = 0, FB_O_BASE = 192, FB_O_DY = 384, FB_O_SIZE = 640;                  // per observation

template <int K>
__global__ __launch_bounds__(256) void front_bwd2_kernel(
    const float* __restrict__ base, const float* __restrict__ cvis, const float* __restrict__ lvis,
    const float* __restrict__ nn_rgb, const float* __restrict__ nn_base, const float* __restrict__ dy1q,
    const float* __restrict__ dy1o, const float* __restrict__ dpred, int h, int w, long runs, float* __restrict__ ws) {
  __shared__ float part[4][T_TOT];
  extern __shared__ __attribute__((aligned(16))) float stage[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* L = stage + wave * (FB_OBS + K * FB_O_SIZE);
  const int i = lane & 15, kk = lane >> 4;
  const int h2 = h >> 1, w2 = w >> 1, rpr = w2 >> 4;                   // runs per half-resolution row
  const long hw = (long)h * w, hw2 = (long)h2 * w2;
  const int c = i & 7, b = i >> 3;                                     // query A rows: (b, c) of M tile a
  const int otap = i / 3, oc = i - 3 * otap;                           // obs A rows / R columns: (tap, channel) for i < 12
  const float inv_k = 1.f / (float)K;

  // staging descriptors (lane constants).  slot 0: base r0 (24 lanes) | base r1 (24) | cvis r0 (8) | cvis r1 (8)
  //                                        slot 1: lvis r0 (8) | lvis r1 (8) | dpred r0 (24) | dpred r1 (24)
  const float* p0; int ch0, r0, j0, l0;
  if (lane < 48) { p0 = base; ch0 = 3; r0 = lane >= 24; j0 = lane - 24 * r0; l0 = FB_QB + r0 * 96 + 4 * j0; }
  else { p0 = cvis; ch0 = 1; r0 = lane >= 56; j0 = lane - 48 - 8 * r0; l0 = FB_QC + r0 * 32 + 4 * j0; }
  const float* p1; int ch1, r1, j1, l1;
  if (lane < 16) { p1 = lvis; ch1 = 1; r1 = lane >= 8; j1 = lane - 8 * r1; l1 = FB_QL + r1 * 32 + 4 * j1; }
  else { p1 = dpred; ch1 = 3; r1 = lane >= 40; j1 = lane - 16 - 24 * r1; l1 = FB_QD + r1 * 96 + 4 * j1; }
  // per observation: slot A: nn_rgb r0 (24) | nn_rgb r1 (24) | nn_base r0 units 0-15;  slot B: nn_base r0 units 16-23 (8) |
  //                  nn_base r1 (24) | 32 idle lanes;  slot C: dy1o, 16 texels x 4 quads
  const bool a_rgb = lane < 48;
  const int ra = a_rgb ? (lane >= 24) : 0, ja = a_rgb ? lane - 24 * ra : lane - 48;
  const int la = (a_rgb ? FB_O_RGB : FB_O_BASE) + ra * 96 + 4 * ja;
  const bool b_on = lane < 32;
  const int rb = lane >= 8, jb = lane < 8 ? 16 + lane : (b_on ? lane - 8 : 0);
  const int lb = FB_O_BASE + rb * 96 + 4 * jb;

  f32x4 gq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  f32x4 rr[2] = {gq[0], gq[0]};
  f32x4 go = gq[0];
  float sq = 0.f, so = 0.f, sp = 0.f;

  struct Regs { f32x4 q0, q1, dq, oa[K], ob[K], od[K]; };
  auto issue = [&](long run, Regs& v) {
    const int xq = (int)(run % rpr);
    const long row = run / rpr;
    const int y = (int)(row % h2), f = (int)(row / h2);
    const long t0 = (long)f * hw + (long)(2 * y) * w + 32 * xq;        // first full-resolution texel of row 2y
    v.q0 = *reinterpret_cast<const f32x4*>(p0 + (t0 + (long)r0 * w) * ch0 + 4 * j0);
    v.q1 = *reinterpret_cast<const f32x4*>(p1 + (t0 + (long)r1 * w) * ch1 + 4 * j1);
    const long tq = (long)f * hw2 + (long)y * w2 + 16 * xq;
    v.dq = *reinterpret_cast<const f32x4*>(dy1q + tq * 16 + 4 * lane);
#pragma unroll
    for (int io = 0; io < K; ++io) {
      const long fo = (long)f * K + io;
      const long o0 = fo * hw + (long)(2 * y) * w + 32 * xq;
      v.oa[io] = *reinterpret_cast<const f32x4*>((a_rgb ? nn_rgb : nn_base) + (o0 + (long)ra * w) * 3 + 4 * ja);
      v.ob[io] = *reinterpret_cast<const f32x4*>(nn_base + (o0 + (long)rb * w) * 3 + 4 * jb);   // (idle lanes re-read unit 0: in bounds)
      v.od[io] = *reinterpret_cast<const f32x4*>(dy1o + (fo * hw2 + (long)y * w2 + 16 * xq) * 16 + 4 * lane);
    }
  };
  auto park = [&](const Regs& v) {
    *reinterpret_cast<f32x4*>(L + l0) = v.q0;
    *reinterpret_cast<f32x4*>(L + l1) = v.q1;
    *reinterpret_cast<f32x4*>(L + FB_DYQ + 4 * lane) = v.dq;
#pragma unroll
    for (int io = 0; io < K; ++io) {
      float* O = L + FB_OBS + io * FB_O_SIZE;
      *reinterpret_cast<f32x4*>(O + la) = v.oa[io];
      if (b_on) *reinterpret_cast<f32x4*>(O + lb) = v.ob[io];
      *reinterpret_cast<f32x4*>(O + FB_O_DY + 4 * lane) = v.od[io];
    }
  };
  auto multiply = [&](long run) {
    const int xq = (int)(run % rpr);
    const int y = (int)((run / rpr) % h2);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int xl = 4 * g + kk;                                       // this lane's half-resolution texel of the run (K index kk)
      const float bq = L[FB_DYQ + xl * 16 + i];
      float br = 0.f;
      if (i < 12) {
        const int fy = 2 * y + (otap >> 1), fx = 2 * (16 * xq + xl) + (otap & 1);
        br = (fy | fx) ? L[FB_QD + (otap >> 1) * 96 + (2 * xl + (otap & 1)) * 3 + oc] : 0.f;   // texel (0,0): set_left_top_corner
      }
      float aq[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int tx = 2 * xl + b;
        float v;
        if (c < 3) v = L[FB_QB + a * 96 + tx * 3 + c];
        else if (c == 3) v = L[FB_QC + a * 32 + tx];
        else if (c == 4) v = L[FB_QL + a * 32 + tx];
        else {
          v = 0.f;
#pragma unroll
          for (int io = 0; io < K; ++io) {
            const float* O = L + FB_OBS + io * FB_O_SIZE;
            v += O[FB_O_RGB + a * 96 + tx * 3 + c - 5] - O[FB_O_BASE + a * 96 + tx * 3 + c - 5];
          }
          v *= inv_k;
        }
        aq[a] = v;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        gq[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[a], bq, gq[a], 0, 0, 0);
        rr[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[a], br, rr[a], 0, 0, 0);
      }
      sq += bq; sp += br;
#pragma unroll
      for (int io = 0; io < K; ++io) {
        const float* O = L + FB_OBS + io * FB_O_SIZE;
        float ao = 0.f;
        if (i < 12) {
          const int oo = (otap >> 1) * 96 + (2 * xl + (otap & 1)) * 3 + oc;
          ao = O[FB_O_RGB + oo] - O[FB_O_BASE + oo];
        }
        const float bo = O[FB_O_DY + xl * 16 + i];
        go = __builtin_amdgcn_mfma_f32_16x16x4f32(ao, bo, go, 0, 0, 0);
        so += bo;
      }
    }
  };
  {
    const long stride = (long)gridDim.x * 4;
    long run = (long)blockIdx.x * 4 + wave;
    Regs v;
    if (run < runs) issue(run, v);
    while (run < runs) {
      park(v);
      fb_wave_sync();
      const long nxt = run + stride;
      if (nxt < runs) issue(nxt, v);
      multiply(run);
      fb_wave_sync();                                                  // the reads are done before the next run is parked
      run = nxt;
    }
  }
  // lane (kk, i) holds rows 4kk..4kk+3, column i of every accumulator
  float* p = part[wave];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      p[T_GQ + (a * 16 + 4 * kk + r) * 16 + i] = gq[a][r];
      p[T_R + (a * 16 + 4 * kk + r) * 16 + i] = rr[a][r];
    }
    p[T_GO + (4 * kk + r) * 16 + i] = go[r];
  }
  sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
  so += __shfl_xor(so, 16); so += __shfl_xor(so, 32);
  sp += __shfl_xor(sp, 16); sp += __shfl_xor(sp, 32);
  if (kk == 0) { p[T_SQ + i] = sq; p[T_SO + i] = so; p[T_P + i] = sp; }
  __syncthreads();
  for (int idx = threadIdx.x; idx < T_TOT; idx += 256)
    ws[(long)blockIdx.x * T_TOT + idx] = (part[0][idx] + part[1][idx]) + (part[2][idx] + part[3][idx]);
}

// totals[idx] = sum over the nblocks partial rows: 16 entries per workgroup, rows dealt to 16 thread groups x 4 running sums
// (a serial walk over ~1000 rows is pure load latency), combined in a fixed order
__global__ __launch_bounds__(256) void front_bwd_reduce_kernel(const float* __restrict__ ws, int nblocks, float* __restrict__ totals) {
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + e;                                 // T_TOT is a multiple of 16
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int bl = g;
  for (; bl + 48 < nblocks; bl += 64) {
    s0 += ws[(long)bl * T_TOT + idx];
    s1 += ws[(long)(bl + 16) * T_TOT + idx];
    s2 += ws[(long)(bl + 32) * T_TOT + idx];
    s3 += ws[(long)(bl + 48) * T_TOT + idx];
  }
  for (; bl < nblocks; bl += 16) s0 += ws[(long)bl * T_TOT + idx];
  part[g][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g) return;
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += part[q][e];
  totals[idx] = s;
}

struct FrontBwdW {
  const float *wq0, *bq0, *wo0, *bo0;      // L0: (1,1,5,16), (1,1,3,16)
  const float *wqa, *woa;                  // stride-2 convs of level 1: (2,2,32,16), (2,2,16,16)
  const float *wh;                         // head (1,1,36,3)
  float *dwq0, *dbq0, *dwo0, *dbo0, *dwqa, *dbqa, *dwoa, *dboa, *dwh;   // accumulated (+=)
};

// output index space of the epilogue
constexpr int E_BQA = 2048, E_WOA = 2064, E_BOA = 3088, E_WQ0 = 3104, E_BQ0 = 3184, E_WO0 = 3200, E_BO0 = 3248,
              E_WH = 3264, E_TOT = 3360;

__global__ __launch_bounds__(256) void front_bwd_epilogue_kernel(const float* __restrict__ totals, FrontBwdW w) {
  __shared__ float t[T_TOT];
  __shared__ float hm[8][3], pp[3];
  for (int idx = threadIdx.x; idx < T_TOT; idx += 256) t[idx] = totals[idx];
  __syncthreads();
  if (threadIdx.x < 24) {                                              // H[c][o] = sum_tap R[(tap, c), (tap, o)]
    const int c = threadIdx.x / 3, o = threadIdx.x % 3;
    float v = 0.f;
    for (int tap = 0; tap < 4; ++tap) v += t[T_R + ((tap >> 1) * 16 + (tap & 1) * 8 + c) * 16 + tap * 3 + o];
    hm[c][o] = v;
  } else if (threadIdx.x < 27) {
    const int o = threadIdx.x - 24;
    pp[o] = (t[T_P + o] + t[T_P + 3 + o]) + (t[T_P + 6 + o] + t[T_P + 9 + o]);
  }
  __syncthreads();
  auto GQ = [&](int tap, int c, int o) { return t[T_GQ + ((tap >> 1) * 16 + (tap & 1) * 8 + c) * 16 + o]; };
  auto GO = [&](int tap, int c, int o) { return t[T_GO + (tap * 3 + c) * 16 + o]; };
  for (int e = threadIdx.x; e < E_TOT; e += 256) {
    float v = 0.f;
    if (e < E_BQA) {                                                   // dW1q[tap][m 32][o 16]
      const int o = e & 15, m = (e >> 4) & 31, tap = e >> 9;
      if (m < 16) { for (int c = 0; c < 5; ++c) v = fmaf(w.wq0[c * 16 + m], GQ(tap, c, o), v); v = fmaf(w.bq0[m], t[T_SQ + o], v); }
      else { for (int c = 0; c < 3; ++c) v = fmaf(w.wo0[c * 16 + m - 16], GQ(tap, 5 + c, o), v); v = fmaf(w.bo0[m - 16], t[T_SQ + o], v); }
      w.dwqa[e] += v;
    } else if (e < E_WOA) {
      w.dbqa[e - E_BQA] += t[T_SQ + e - E_BQA];
    } else if (e < E_BOA) {                                            // dW1o[tap][m 16][o 16]
      const int r = e - E_WOA, o = r & 15, m = (r >> 4) & 15, tap = r >> 8;
      for (int c = 0; c < 3; ++c) v = fmaf(w.wo0[c * 16 + m], GO(tap, c, o), v);
      v = fmaf(w.bo0[m], t[T_SO + o], v);
      w.dwoa[r] += v;
    } else if (e < E_WQ0) {
      w.dboa[e - E_BOA] += t[T_SO + e - E_BOA];
    } else if (e < E_BQ0) {                                            // dW0q[c 5][m 16]
      const int r = e - E_WQ0, m = r & 15, c = r >> 4;
      for (int tap = 0; tap < 4; ++tap)
        for (int o = 0; o < 16; ++o) v = fmaf(GQ(tap, c, o), w.wqa[(tap * 32 + m) * 16 + o], v);
      for (int o = 0; o < 3; ++o) v = fmaf(hm[c][o], w.wh[(4 + m) * 3 + o], v);
      w.dwq0[r] += v;
    } else if (e < E_WO0) {                                            // db0q[m]
      const int m = e - E_BQ0;
      for (int tap = 0; tap < 4; ++tap)
        for (int o = 0; o < 16; ++o) v = fmaf(w.wqa[(tap * 32 + m) * 16 + o], t[T_SQ + o], v);
      for (int o = 0; o < 3; ++o) v = fmaf(w.wh[(4 + m) * 3 + o], pp[o], v);
      w.dbq0[m] += v;
    } else if (e < E_BO0) {                                            // dW0o[c 3][m 16]
      const int r = e - E_WO0, m = r & 15, c = r >> 4;
      for (int tap = 0; tap < 4; ++tap)
        for (int o = 0; o < 16; ++o) {
          v = fmaf(GQ(tap, 5 + c, o), w.wqa[(tap * 32 + 16 + m) * 16 + o], v);
          v = fmaf(GO(tap, c, o), w.woa[(tap * 16 + m) * 16 + o], v);
        }
      for (int o = 0; o < 3; ++o) v = fmaf(hm[5 + c][o], w.wh[(20 + m) * 3 + o], v);
      w.dwo0[r] += v;
    } else if (e < E_WH) {                                             // db0o[m]
      const int m = e - E_BO0;
      for (int tap = 0; tap < 4; ++tap)
        for (int o = 0; o < 16; ++o) {
          v = fmaf(w.wqa[(tap * 32 + 16 + m) * 16 + o], t[T_SQ + o], v);
          v = fmaf(w.woa[(tap * 16 + m) * 16 + o], t[T_SO + o], v);
        }
      for (int o = 0; o < 3; ++o) v = fmaf(w.wh[(20 + m) * 3 + o], pp[o], v);
      w.dbo0[m] += v;
    } else {                                                           // dWh[4 + m][o], m < 32
      const int r = e - E_WH, o = r % 3, m = r / 3;
      if (m < 16) { for (int c = 0; c < 5; ++c) v = fmaf(w.wq0[c * 16 + m], hm[c][o], v); v = fmaf(w.bq0[m], pp[o], v); }
      else { for (int c = 0; c < 3; ++c) v = fmaf(w.wo0[c * 16 + m - 16], hm[5 + c][o], v); v = fmaf(w.bo0[m - 16], pp[o], v); }
      w.dwh[12 + r] += v;
    }
  }
}

int front_bwd_blocks(long groups) {
  long blocks = (groups + 3) / 4;
  if (blocks > 1536) blocks = 1536;                                    // 6 waves / SIMD resident
  return (int)blocks;
}

}  // namespace

extern "C" long nlt_front_backward_workspace_floats(int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 7)) return -1;
  const long groups = (long)n * (h / 2) * (w / 8);
  return ((long)front_bwd_blocks(groups) + 1) * T_TOT;
}

extern "C" int nlt_front_backward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                                  const float* nn_base, int n, int k, int h, int w, const float* dy1q, const float* dy1o,
                                  const float* dpred, const float* wq0, const float* bq0, const float* wo0,
                                  const float* bo0, const float* wqa, const float* woa, const float* wh,
                                  float* dwq0, float* dbq0, float* dwo0, float* dbo0, float* dwqa, float* dbqa,
                                  float* dwoa, float* dboa, float* dwh, float* workspace, void* stream) {
  if (!base || !cvis || !lvis || !nn_rgb || !nn_base || !dy1q || !dy1o || !dpred || !wq0 || !bq0 || !wo0 || !bo0 || !wqa ||
      !woa || !wh || !dwq0 || !dbq0 || !dwo0 || !dbo0 || !dwqa || !dbqa || !dwoa || !dboa || !dwh || !workspace)
    return NLT_ERR_BAD_ARG;
  if (n <= 0 || k <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  if ((h & 1) || (w & 7)) return NLT_ERR_UNSUPPORTED;                  // groups of 4 half-resolution texels along x
  if ((long long)n * k * h * w * 3 >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long groups = (long)n * (h / 2) * (w / 8);
  const int blocks = front_bwd_blocks(groups);
  float* totals = workspace + (long)blocks * T_TOT;
  static const bool pipe = [] { const char* e = getenv("NLT_FRONT_BWD_PIPE"); return !(e && e[0] == '0'); }();
  static const bool staged = [] { const char* e = getenv("NLT_FRONT_BWD_STAGED"); return !(e && e[0] == '0'); }();
  const bool aligned = nlt_aligned16(base) && nlt_aligned16(cvis) && nlt_aligned16(lvis) && nlt_aligned16(nn_rgb) &&
                       nlt_aligned16(nn_base) && nlt_aligned16(dy1q) && nlt_aligned16(dy1o) && nlt_aligned16(dpred);
  if (staged && aligned && (w & 31) == 0 && k <= 4) {
    const long runs = (long)n * (h / 2) * (w / 32);
    const size_t lds = (size_t)4 * (FB_OBS + k * FB_O_SIZE) * sizeof(float);
#define NLT_FB2(K_) hipLaunchKernelGGL(front_bwd2_kernel<K_>, dim3(blocks), dim3(256), lds, s, base, cvis, lvis, nn_rgb, nn_base, \
                                       dy1q, dy1o, dpred, h, w, runs, workspace)
    if (k == 1) NLT_FB2(1); else if (k == 2) NLT_FB2(2); else if (k == 3) NLT_FB2(3); else NLT_FB2(4);
#undef NLT_FB2
  } else if (pipe)
    hipLaunchKernelGGL(front_bwd_kernel<true>, dim3(blocks), dim3(256), 0, s, base, cvis, lvis, nn_rgb, nn_base, dy1q, dy1o, dpred,
                       k, h, w, groups, workspace);
  else
    hipLaunchKernelGGL(front_bwd_kernel<false>, dim3(blocks), dim3(256), 0, s, base, cvis, lvis, nn_rgb, nn_base, dy1q, dy1o, dpred,
                       k, h, w, groups, workspace);
  NLT_CHECK_LAUNCH();
  static_assert(T_TOT % 16 == 0, "reduce kernel: 16 entries per workgroup");
  hipLaunchKernelGGL(front_bwd_reduce_kernel, dim3(T_TOT / 16), dim3(256), 0, s, workspace, blocks, totals);
  NLT_CHECK_LAUNCH();
  FrontBwdW fw = {wq0, bq0, wo0, bo0, wqa, woa, wh, dwq0, dbq0, dwo0, dbo0, dwqa, dbqa, dwoa, dboa, dwh};
  hipLaunchKernelGGL(front_bwd_epilogue_kernel, dim3(1), dim3(256), 0, s, totals, fw);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
