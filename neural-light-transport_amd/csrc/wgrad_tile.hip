// Weight / bias gradients, second generation:  dW[k][n] = sum_rows X[row][k] * dP[row][n].
//
// The reduction runs over texel rows, 4 per v_mfma_f32_16x16x4_f32 step, so operand traffic decides the speed.
// Here both operands are fetched with ONE 16-byte NHWC load per lane per step and feed 16 MFMAs:
//   lane (i = l & 15, kk = l >> 4) loads the channel QUAD i of row kk of X  -> A_e[i][kk] = quad[e], e = 0..3
//   lane (j = l & 15, kk = l >> 4) loads the output  QUAD j of row kk of dP -> B_f[kk][j] = quad[f], f = 0..3
//   D_{e,f} = A_e * B_f accumulates dW[channel 4*quad_i + e][output 4*quad_j + f]
// i.e. a wave owns a 64 x 64 block of dW (16 k-quads x 16 n-quads; k-quads run over tap x virtual-concat channel
// quads, n-quads over output channels or, for Conv2DTranspose k2s2, over (a,b,o)) for one slice of the rows.
// Row slices write their partial blocks to a workspace with 16-byte stores; a second launch adds the slices in
// order and accumulates into the Keras-layout gradient -- deterministic, and no atomics (the first-generation
// kernel's thousands of waves adding into the 64 addresses of a 4 -> 4 layer were its bottleneck).
#include "nlt_common.h"
#include <cstdlib>

namespace {

template <int MODE>
__device__ __forceinline__ long keras_widx(int t, int c, int ncol, int cin, int cout) {
  if (MODE == NLT_CONV1X1 || MODE == NLT_CONV_K2S2 || MODE == NLT_CONV_K2S1) return ((long)t * cin + c) * cout + ncol;
  if (MODE == NLT_DECONV_K2S1) return ((long)t * cout + ncol) * cin + c;
  return (long)ncol * cin + c;   // DECONV_K2S2: ncol = (a*2+b)*cout + o
}

struct WT {
  ConvP c;            // geometry + X sources
  const float* dp; int ldp;
  float* dw; float* db;
  float* ws; float* wsb;
  const float* zeros;  // 16 zero bytes in device memory (the walk kernel's operand source past the end of a run)
  int kq, nq;         // k-quads (taps * (c0 + c1) / 4), n-quads (N / 4)
  int kblocks, nblocks, msplits, rows_per_split;
  int xcd_order;      // slice ms pinned to XCD ms % 8, its tiles consecutive there (wgrad_block)
};

// Workgroup -> (row slice, k-block, n-block).  Every tile of dW re-reads the slice's X rows (once per n-block and tap) and dP
// rows (once per k-block): 134 MB of L2 misses for 17 MB of operands at level 4 when those workgroups are dealt round-robin
// over the 8 XCDs (measured: FETCH_SIZE, TCC_MISS; that traffic, not the matrix pipe, was the kernel's time).  Workgroup ids go
// to XCDs modulo 8, so slice ms is pinned to XCD ms % 8 and its tiles are consecutive there: the slice is fetched into ONE L2 once.
__device__ __forceinline__ bool wgrad_block(const WT& w, int& ms, int& kb, int& nb) {
  const int tiles = w.kblocks * w.nblocks;
  if (!w.xcd_order) {                                                // few slices (deep levels): slices fastest, every XCD busy
    int blk = blockIdx.x;
    ms = blk % w.msplits; blk /= w.msplits;
    nb = blk % w.nblocks;
    kb = blk / w.nblocks;
    return true;
  }
  const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
  const int tile = r % tiles;
  ms = (r / tiles) * 8 + xcd;
  nb = tile % w.nblocks;
  kb = tile / w.nblocks;
  return ms < w.msplits;
}

template <int MODE>
__global__ __launch_bounds__(256) void wgrad_tile_kernel(WT w) {
  __shared__ __attribute__((aligned(16))) f32x4 part[3][17][64];      // waves 1..3 -> wave 0 (16 blocks + bias sums)
  const ConvP& p = w.c;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ms, kb, nb;                                                    // one workgroup = one (row slice, kb, nb)
  if (!wgrad_block(w, ms, kb, nb)) return;
  const int i = lane & 15, kk = lane >> 4;
  const int q0 = p.c0 >> 2, qpt = (p.c0 + p.c1) >> 2;

  // A role: this lane's k-quad = (tap, channel quad of the virtual concat)
  const int kq = kb * 16 + i;
  const bool a_ok = kq < w.kq;
  const int tap = a_ok ? kq / qpt : 0;
  const int cq = a_ok ? kq - tap * qpt : 0;
  const bool from1 = cq >= q0;
  const float* asrc = from1 ? p.src1 + 4 * (cq - q0) : p.src0 + 4 * cq;
  const int ald = from1 ? p.ld1 : p.ld0;
  // B role: this lane's n-quad
  const int nq = nb * 16 + i;
  const bool b_ok = nq < w.nq;
  const int ncol = b_ok ? 4 * nq : 0;
  const int ab = MODE == NLT_DECONV_K2S2 ? ncol / p.cout : 0;
  const int oc = MODE == NLT_DECONV_K2S2 ? ncol - ab * p.cout : ncol;

  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The slice's rows are dealt to the 4 waves in steps of 4 rows (wave wv takes steps wv, wv + 4, ...).  Operand
  // loads run two steps ahead of the MFMAs (three register sets), and the (frame, y, x) of a lane's row is advanced
  // incrementally instead of being re-derived by integer division every step.
  const int m_begin = ms * w.rows_per_split;
  int m_end = m_begin + w.rows_per_split;
  if (m_end > p.M) m_end = p.M;
  int m = m_begin + 4 * wv + kk;                                   // this lane's row of the NEXT step to be loaded
  int rx, ry, rf;
  {
    const int mc = m < p.M ? m : p.M - 1;
    rx = mc % p.gw; ry = (mc / p.gw) % p.gh; rf = mc / (p.gw * p.gh);
  }
  auto issue = [&](f32x4& av, f32x4& bv) {
    const bool rv = m < m_end;
    const int tex = conv_tap_texel<MODE>(p, rf, ry, rx, tap);
    const size_t tx = (rv && tex >= 0) ? (size_t)tex : 0;
    av = *reinterpret_cast<const f32x4*>(asrc + tx * ald);                      // unconditional, clamped address
    if (!(rv && a_ok && tex >= 0)) av = (f32x4){0.f, 0.f, 0.f, 0.f};
    size_t otex = rv ? ((size_t)rf * p.gh + ry) * p.gw + rx : 0;
    if (MODE == NLT_DECONV_K2S2) otex = rv ? ((size_t)rf * p.oh + 2 * ry + (ab >> 1)) * p.ow + 2 * rx + (ab & 1) : 0;
    bv = *reinterpret_cast<const f32x4*>(w.dp + otex * w.ldp + oc);
    if (!(rv && b_ok)) bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    m += 16; rx += 16;                                               // the wave's next step is 16 rows further
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                    // gw >= 4 (checked by the host): at most 4 wraps, branch-free
      const bool wx = rx >= p.gw;
      rx -= wx ? p.gw : 0;
      ry += wx ? 1 : 0;
      const bool wy = ry >= p.gh;
      ry = wy ? 0 : ry;
      rf += wy ? 1 : 0;
    }
  };
  auto compute = [&](const f32x4& av, const f32x4& bv) {
    bsum += bv;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4)
        acc[e][f4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[f4], acc[e][f4], 0, 0, 0);
  };
  const int first = m_begin + 4 * wv;
  const int nsteps = first < m_end ? (m_end - first + 15) / 16 : 0;
  f32x4 a0, b0, a1, b1, a2, b2;
  issue(a0, b0);
  issue(a1, b1);
  for (int s3 = 0; s3 < nsteps; s3 += 3) {                           // steps past nsteps carry zero operands
    issue(a2, b2); compute(a0, b0);
    issue(a0, b0); compute(a1, b1);
    issue(a1, b1); compute(a2, b2);
  }

  // the 4 waves' partial blocks are added in wave order (fixed -> deterministic) through LDS ...
  if (wv > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) part[wv - 1][e * 4 + f][lane] = acc[e][f];
    part[wv - 1][16][lane] = bsum;
  }
  __syncthreads();
  if (wv > 0) return;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[e][f] += part[o][e * 4 + f][lane];
    bsum += part[o][16][lane];
  }
  // ... and the slice's block goes to the workspace [ms][kb][nb][e][f][lane] (f32x4 = the 4 D rows this lane holds)
  f32x4* dst = reinterpret_cast<f32x4*>(w.ws) + ((((size_t)ms * w.kblocks + kb) * w.nblocks + nb) * 16) * 64 + lane;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) dst[(e * 4 + f) * 64] = acc[e][f];
  if (w.wsb && kb == 0) {                                                       // column sums of dP for the bias
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bsum[f] += __shfl_xor(bsum[f], 16);
      bsum[f] += __shfl_xor(bsum[f], 32);
    }
    if (kk == 0) reinterpret_cast<f32x4*>(w.wsb)[((size_t)ms * w.nblocks + nb) * 16 + i] = bsum;
  }
}

// The same block, with the rows WALKED instead of re-derived (the form used whenever the row grid is a multiple of 4
// texels wide -- every released shape).  The generic kernel above spends ~70 VALU instructions per 16-MFMA step on
// (frame, y, x) bookkeeping, 64-bit multiplies for two addresses and exec-masked loads; with two waves per SIMD that
// address arithmetic, not the matrix pipe, set its speed.  Here a wave takes a contiguous run of its slice, 4 rows per
// step, and because 4 | gw those 4 rows never straddle an image row: the position (x0, y) of a step is WAVE-UNIFORM
// (scalar registers), each lane keeps two pointers advanced by lane-constant increments (a second constant on the
// steps that wrap to the next image row, where the stride-2 families jump), the loads are unconditional, and padding
// validity (k2s1 families only) is two compares against lane constants.
constexpr int WALK_PF = 6;

template <int MODE>
__global__ __launch_bounds__(256) void wgrad_walk_kernel(WT w) {
  __shared__ __attribute__((aligned(16))) f32x4 part[3][17][64];
  const ConvP& p = w.c;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int ms, kb, nb;                                                    // one workgroup = one (row slice, kb, nb)
  if (!wgrad_block(w, ms, kb, nb)) return;
  const int i = lane & 15, kk = lane >> 4;
  const int q0 = p.c0 >> 2, qpt = (p.c0 + p.c1) >> 2;

  const int kq = kb * 16 + i;
  const bool a_ok = kq < w.kq;
  const int tap = a_ok ? kq / qpt : 0;
  const int cq = a_ok ? kq - tap * qpt : 0;
  const bool from1 = cq >= q0;
  const float* asrc = from1 ? p.src1 + 4 * (cq - q0) : p.src0 + 4 * cq;
  const int ald = from1 ? p.ld1 : p.ld0;
  const int ta = tap >> 1, tb = tap & 1;
  const int nq = nb * 16 + i;
  const bool b_ok = nq < w.nq;
  const int ncol = b_ok ? 4 * nq : 0;
  const int ab = MODE == NLT_DECONV_K2S2 ? ncol / p.cout : 0;
  const int oc = MODE == NLT_DECONV_K2S2 ? ncol - ab * p.cout : ncol;

  // the wave's run of rows [m0, m1): a quarter of the slice, a multiple of 4 rows (rows_per_split % 16 == 0, M % 4 == 0)
  const int chunk = w.rows_per_split >> 2;
  const int m0 = __builtin_amdgcn_readfirstlane(ms * w.rows_per_split + wv * chunk);
  int m1 = m0 + chunk;
  if (m1 > p.M) m1 = p.M;
  const int nsteps = m1 > m0 ? (m1 - m0) >> 2 : 0;
  // wave-uniform position of the step's first row: x0 (multiple of 4), y; R = image row index over all frames
  int x0 = m0 % p.gw;
  const int R0 = m0 / p.gw;
  int y = R0 % p.gh;

  // lane pointers at the step-0 row m0 + kk (unclamped tap position; validity is tracked separately)
  long atex, btex;
  if (MODE == NLT_CONV_K2S2) atex = ((long)(2 * R0 + ta)) * p.w + 2 * (x0 + kk) + tb;
  else if (MODE == NLT_CONV_K2S1) atex = (long)m0 + kk + ta * p.w + tb;
  else if (MODE == NLT_DECONV_K2S1) atex = (long)m0 + kk - ta * p.w - tb;
  else atex = (long)m0 + kk;
  if (MODE == NLT_DECONV_K2S2) btex = ((long)(2 * R0 + (ab >> 1))) * p.ow + 2 * (x0 + kk) + (ab & 1);
  else btex = (long)m0 + kk;
  const float* pa = nsteps > 0 ? asrc + atex * ald : asrc;          // an empty run (slice past M) must not form an address past the buffer
  const float* pb = w.dp + btex * w.ldp + oc;
  // per-step pointer increments (floats): plain step / step that wraps to the next image row
  const int a_inc = (MODE == NLT_CONV_K2S2 ? 8 : 4) * ald;
  const int a_inc_wrap = MODE == NLT_CONV_K2S2 ? (8 + p.w) * ald : a_inc;          // p.w = 2 gw
  const int b_inc = (MODE == NLT_DECONV_K2S2 ? 8 : 4) * w.ldp;
  const int b_inc_wrap = MODE == NLT_DECONV_K2S2 ? (8 + p.ow) * w.ldp : b_inc;
  // padding validity of this lane's tap, as bounds on the uniform (x0, y)
  //   k2s1:           x0 + kk + tb < gw  and  y + ta < gh        ->  x0 < xlim, y < ylim
  //   transposed k2s1: x0 + kk - tb >= 0 and  y - ta >= 0        ->  x0 >= xlo (only x0 = 0, kk = 0, tb = 1 fails), y >= ta
  const int xlim = p.gw - kk - tb, ylim = p.gh - ta;
  const int xlo = tb - kk;

  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Steps past the end of the run (the pipeline issues two ahead; a run clipped by M may not fill its last triple) read
  // dP from a zero page, so their products vanish without a select on the data; X is re-read where the walk stopped.
  const float* const zpage = w.zeros;
  int issued = 0;
  auto issue = [&](f32x4& av, f32x4& bv) {
    bool ok = a_ok;
    if (MODE == NLT_CONV_K2S1) ok = ok && x0 < xlim && y < ylim;
    if (MODE == NLT_DECONV_K2S1) ok = ok && x0 >= xlo && y >= ta;
    const float* la = pa;
    if (MODE == NLT_CONV_K2S1 || MODE == NLT_DECONV_K2S1) la = ok ? pa : asrc;       // a padded tap must not be dereferenced
    av = *reinterpret_cast<const f32x4*>(la);
    bv = *reinterpret_cast<const f32x4*>(pb);
    if (MODE == NLT_CONV_K2S1 || MODE == NLT_DECONV_K2S1) av = ok ? av : zero4;
    ++issued;
    const bool adv = issued < nsteps;                                  // everything below is wave-uniform
    const int xn = x0 + 4;
    const bool wrap = xn >= p.gw;
    const float* pan = pa + (wrap ? a_inc_wrap : a_inc);
    const float* pbn = pb + (wrap ? b_inc_wrap : b_inc);
    pa = adv ? pan : pa;
    pb = adv ? pbn : zpage;
    const int yn = y + 1 >= p.gh ? 0 : y + 1;
    y = (adv && wrap) ? yn : y;
    x0 = adv ? (wrap ? 0 : xn) : x0;
  };
  auto compute = [&](const f32x4& av, const f32x4& bv) {
    bsum += bv;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4)
        acc[e][f4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[f4], acc[e][f4], 0, 0, 0);
  };
  if (nsteps == 0) pb = zpage;
  {
    // operand loads run PF - 1 steps ahead of the MFMAs (PF register sets; the host sizes the runs to multiples of PF steps)
    constexpr int PF = WALK_PF;
    f32x4 av[PF], bv[PF];
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) issue(av[j], bv[j]);
    for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        // (the scheduler would otherwise hoist every load of the iteration to its top and drain them together)
        issue(av[(j + PF - 1) % PF], bv[(j + PF - 1) % PF]);
        __builtin_amdgcn_sched_barrier(0);
        compute(av[j], bv[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // lanes whose k-quad / n-quad is padding carry garbage operands (their loads are unconditional): the rows / columns
  // they produce are dropped by the reduce pass (kq >= w.kq || nq >= w.nq), but the bias sums need real zeros
  if (!b_ok) bsum = zero4;

  if (wv > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) part[wv - 1][e * 4 + f][lane] = acc[e][f];
    part[wv - 1][16][lane] = bsum;
  }
  __syncthreads();
  if (wv > 0) return;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[e][f] += part[o][e * 4 + f][lane];
    bsum += part[o][16][lane];
  }
  f32x4* dst = reinterpret_cast<f32x4*>(w.ws) + ((((size_t)ms * w.kblocks + kb) * w.nblocks + nb) * 16) * 64 + lane;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) dst[(e * 4 + f) * 64] = acc[e][f];
  if (w.wsb && kb == 0) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bsum[f] += __shfl_xor(bsum[f], 16);
      bsum[f] += __shfl_xor(bsum[f], 32);
    }
    if (kk == 0) reinterpret_cast<f32x4*>(w.wsb)[((size_t)ms * w.nblocks + nb) * 16 + i] = bsum;
  }
}

// Pass 2: 64 dW elements of the block layout per workgroup; the slices are dealt to 16 thread groups (16-byte loads, 8 in
// flight per thread) and combined in a fixed order.
template <int MODE> __device__ void wgrad_bias_block(const WT& w, int ocq);

template <int MODE>
__device__ __forceinline__ void wgrad_scatter(const WT& w, long idx, float s) {
  const ConvP& p = w.c;
  const int r = idx & 3, ln = (idx >> 2) & 63, ef = (idx >> 8) & 15;
  const long blk = idx >> 12;
  const int nb = blk % w.nblocks, kb = blk / w.nblocks;
  const int kq = kb * 16 + 4 * (ln >> 4) + r;           // D row = k-quad inside the block
  const int nq = nb * 16 + (ln & 15);                   // D col = n-quad
  if (kq >= w.kq || nq >= w.nq) return;
  const int qpt = (p.c0 + p.c1) >> 2;
  const int tap = kq / qpt;
  const int c = 4 * (kq - tap * qpt) + (ef >> 2);
  const int ncol = 4 * nq + (ef & 3);
  w.dw[keras_widx<MODE>(tap, c, ncol, p.c0 + p.c1, p.cout)] += s;
}

// FEW = true (<= 8 slices: the mid-network layers, whose 64 x 64 blocks of dW are many and whose slices are few): one
// thread adds the slices of FOUR consecutive block entries with 16-byte loads, all of them in flight at once -- 1024
// entries per workgroup.  (The wave-per-slice-group form below gives such a launch 16 384 workgroups of one 4-byte
// load per thread: 16 us for 12 MB.)  Fixed order: ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)).
template <int MODE, bool FEW>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WT w) {
  __shared__ __attribute__((aligned(16))) float part[16][64];
  const long per_slice = (long)w.kblocks * w.nblocks * 4096;
  const long dw_blocks = per_slice / (FEW ? 1024 : 64);
  if ((long)blockIdx.x >= dw_blocks) {                                // the trailing cout / 4 workgroups: the bias (no extra launch)
    wgrad_bias_block<MODE>(w, (int)(blockIdx.x - dw_blocks));
    return;
  }
  if (FEW) {
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[8];
#pragma unroll
    for (int ms = 0; ms < 8; ++ms)
      v[ms] = ms < w.msplits ? *reinterpret_cast<const f32x4*>(w.ws + (size_t)ms * per_slice + idx) : z;
    const f32x4 s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
    for (int r = 0; r < 4; ++r) wgrad_scatter<MODE>(w, idx + r, s[r]);
    return;
  }
  // many slices (the shallow levels: few blocks of dW, hundreds of slices): thread (q, g) adds the slices g, g + 16, ... of the
  // entry QUAD q with 16-byte loads, 8 of them in flight (the 4-byte, 4-stream form of round 2 was pure latency: 23 us for 8 MB)
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long idx = (long)blockIdx.x * 64 + 4 * q;
  const f32x4* src = reinterpret_cast<const f32x4*>(w.ws + idx);
  const size_t stride = (size_t)per_slice / 4;                         // f32x4 per slice
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int ms = g;
  for (; ms + 112 < w.msplits; ms += 128) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(ms + 16 * j) * stride];
    s0 += (v[0] + v[1]) + (v[2] + v[3]);
    s1 += (v[4] + v[5]) + (v[6] + v[7]);
  }
  for (; ms < w.msplits; ms += 16) s0 += src[(size_t)ms * stride];
  reinterpret_cast<f32x4*>(&part[g][0])[q] = s0 + s1;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  const int e = threadIdx.x;
  float t[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) t[a] = (part[4 * a][e] + part[4 * a + 1][e]) + (part[4 * a + 2][e] + part[4 * a + 3][e]);
  wgrad_scatter<MODE>(w, (long)blockIdx.x * 64 + e, (t[0] + t[1]) + (t[2] + t[3]));
}

// Bias: one workgroup (of the reduce launch) per quad of output channels; 256 threads share the (slice, ab) terms,
// fixed-order LDS tree.
template <int MODE>
__device__ void wgrad_bias_block(const WT& w, int ocq) {
  __shared__ f32x4 part[256];
  const ConvP& p = w.c;
  const int t = threadIdx.x;
  const int nab = MODE == NLT_DECONV_K2S2 ? 4 : 1;
  const f32x4* wsb4 = reinterpret_cast<const f32x4*>(w.wsb);
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = t; it < nab * w.msplits; it += 256) {
    const int ab = it % nab, ms = it / nab;
    const int nq = ((ab * p.cout) >> 2) + ocq;                       // n-quad of column ab*cout + 4*ocq
    s += wsb4[((size_t)ms * w.nblocks + nq / 16) * 16 + nq % 16];
  }
  part[t] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) part[t] += part[t + off];
    __syncthreads();
  }
  if (t == 0) {
    const f32x4 v = part[0];
    w.db[4 * ocq] += v[0]; w.db[4 * ocq + 1] += v[1]; w.db[4 * ocq + 2] += v[2]; w.db[4 * ocq + 3] += v[3];
  }
}

bool fill(WT& w, int mode, long* ws_floats) {
  const ConvP& p = w.c;
  const int taps = (mode == NLT_CONV1X1 || mode == NLT_DECONV_K2S2) ? 1 : 4;
  w.kq = taps * ((p.c0 + p.c1) >> 2);
  w.nq = p.N >> 2;
  w.kblocks = (w.kq + 15) / 16;
  w.nblocks = (w.nq + 15) / 16;
  static const long target = [] { const char* e = getenv("NLT_WGRAD_WANT"); return e ? atol(e) : 512l; }();
  long want = target / ((long)w.kblocks * w.nblocks);        // ~512 workgroups = the 2 per CU that are resident at once (measured:
                                                             // 1024 -> 512 takes 4 us off a 35 us mid-network launch: half the slices to reduce)
  if (want < 1) want = 1;
  long rows = (p.M + want - 1) / want;
  if (rows < 256) rows = 256;                                 // >= 16 MFMA steps per wave
  const bool walk = p.gw % 4 == 0 && !nlt_wgrad_generic_only();
  const long unit = walk ? 16 * WALK_PF : 16;                 // the walk kernel's runs: whole pipeline rounds
  rows = (rows + unit - 1) / unit * unit;
  w.msplits = (int)((p.M + rows - 1) / rows);
  if (w.msplits >= 8 && w.msplits % 8) {                      // whole XCD rounds if a slightly shorter slice gives them
    const long target = (w.msplits + 7) / 8 * 8;
    const long r2 = ((p.M + target - 1) / target + unit - 1) / unit * unit;
    if (r2 >= 16 * unit / 16 && (p.M + r2 - 1) / r2 == target) { rows = r2; w.msplits = (int)target; }
  }
  w.rows_per_split = (int)rows;
  w.xcd_order = w.msplits >= 8 && (w.msplits % 8 == 0 || w.msplits >= 24);
  *ws_floats = (long)w.msplits * w.kblocks * w.nblocks * 4096 + (long)w.msplits * w.nblocks * 64;
  return true;
}

template <int MODE>
int run(WT& w, hipStream_t s) {
  const long groups = (long)w.kblocks * w.nblocks * (w.xcd_order ? (w.msplits + 7) / 8 * 8 : w.msplits);   // (whole XCD rounds)
  w.zeros = (w.c.gw % 4 == 0 && !nlt_wgrad_generic_only()) ? nlt_zero_page() : nullptr;
  if (w.zeros)
    hipLaunchKernelGGL(wgrad_walk_kernel<MODE>, dim3((unsigned)groups), dim3(256), 0, s, w);
  else
    hipLaunchKernelGGL(wgrad_tile_kernel<MODE>, dim3((unsigned)groups), dim3(256), 0, s, w);
  const long items = (long)w.kblocks * w.nblocks * 4096;
  static const bool few_ok = [] { const char* e = getenv("NLT_WGRAD_REDUCE_FEW"); return !(e && e[0] == '0'); }();   // A/B switch
  if (w.msplits <= 8 && few_ok)
    hipLaunchKernelGGL((wgrad_reduce_kernel<MODE, true>), dim3((unsigned)(items / 1024 + (w.db ? w.c.cout / 4 : 0))), dim3(256), 0, s, w);
  else
    hipLaunchKernelGGL((wgrad_reduce_kernel<MODE, false>), dim3((unsigned)(items / 64 + (w.db ? w.c.cout / 4 : 0))), dim3(256), 0, s, w);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

int prepare(WT& w, int mode, const float* src0, int ld0, int c0, const float* src1, int ld1, int c1, int n, int h, int wd,
            const float* dpre, int ldp, int cout, float* dw, float* db, long* ws_floats) {
  float* dummy = dw ? dw : reinterpret_cast<float*>(16);
  const float* d0 = src0 ? src0 : dummy;
  const int st = nlt_fill_conv_params(w.c, mode, d0, ld0, c0, c1 ? (src1 ? src1 : dummy) : nullptr, ld1, c1, n, h, wd, dummy, dummy,
                                      cout, dummy, cout, 0, 0.f, nullptr, 0, 0);
  if (st != NLT_OK) return st;
  if ((c0 & 3) || (c1 & 3) || (cout & 3) || (ld0 & 3) || (c1 && (ld1 & 3)) || (ldp & 3) || ldp < cout) return NLT_ERR_UNSUPPORTED;
  if (w.c.gw < 4) return NLT_ERR_UNSUPPORTED;                      // the incremental row walk assumes >= 4 texels per grid row
  w.dp = dpre; w.ldp = ldp; w.dw = dw; w.db = db;
  fill(w, mode, ws_floats);
  return NLT_OK;
}

}  // namespace

extern "C" long nlt_wgrad_workspace_floats(int mode, int c0, int c1, int n, int h, int w, int cout) {
  WT t;
  long need = -1;
  if (prepare(t, mode, nullptr, c0 > 4 ? c0 : 4, c0, nullptr, c1 > 4 ? c1 : 4, c1, n, h, w, nullptr, cout, cout, nullptr, nullptr, &need) != NLT_OK)
    return -1;
  return need;
}

extern "C" int nlt_conv_backward_weights_tiled(int mode,
                                               const float* src0, int ld0, int c0, const float* src1, int ld1, int c1,
                                               int n, int h, int w, const float* dpre, int ldp, int cout,
                                               float* dw_keras, float* dbias, float* workspace, long workspace_floats,
                                               void* stream) {
  if (!src0 || !dpre || !dw_keras || !workspace) return NLT_ERR_BAD_ARG;
  if (c1 > 0 && !src1) return NLT_ERR_BAD_ARG;
  WT t;
  long need = 0;
  const int st = prepare(t, mode, src0, ld0, c0, src1, ld1, c1, n, h, w, dpre, ldp, cout, dw_keras, dbias, &need);
  if (st != NLT_OK) return st;
  if (workspace_floats < need) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(src0) || (c1 && !nlt_aligned16(src1)) || !nlt_aligned16(dpre) || !nlt_aligned16(workspace)) return NLT_ERR_BAD_ARG;
  t.ws = workspace;
  t.wsb = dbias ? workspace + (long)t.msplits * t.kblocks * t.nblocks * 4096 : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (mode) {
    case NLT_CONV1X1: return run<NLT_CONV1X1>(t, s);
    case NLT_CONV_K2S2: return run<NLT_CONV_K2S2>(t, s);
    case NLT_CONV_K2S1: return run<NLT_CONV_K2S1>(t, s);
    case NLT_DECONV_K2S2: return run<NLT_DECONV_K2S2>(t, s);
    case NLT_DECONV_K2S1: return run<NLT_DECONV_K2S1>(t, s);
  }
  return NLT_ERR_BAD_ARG;
}
