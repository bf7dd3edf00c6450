// Implicit-GEMM conv on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
//
// D[cout][texel] = W^T[cout][K] * X^T[K][texel]: the WEIGHTS are the MFMA "A" operand and the
// TEXELS the "B" operand, so each lane ends up holding 4 consecutive output channels of one
// texel -> one 16-byte NHWC store per lane, bias/LeakyReLU applied in registers.
//
// Lane l = (kk = l>>4, i = l&15).  Inside a K-chunk of 16 the k index is permuted so that lane
// kk owns channels 4kk..4kk+3: both operands are then fetched with ONE 16-byte load per lane
// (texels straight from the NHWC tensor, weights from the pre-packed fragment array) and feed
// four consecutive MFMA k-steps.  No LDS, no barriers: every texel element is used by exactly
// one wave (register-level reuse across its CT column tiles) and the weights are L2-resident.
//
// Split-K (few GEMM rows, long K: the deep levels and the small released shapes) -- ONE launch since r06.  A workgroup is one
// output tile x one group of NW K slices (NW = 4 waves, or 16 for >= 16 slices of a narrow tile): its waves each walk a slice,
// meet in LDS (slice order), and
//   * ksplit <= NW: wave 0 finishes the tile (bias / map / activation) -- no workspace, no second pass;
//   * ksplit  > NW: wave 0 writes the group's partial tile to the workspace, releases it (agent scope) and draws a ticket from the
//     tile's counter; the workgroup that draws the last ticket acquires, its waves add the groups' slabs in group order
//     (wave w: groups w, w + NW, ...; then the NW sums in wave order: the same order whoever arrives last), finishes the tile and
//     puts the counter back to zero.  The order of additions is a function of the launch shape only: bit-reproducible.
// Rounds 2-5 ran a second launch (`splitk_epilogue*_kernel`) for this: 20 of the 62 launches of a depth-1024 / 256^2 forward.
// Cross-workgroup visibility follows cdna_hip_programming.md section 6 G16 (per-XCD L2s are not coherent): plain slab stores ->
// s_waitcnt vmcnt(0) -> barrier -> one lane: release fence (agent) -> restated wait -> relaxed agent fetch_add; last arriver: one
// lane acquire fence (agent) -> barrier -> plain loads.
#include "nlt_common.h"
#include "pack_common.h"

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;

__host__ __device__ inline int chunks16(int c) { return (c + 15) >> 4; }

template <int MODE>
__device__ __forceinline__ int keras_widx(int t, int c, int ncol, int cin, int cout) {
  if (MODE == NLT_CONV1X1 || MODE == NLT_CONV_K2S2 || MODE == NLT_CONV_K2S1) return (t * cin + c) * cout + ncol;
  if (MODE == NLT_DECONV_K2S1) return (t * cout + ncol) * cin + c;
  return ncol * cin + c;
}

template <int MODE>
__global__ void pack_weights_kernel(const float* __restrict__ wk, int c0, int c1, int cout, int N,
                                    int ntiles, long total, float* __restrict__ wp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  wp[idx] = nlt_mfma_fragment<MODE>(wk, idx, c0, c1, cout, N, ntiles, cout, 0);
}

// k2s1 families on a 1 x 1 texel grid: taps 1-3 only ever see zero padding, so their weights (3/4 of the layer) are
// neither streamed nor multiplied; tap 0's chunks come first in the packed order.
template <int MODE>
__host__ __device__ inline int live_taps(const ConvP& p) {
  return ((MODE == NLT_CONV_K2S1 || MODE == NLT_DECONV_K2S1) && p.gh == 1 && p.gw == 1) ? 1 : ConvTraits<MODE>::TAPS;
}

// observation half of dfm[l] (ConvP::split_*): dmean (+ the observation path's own gradient) times LeakyReLU'(obs y)
__device__ __forceinline__ void split_store(const ConvP& p, int otex, int oc, f32x4 v) {
  const size_t at = (size_t)otex * p.split_c + (oc - p.split_c);
  f32x4* d = reinterpret_cast<f32x4*>(p.split_d + at);
  if (p.split_partial) v += *d;
  const f32x4 mk = *reinterpret_cast<const f32x4*>(p.split_y + at);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] *= (mk[j] > 0.f) ? 1.f : p.split_alpha;
  *d = v;
}

template <int MODE, int RT, int CT, int PF, int NW>           // NW = 0: no split, 4 independent waves; NW = 4 | 16: split-K workgroup
__global__ __launch_bounds__(NW ? 64 * NW : 256) void conv_mfma_kernel(ConvP p, int mtiles, int ngroups, int ntiles, int ksplit,
                                                        float* ws) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int ng, mt, ks, kg = 0, tile = 0;
  constexpr bool SPLIT = NW > 0;
  if constexpr (SPLIT) {                         // workgroup = (tile, group of NW K slices); wave = slice in the group
    const int kgroups = (ksplit + NW - 1) / NW;
    tile = blockIdx.x / kgroups;
    kg = blockIdx.x - tile * kgroups;
    ng = tile % ngroups;
    mt = tile / ngroups;
    ks = kg * NW + wv;                            // (ks >= ksplit: an empty slice, zeros -- the wave still meets the barriers)
  } else {
    int wave = blockIdx.x * 4 + wv;
    ng = wave % ngroups; wave /= ngroups;
    mt = wave % mtiles;
    ks = wave / mtiles;                          // (two-launch split-K: every K slice a wave of its own)
    if (ks >= ksplit) return;
  }
  const int px = lane & 15, kk = lane >> 4;

  int rf[RT], ry[RT], rx[RT], rm[RT];
  bool rv[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int m = (mt * RT + rt) * 16 + px;
    rv[rt] = m < p.M;
    const int mc = rv[rt] ? m : p.M - 1;
    rm[rt] = mc;
    rx[rt] = mc % p.gw;
    ry[rt] = (mc / p.gw) % p.gh;
    rf[rt] = mc / (p.gw * p.gh);
  }

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(p.wgt) + (size_t)(ng * CT) * 64 + lane;
  const size_t wstride = (size_t)ntiles * 64;   // f32x4 per K-chunk
  const int ch0 = chunks16(p.c0), ch1 = chunks16(p.c1);
  const int cps = ch0 + ch1;                    // K-chunks per tap: source 0 then source 1
  const int total = live_taps<MODE>(p) * cps;
  const int per = (total + ksplit - 1) / ksplit;
  const int kbeg = ks * per < total ? ks * per : total;
  const int kend = kbeg + per < total ? kbeg + per : total;

  // Fragment loads are UNCONDITIONAL (clamped addresses, value selected afterwards): a predicated
  // load becomes an exec-masked branch per load and serialises the wave's memory requests.
  // The next chunk's fragments are requested before the current chunk's MFMAs are issued.
  bool tv[RT];
  const float* p0[RT];
  const float* p1[RT];
  auto set_tap = [&](int t) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int tex = rv[rt] ? conv_tap_texel<MODE>(p, rf[rt], ry[rt], rx[rt], t) : -1;
      tv[rt] = tex >= 0;
      const size_t tx = tv[rt] ? (size_t)tex : 0;
      p0[rt] = p.src0 + tx * p.ld0 + 4 * kk;
      p1[rt] = p.c1 ? p.src1 + tx * p.ld1 + 4 * kk : p0[rt];
    }
  };
  auto load_frags = [&](int r, int kci, f32x4 (&a)[CT], f32x4 (&b)[RT]) {
    const bool s = r >= ch0;
    const int k0 = (s ? r - ch0 : r) << 4;
    const bool kin = (k0 + 4 * kk) < (s ? p.c1 : p.c0);
    const int koff = kin ? k0 : -4 * kk;      // out-of-segment lanes re-read the row start
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 v = *reinterpret_cast<const f32x4*>((s ? p1[rt] : p0[rt]) + koff);
      const bool ok = kin && tv[rt];
      b[rt] = (f32x4){ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f};
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) a[ct] = wp[(size_t)kci * wstride + ct * 64];
  };
  if constexpr (PF == 2) {
    f32x4 a_cur[CT], b_cur[RT], a_nxt[CT], b_nxt[RT];
    int t_n = kbeg / cps, r_n = kbeg - t_n * cps;   // (tap, chunk-in-tap) of the NEXT load
    if (kbeg < kend) {
      set_tap(t_n);
      load_frags(r_n, kbeg, a_cur, b_cur);
    }
    for (int kc = kbeg; kc < kend; ++kc) {
      if (kc + 1 < kend) {
        if (++r_n == cps) { r_n = 0; set_tap(++t_n); }
        load_frags(r_n, kc + 1, a_nxt, b_nxt);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[ct][s4], b_cur[rt][s4], acc[rt][ct], 0, 0, 0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) a_cur[ct] = a_nxt[ct];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) b_cur[rt] = b_nxt[rt];
    }

  } else {
    // Fragment loads run PF - 1 chunks ahead of the MFMAs (PF register sets).  With one chunk of lookahead a wave of the
    // mid-network launches spent 36 % of its cycles waiting for memory and the matrix pipe sat at 0.33 (PMC, level-4
    // backward-data, 2 waves per SIMD).  Loads stay UNCONDITIONAL (past the slice they re-read its last chunk and the texel
    // operand is zeroed): a branch around them makes the wait-count insertion drain every outstanding load.
    f32x4 af[PF][CT], bf[PF][RT];
    int t_n = kbeg / cps, r_n = kbeg - t_n * cps;   // (tap, chunk-in-tap) of the NEXT load
    int kload = kbeg;
    if (kbeg < kend) set_tap(t_n);
    auto issue = [&](f32x4 (&a)[CT], f32x4 (&b)[RT]) {
      const bool live = kload < kend;               // wave-uniform
      if (live && kload > kbeg) {
        if (++r_n == cps) { r_n = 0; set_tap(++t_n); }
      }
      load_frags(r_n, live ? kload : kend - 1, a, b);
      if (!live) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) b[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      ++kload;
    };
    auto compute = [&](const f32x4 (&a)[CT], const f32x4 (&b)[RT]) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct][s4], b[rt][s4], acc[rt][ct], 0, 0, 0);
    };
    if (kbeg < kend) {
#pragma unroll
      for (int j = 0; j < PF - 1; ++j) issue(af[j], bf[j]);
      for (int kc = kbeg; kc < kend; kc += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          issue(af[(j + PF - 1) % PF], bf[(j + PF - 1) % PF]);
          compute(af[j], bf[j]);
        }
      }
    }

  }
  if constexpr (SPLIT) {
    __shared__ f32x4 red[(NW - 1) * RT * CT * 64 + 1];          // waves 1..NW-1's tiles; the last slot carries "this workgroup finishes"
    const int kgroups = (ksplit + NW - 1) / NW;
    auto meet = [&]() {                                         // acc of wave 0 = sum over the waves, wave order
      if (wv) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) red[((wv - 1) * RT * CT + rt * CT + ct) * 64 + lane] = acc[rt][ct];
      }
      __syncthreads();
      if (!wv) {
#pragma unroll
        for (int j = 0; j < NW - 1; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] += red[(j * RT * CT + rt * CT + ct) * 64 + lane];
      }
    };
    meet();
    if (kgroups > 1) {
      gu32* cnt = (gu32*)ws;                                     // (agent-scope words: global address space, never flat)
      float* slabs = ws + NLT_SPLITK_COUNTERS;
      const int npad = ntiles * 16;
      const size_t slab = (size_t)p.M * npad;
      if (!wv) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int ncol = (ng * CT + ct) * 16 + kk * 4;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            if (rv[rt]) *reinterpret_cast<f32x4*>(slabs + (size_t)kg * slab + (size_t)rm[rt] * npad + ncol) = acc[rt][ct];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                          // (also: every wave is done reading `red`)
      int* flag = reinterpret_cast<int*>(&red[(NW - 1) * RT * CT * 64]);
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (restated: ROCm 7.2 may drop the wait behind buffer_wbl2)
        const unsigned t = __hip_atomic_fetch_add(cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == (unsigned)(kgroups - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *flag = last;
      }
      __syncthreads();
      if (!*flag) return;
      // the last workgroup of this tile: wave w adds groups w, w + NW, ... (group order), then the waves meet in wave order
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll(NW == 4 ? 4 : 1)
      for (int g = wv; g < kgroups; g += NW) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int ncol = (ng * CT + ct) * 16 + kk * 4;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)                       // (clamped rows re-read a valid row; never stored)
            acc[rt][ct] += *reinterpret_cast<const f32x4*>(slabs + (size_t)g * slab + (size_t)rm[rt] * npad + ncol);
        }
      }
      meet();
      if (threadIdx.x == 0) __hip_atomic_store(cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    }
    if (wv) return;
  }

  if constexpr (!SPLIT) {
    if (ksplit > 1) {                           // raw partial sums -> workspace [ks][M][ntiles*16]; the second launch finishes
      float* part = ws + NLT_SPLITK_COUNTERS;
      const int npad = ntiles * 16;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int ncol = (ng * CT + ct) * 16 + kk * 4;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (rv[rt]) *reinterpret_cast<f32x4*>(part + ((size_t)ks * p.M + rm[rt]) * npad + ncol) = acc[rt][ct];
      }
      return;
    }
  }

  // Epilogue: lane holds outputs [ncol, ncol+4) of texel px for every (rt, ct).
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int ncol = (ng * CT + ct) * 16 + kk * 4;
    if (ncol >= p.N) continue;
    int oc = ncol, ab = 0;
    if (MODE == NLT_DECONV_K2S2) { ab = ncol / p.cout; oc = ncol - ab * p.cout; }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + oc);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (!rv[rt]) continue;
      int otex = rm[rt];
      if (MODE == NLT_DECONV_K2S2) otex = (rf[rt] * p.oh + 2 * ry[rt] + (ab >> 1)) * p.ow + 2 * rx[rt] + (ab & 1);
      f32x4 v = acc[rt][ct] + bv;
      if (p.bmap) v += *reinterpret_cast<const f32x4*>(p.bmap + (size_t)(p.bmap_mod ? otex % p.bmap_mod : otex) * p.cout + oc);
      f32x4* o = reinterpret_cast<f32x4*>(p.out + (size_t)otex * p.ldo + oc);
      if (p.accumulate) v += *o;
      if (p.split_c && oc >= p.split_c) { split_store(p, otex, oc, v); continue; }
      if (p.mask_src) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask_src + (size_t)otex * p.ldm + oc);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= (mk[j] > 0.f) ? 1.f : p.alpha;
      } else if (p.act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : p.alpha * v[j];
      }
      *o = v;
    }
  }
}

// bias / accumulate / mask / LeakyReLU and the mode's output addressing of one output quad, exactly as the single-pass epilogue
template <int MODE>
__device__ __forceinline__ void splitk_finish(const ConvP& p, int m, int ncol, f32x4 v) {
  int oc = ncol, ab = 0;
  if (MODE == NLT_DECONV_K2S2) { ab = ncol / p.cout; oc = ncol - ab * p.cout; }
  int otex = m;
  if (MODE == NLT_DECONV_K2S2) {
    const int x = m % p.gw, y = (m / p.gw) % p.gh, f = m / (p.gw * p.gh);
    otex = (f * p.oh + 2 * y + (ab >> 1)) * p.ow + 2 * x + (ab & 1);
  }
  v += *reinterpret_cast<const f32x4*>(p.bias + oc);
  if (p.bmap) v += *reinterpret_cast<const f32x4*>(p.bmap + (size_t)(p.bmap_mod ? otex % p.bmap_mod : otex) * p.cout + oc);
  f32x4* o = reinterpret_cast<f32x4*>(p.out + (size_t)otex * p.ldo + oc);
  if (p.accumulate) v += *o;
  if (p.split_c && oc >= p.split_c) { split_store(p, otex, oc, v); return; }
  if (p.mask_src) {
    const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask_src + (size_t)otex * p.ldm + oc);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= (mk[j] > 0.f) ? 1.f : p.alpha;
  } else if (p.act) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : p.alpha * v[j];
  }
  *o = v;
}

// TWO-LAUNCH split-K (ksplit < 0; rounds 2-5's only form): independent waves write their slices' raw partial sums, this second
// launch sums them in slice order (deterministic), then the usual epilogue.  Kept beside the one-launch form because it is the
// faster one where a handful of GEMM rows meet 64-128 slices (depth 1024 below 4 x 4 texels: every slice a wave of its own on any
// CU, the sum spread over thousands of threads; tools/bench_deep.py) -- the plan-time trials choose per launch.  Few slices
// (the mid-network shapes: thousands of rows, 4-8 slices): one thread per output quad walks them.
template <int MODE>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(ConvP p, int ntiles, int ksplit, const float* __restrict__ ws0) {
  const float* __restrict__ ws = ws0 + NLT_SPLITK_COUNTERS;
  const int quads = p.N >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.M * quads) return;
  const int m = idx / quads;
  const int ncol = (idx - (long)m * quads) * 4;
  const int npad = ntiles * 16;
  f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)m * npad + ncol);
  for (int ks = 1; ks < ksplit; ++ks) v += *reinterpret_cast<const f32x4*>(ws + ((size_t)ks * p.M + m) * npad + ncol);
  splitk_finish<MODE>(p, m, ncol, v);
}

// Many slices (the deep levels: a handful of rows, 16-128 slices):
template <int MODE>
__global__ __launch_bounds__(256) void splitk_epilogue_wide_kernel(ConvP p, int ntiles, int ksplit, const float* __restrict__ ws0) {
  const float* __restrict__ ws = ws0 + NLT_SPLITK_COUNTERS;
  // 32 output quads per workgroup x 8 slice lanes: lane group j adds slices j, j + 8, ... (four independent running
  // sums: a serial walk over 64-128 slices is pure load latency), the 8 partial sums meet in LDS in a fixed order.
  __shared__ f32x4 part[8][32];
  const int quads = p.N >> 2;
  const int il = threadIdx.x & 31, ksl = threadIdx.x >> 5;
  const long idx = (long)blockIdx.x * 32 + il;
  const bool live = idx < (long)p.M * quads;
  const int m = live ? idx / quads : 0;
  const int ncol = live ? (idx - (long)m * quads) * 4 : 0;
  const int npad = ntiles * 16;
  const size_t slice = (size_t)p.M * npad;
  const float* src = ws + (size_t)m * npad + ncol;
  f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int ks = ksl;
  for (; ks + 24 < ksplit; ks += 32) {
    s0 += *reinterpret_cast<const f32x4*>(src + (size_t)ks * slice);
    s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(ks + 8) * slice);
    s2 += *reinterpret_cast<const f32x4*>(src + (size_t)(ks + 16) * slice);
    s3 += *reinterpret_cast<const f32x4*>(src + (size_t)(ks + 24) * slice);
  }
  for (; ks < ksplit; ks += 8) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)ks * slice);
  part[ksl][il] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ksl || !live) return;
  f32x4 v = part[0][il];
#pragma unroll
  for (int j = 1; j < 8; ++j) v += part[j][il];
  splitk_finish<MODE>(p, m, ncol, v);
}

int taps_of(int mode) { return (mode == NLT_CONV1X1 || mode == NLT_DECONV_K2S2) ? 1 : 4; }

template <int MODE, int RT, int CT>
int launch_tile(const ConvP& p, int ksplit, float* ws, hipStream_t s) {
  const int ntiles = (p.N + 15) >> 4;
  const int ngroups = ntiles / CT;
  const int mtiles = (p.M + 16 * RT - 1) / (16 * RT);
  const int total = live_taps<MODE>(p) * (chunks16(p.c0) + chunks16(p.c1));
  const bool two_launches = ksplit < 0;                           // (negative: the two-launch form)
  if (two_launches) ksplit = -ksplit;
  if (ksplit > total) ksplit = total;
  if (ksplit < 1 || !ws) ksplit = 1;
  const long tiles = (long)mtiles * ngroups;
  if (two_launches && ksplit > 1) {
    const unsigned blocks = (unsigned)((tiles * ksplit + 3) / 4);
    if ((total + ksplit - 1) / ksplit >= 24)
      hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, 3, 0>), dim3(blocks), dim3(256), 0, s, p, mtiles, ngroups, ntiles, ksplit, ws);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, 2, 0>), dim3(blocks), dim3(256), 0, s, p, mtiles, ngroups, ntiles, ksplit, ws);
    const long items = (long)p.M * (p.N >> 2);
    if (ksplit > 8)
      hipLaunchKernelGGL(splitk_epilogue_wide_kernel<MODE>, dim3((unsigned)((items + 31) / 32)), dim3(256), 0, s, p, ntiles, ksplit, ws);
    else
      hipLaunchKernelGGL(splitk_epilogue_kernel<MODE>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, p, ntiles, ksplit, ws);
    NLT_CHECK_LAUNCH();
    return NLT_OK;
  }
  if (ksplit > 4 && tiles > NLT_SPLITK_COUNTERS) ksplit = 4;     // (one ticket counter per tile; such a launch has waves enough)
  // long K loops (>= 24 sixteen-channel chunks per wave): three register sets, loads two chunks ahead; short ones keep the
  // two-set loop (a deeper pipeline costs them its prologue and up to two zero-operand rounds: measured slower below ~16 chunks)
  const int per_wave = (total + ksplit - 1) / ksplit;
  if (ksplit > 1) {
    // 16 slices and more on a wave tile of up to four fragments: sixteen-wave workgroups (the whole reduction stays in LDS up to 16 slices, and a
    // deep level's 64-128 slices meet in memory as 4-8 partial tiles instead of 16-32)
    if constexpr (RT * CT <= 4) {
      if (ksplit >= 16) {
        const unsigned blocks = (unsigned)(tiles * ((ksplit + 15) >> 4));
        if (per_wave >= 24 && RT * CT <= 2)
          hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, (RT * CT <= 2 ? 3 : 2), 16>), dim3(blocks), dim3(1024), 0, s, p, mtiles, ngroups, ntiles, ksplit, ws);
        else
          hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, 2, 16>), dim3(blocks), dim3(1024), 0, s, p, mtiles, ngroups, ntiles, ksplit, ws);
        NLT_CHECK_LAUNCH();
        return NLT_OK;
      }
    }
    const unsigned blocks = (unsigned)(tiles * ((ksplit + 3) >> 2));
    if (per_wave >= 24)
      hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, 3, 4>), dim3(blocks), dim3(256), 0, s, p, mtiles, ngroups, ntiles, ksplit, ws);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, 2, 4>), dim3(blocks), dim3(256), 0, s, p, mtiles, ngroups, ntiles, ksplit, ws);
  } else {
    const unsigned blocks = (unsigned)((tiles + 3) / 4);
    if (per_wave >= 24)
      hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, 3, 0>), dim3(blocks), dim3(256), 0, s, p, mtiles, ngroups, ntiles, 1, ws);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MODE, RT, CT, 2, 0>), dim3(blocks), dim3(256), 0, s, p, mtiles, ngroups, ntiles, 1, ws);
  }
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

template <int MODE>
int launch_mode(const ConvP& p, int tile_hint, int ksplit, float* ws, hipStream_t s) {
  const int ntiles = (p.N + 15) >> 4;
  int RT = 0, CT = 0;
  if (tile_hint > 0) { RT = tile_hint >> 4; CT = tile_hint & 15; }
  else {
    // largest wave tile that still gives the chip >= 2 waves per SIMD (1024 SIMDs)
    static const int cand[9][2] = {{4, 4}, {2, 4}, {4, 2}, {2, 2}, {1, 4}, {4, 1}, {1, 2}, {2, 1}, {1, 1}};
    RT = 1; CT = 1;
    for (int i = 0; i < 9; ++i) {
      const int r = cand[i][0], c = cand[i][1];
      if (ntiles % c) continue;
      const long waves = (long)((p.M + 16 * r - 1) / (16 * r)) * (ntiles / c);
      if (waves >= 2048) { RT = r; CT = c; break; }
    }
  }
  if (CT <= 0 || ntiles % CT) return NLT_ERR_UNSUPPORTED;
#define NLT_TILE(R, C) if (RT == R && CT == C) return launch_tile<MODE, R, C>(p, ksplit, ws, s);
  NLT_TILE(4, 4) NLT_TILE(2, 4) NLT_TILE(4, 2) NLT_TILE(2, 2) NLT_TILE(1, 4)
  NLT_TILE(4, 1) NLT_TILE(1, 2) NLT_TILE(2, 1) NLT_TILE(1, 1)
#undef NLT_TILE
  return NLT_ERR_UNSUPPORTED;
}

}  // namespace

bool nlt_conv_mfma_supported(int mode, const ConvP& p) {
  (void)mode;
  if ((p.c0 & 3) || (p.c1 & 3) || (p.cout & 3) || (p.ld0 & 3) || (p.ldo & 3)) return false;
  if (p.c1 && (p.ld1 & 3)) return false;
  if (p.mask_src && (p.ldm & 3)) return false;
  if (!nlt_aligned16(p.src0) || !nlt_aligned16(p.out) || !nlt_aligned16(p.bias) || !nlt_aligned16(p.wgt)) return false;
  if (p.c1 && !nlt_aligned16(p.src1)) return false;
  if (p.mask_src && !nlt_aligned16(p.mask_src)) return false;
  return true;
}

int nlt_conv_mfma_launch(int mode, const ConvP& p, int tile_hint, hipStream_t s, int ksplit, float* ws) {
  if (!nlt_conv_mfma_supported(mode, p)) return NLT_ERR_UNSUPPORTED;
  switch (mode) {
    case NLT_CONV1X1: return launch_mode<NLT_CONV1X1>(p, tile_hint, ksplit, ws, s);
    case NLT_CONV_K2S2: return launch_mode<NLT_CONV_K2S2>(p, tile_hint, ksplit, ws, s);
    case NLT_CONV_K2S1: return launch_mode<NLT_CONV_K2S1>(p, tile_hint, ksplit, ws, s);
    case NLT_DECONV_K2S2: return launch_mode<NLT_DECONV_K2S2>(p, tile_hint, ksplit, ws, s);
    case NLT_DECONV_K2S1: return launch_mode<NLT_DECONV_K2S1>(p, tile_hint, ksplit, ws, s);
  }
  return NLT_ERR_BAD_ARG;
}

extern "C" long nlt_packed_weight_floats(int mode, int c0, int c1, int cout) {
  if (mode < NLT_CONV1X1 || mode > NLT_DECONV_K2S1 || c0 <= 0 || c1 < 0 || cout <= 0) return -1;
  const int N = (mode == NLT_DECONV_K2S2) ? 4 * cout : cout;
  return (long)taps_of(mode) * (chunks16(c0) + chunks16(c1)) * ((N + 15) >> 4) * 256;
}

extern "C" int nlt_pack_conv_weights(int mode, const float* w_keras, int c0, int c1, int cout,
                                     float* w_packed, void* stream) {
  if (!w_keras || !w_packed) return NLT_ERR_BAD_ARG;
  const long total = nlt_packed_weight_floats(mode, c0, c1, cout);
  if (total <= 0) return NLT_ERR_BAD_ARG;
  const int N = (mode == NLT_DECONV_K2S2) ? 4 * cout : cout;
  const int ntiles = (N + 15) >> 4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)((total + 255) / 256);
#define NLT_PACK(MODE) hipLaunchKernelGGL(pack_weights_kernel<MODE>, dim3(blocks), dim3(256), 0, s, w_keras, c0, c1, cout, N, ntiles, total, w_packed)
  switch (mode) {
    case NLT_CONV1X1: NLT_PACK(NLT_CONV1X1); break;
    case NLT_CONV_K2S2: NLT_PACK(NLT_CONV_K2S2); break;
    case NLT_CONV_K2S1: NLT_PACK(NLT_CONV_K2S1); break;
    case NLT_DECONV_K2S2: NLT_PACK(NLT_DECONV_K2S2); break;
    case NLT_DECONV_K2S1: NLT_PACK(NLT_DECONV_K2S1); break;
  }
#undef NLT_PACK
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
