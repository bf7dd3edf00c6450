// Second generation of the Winograd F(2x2, 2x2) stride-1 k2 conv (conv_wino.hip has the algebra and the first, register-staged
// kernel).  Measured on the first generation (profiles/README.md r04: switching its phases off one at a time): the matrix loop
// alone 0.061 ms, the staging alone 0.053 ms, together 0.080 ms at 64 channels -- a stage's loads were requested ONE stage
// (1.1 us of MFMAs) before their use against ~3 us of loaded HBM latency, every thread carried 36 staging registers next to 144
// accumulators (no room to look further ahead), and the transformed windows went through LDS (2.25x the raw bytes).  Here:
//
//   * the RAW haloed tile (9 x 33 texels x 8 channels per stage) and the stage's pre-transformed weights go global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass); raw tiles are requested THREE stages ahead into a
//     ring of four buffers, weights (L2-resident) one stage ahead into two; each wave waits for its own pieces with a COUNTED
//     s_waitcnt vmcnt(n) that leaves the newest raw request in flight, then one s_barrier per stage orders everybody's pieces;
//   * the input transform B^T d B happens in registers, per wave, right before the MFMAs: lane (kk, j) reads the 3 x 3 window of
//     block j for its two channels (nine ds_read_b64: the raw tile is stored planar by channel quad and SPLIT BY x PARITY, so
//     the 16 blocks of a row read 16 consecutive slots -- conflict-free) and forms the nine B operands with 12 subtractions;
//   * a wave owns one block row x 32 output channels: 72 accumulator registers (+ 32 for the running observation mean), so a
//     64-channel workgroup is 8 waves (4 block rows x 2 channel halves, one workgroup per CU, two waves per SIMD) and a
//     32-channel one 4 waves (two to three workgroups per CU); the observation mean stays in registers in BOTH forms.
//
// DMA pieces (1 KiB = 64 slots each) per 8-channel stage: raw 10 (4 planes of 160 slots: [quad 2][x parity 2][9 rows x 17]),
// weights 18 / 9; piece k of the stage belongs to wave k mod (waves).  Padding texels and the unused slots of a plane read a
// page of zeros.  Stores of the epilogue share the VM counter with the DMA pieces and may retire out of order with them: the
// wait after an epilogue is a full drain.
#include "nlt_common.h"
#include "pack_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BY = 4, BX = 16;               // blocks per workgroup: 4 rows x 16 columns = 8 x 32 output texels
constexpr int PLANE = 160;                   // slots per (channel quad, x parity) plane: 9 rows x 17 = 153, padded (2 PLANE = 0 mod 16)
constexpr int RAW_SLOTS = 4 * PLANE;         // per 8-channel stage
constexpr int NRP = RAW_SLOTS / 64;          // raw DMA pieces per stage
constexpr int RAW_DEPTH = 4, PF = 3;         // ring of raw buffers, stages of prefetch
constexpr int U_DEPTH = 3;                  // ring of weight buffers (requested two stages ahead)

struct Wino2P {
  const float* src; const float* packed; const float* bias; const float* zeros;
  float* out; float* mean_out;
  int ld, cin, frames, kobs, h, w;
  int cout, ldo, ldm;
  int tiles_y, tiles_x, nc8;
  int act; float alpha;
  const float* mask_src; int ld_mask; int accumulate;
};

__device__ __forceinline__ int xcd_tile_w2(int b, int nblocks) {
  return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3);
}

// One ds_read_b64 per call, never fused with a neighbour: hipcc merges two 8-byte LDS loads of a lane into ds_read2_b64 /
// ds_read2st64_b64, which the LDS services 16 lanes at a time over 32 banks (MI355X_MICROARCH.md, LDS table) -- the 16 blocks of
// a row, 16 bytes apart, then collide two by two (r04 PMC: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).  The plain
// ds_read_b64 covers 32 lanes x 8 bytes = 256 contiguous bytes per LDS cycle: conflict-free in this layout.
__device__ __forceinline__ f32x2 lds_b64(const char* p) {
  typedef const volatile __attribute__((address_space(3))) f32x2 lds_f32x2;
  return *(lds_f32x2*)(lds_void*)p;
}

__device__ __forceinline__ void dma16(const float* g, f32x4* l) {
  __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
}

template <bool TR, int WNW, bool MEAN>
__global__ __launch_bounds__(256 * WNW, 2) void conv_wino2_kernel(Wino2P p) {
  constexpr int NW = 4 * WNW, TNT = 2 * WNW, CT = 2;
  constexpr int U_SLOTS = 9 * TNT * 2 * 16, NUP = U_SLOTS / 64;
  constexpr int MR = (NRP + NW - 1) / NW, MU = (NUP + NW - 1) / NW;
  constexpr int U_BASE = RAW_DEPTH * RAW_SLOTS;
  __shared__ f32x4 lds[U_BASE + U_DEPTH * U_SLOTS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 4, j = lane & 15;
  const int wr = wave & 3, wc = wave >> 2;
  int tile = xcd_tile_w2(blockIdx.x, gridDim.x);
  const int tx0 = (tile % p.tiles_x) * (2 * BX); tile /= p.tiles_x;
  const int ty0 = (tile % p.tiles_y) * (2 * BY);
  const int f = tile / p.tiles_y;
  const int g = blockIdx.y;
  const int total = p.nc8 * p.kobs;
  const long in_frame = (long)p.h * p.w * p.ld;

  // ---- this lane's part of the wave's raw pieces (k = wave + m NW): source texel of slot k * 64 + lane
  int raw_off[MR]; bool raw_ok[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int r = (wave + m * NW) * 64 + lane;
    const int plane = r / PLANE, idx = r - plane * PLANE;
    const int row = idx / 17, xh = idx - row * 17;
    const int x = 2 * xh + (plane & 1);
    const int gy = ty0 + row - (TR ? 1 : 0), gx = tx0 + x - (TR ? 1 : 0);
    raw_ok[m] = wave + m * NW < NRP && idx < 153 && x <= 2 * BX && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
    raw_off[m] = raw_ok[m] ? (gy * p.w + gx) * p.ld + (plane >> 1) * 4 : 0;
  }
  const int n_raw = (NRP - 1 - wave) / NW + 1;                          // raw pieces this wave issues per stage (wave < NRP always)
  const int u0 = ((wave - NRP) % NW + NW) % NW;                         // this wave's first weight piece

  // stage counters of the two request streams (clamped at the last stage: the requests past the end re-read it into a buffer
  // nobody reads any more, which keeps every wave's VM count per iteration constant)
  int ri = 0, rc = 0, uc = 0, ut = 0, rt_ = 0;
  auto issue_raw = [&](int buf) {
    const float* fp = p.src + (long)(f * p.kobs + ri) * in_frame + rc * 8;
#pragma unroll
    for (int m = 0; m < MR; ++m)
      if (wave + m * NW < NRP) dma16(raw_ok[m] ? fp + raw_off[m] : p.zeros, lds + buf * RAW_SLOTS + (wave + m * NW) * 64);
    if (rt_ + 1 < total) { ++rt_; if (++rc == p.nc8) { rc = 0; ++ri; } }
  };
  auto issue_u = [&](int buf) {
    const float* up = p.packed + ((long)g * p.nc8 + uc) * (U_SLOTS * 4) + lane * 4;
#pragma unroll
    for (int m = 0; m < MU; ++m)
      if (u0 + m * NW < NUP) dma16(up + (u0 + m * NW) * 256, lds + U_BASE + buf * U_SLOTS + (u0 + m * NW) * 64);
    if (ut + 1 < total) { ++ut; if (++uc == p.nc8) uc = 0; }
  };

  f32x4 acc[9][CT], mean[MEAN ? CT : 1][4];
#pragma unroll
  for (int ps = 0; ps < 9; ++ps)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ps][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (MEAN) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int u = 0; u < 4; ++u) mean[ct][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // byte offsets of this lane's LDS reads: window texel (r, sx) of block j in row wr -> plane (quad kk >> 1, parity sx & 1),
  // slot (2 wr + r) * 17 + j + (sx >> 1); weight fragment (position, column tile) -> slot ((ps * TNT + ct) * 2 + (kk >> 1)) * 16 + j
  const int half = (kk & 1) * 8;
  const int wbase = ((kk >> 1) * 2 * PLANE + (2 * wr) * 17 + j) * 16 + half;
  const int abase = U_BASE * 16 + ((wc * CT * 2 + (kk >> 1)) * 16 + j) * 16 + half;

#pragma unroll
  for (int s = 0; s < PF; ++s) issue_raw(s);
  issue_u(0); issue_u(1);
  int fc = 0, fi = 0;                                                    // stage within the frame, observation frame
  int ubi = 2, ubr = 0;                                                  // weight ring: next buffer to request into / to read from

  // The fragments of stage s + 1 are READ (LDS -> registers) under the MFMAs of stage s: a stage's reads used to sit between its
  // barrier and its first MFMA with every wave of the CU at that same point -- the matrix pipe idle for the LDS round trip (r04
  // PMC: 31 % of the wave cycles parked at waitcnt / barrier).  So a stage's data has to be in LDS one barrier EARLIER: weights
  // are requested two stages ahead into a ring of three, the raw tile three ahead into four (as before).
  auto read_frags = [&](int s, f32x2 (&d)[9], f32x2 (&af)[9][CT]) {
    const char* R = reinterpret_cast<const char*>(lds) + (s & (RAW_DEPTH - 1)) * (RAW_SLOTS * 16) + wbase;
    const char* A = reinterpret_cast<const char*>(lds) + ubr * (U_SLOTS * 16) + abase;
    if (++ubr == U_DEPTH) ubr = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int sx = 0; sx < 3; ++sx)
        d[r * 3 + sx] = lds_b64(R + ((sx & 1) * PLANE + r * 17 + (sx >> 1)) * 16);
#pragma unroll
    for (int ps = 0; ps < 9; ++ps)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) af[ps][ct] = lds_b64(A + ((ps * TNT + ct) * 2 * 16) * 16);
  };
  auto transform = [&](f32x2 (&d)[9]) {                                  // B^T d B in place: rows, then columns
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) { d[sx] = d[sx] - d[3 + sx]; d[6 + sx] = d[6 + sx] - d[3 + sx]; }
#pragma unroll
    for (int x = 0; x < 3; ++x) { d[x * 3] = d[x * 3] - d[x * 3 + 1]; d[x * 3 + 2] = d[x * 3 + 2] - d[x * 3 + 1]; }
  };

  f32x2 v0[9], a0[9][CT], v1[9], a1[9][CT];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(0, v0, a0);
  transform(v0);
  bool drain = false;

  auto stage = [&](int s, f32x2 (&vc)[9], f32x2 (&ac)[9][CT], f32x2 (&vn)[9], f32x2 (&an)[9][CT]) {
    if (s > 0) {
      if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (n_raw == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (n_raw == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      drain = false;
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    issue_u(ubi);                                                        // weights of stage s + 2, then the raw tile of stage s + 3:
    if (++ubi == U_DEPTH) ubi = 0;                                       // the next wait leaves exactly the latter in flight
    issue_raw((s + PF) & (RAW_DEPTH - 1));
    if (s + 1 < total) read_frags(s + 1, vn, an);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ps = 0; ps < 9; ++ps)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[ps][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[ps][ct][s2], vc[ps][s2], acc[ps][ct], 0, 0, 0);
    if (++fc == p.nc8) {                                                 // this (observation) frame is complete: A^T M A, epilogue
      const int i = fi++;
      fc = 0;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int oc = (g * TNT + wc * CT + ct) * 16 + 4 * kk;
        const f32x4 bias4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + oc) : (f32x4){0.f, 0.f, 0.f, 0.f};   // (the wait after an epilogue drains anyway)
        f32x4 r0[3], r1[3];
#pragma unroll
        for (int nu = 0; nu < 3; ++nu) {
          r0[nu] = acc[nu][ct] + acc[3 + nu][ct];
          r1[nu] = acc[3 + nu][ct] + acc[6 + nu][ct];
        }
        const f32x4 y[4] = {r0[0] + r0[1], r0[1] + r0[2], r1[0] + r1[1], r1[1] + r1[2]};     // (u, v) = (0,0) (0,1) (1,0) (1,1)
#pragma unroll
        for (int ps = 0; ps < 9; ++ps) acc[ps][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int uv = 0; uv < 4; ++uv) {
          const int gy = ty0 + 2 * wr + (uv >> 1), gx = tx0 + 2 * j + (uv & 1);
          const bool in = gy < p.h && gx < p.w;
          f32x4 o = y[uv] + bias4;
          const long ot = ((long)(f * p.kobs + i) * p.h + gy) * p.w + gx;
          if (p.mask_src || p.accumulate) {                              // backward-data epilogue
            if (in) {
              f32x4* op = reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc);
              if (p.accumulate) o += *op;
              if (p.mask_src) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask_src + ot * p.ld_mask + oc);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= (mk[e] > 0.f) ? 1.f : p.alpha;
              }
              *op = o;
            }
            continue;
          }
          if (p.act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : p.alpha * o[e];
          }
          if (MEAN) mean[ct][uv] += o;
          if (in) {
            if (p.out) *reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc) = o;
            if (MEAN && p.mean_out && i == p.kobs - 1) {
              const long mt = ((long)f * p.h + gy) * p.w + gx;
              *reinterpret_cast<f32x4*>(p.mean_out + mt * p.ldm + oc) = mean[ct][uv] * (1.f / (float)p.kobs);
            }
          }
        }
      }
      drain = true;                                                      // stores (and mask loads) sit on the VM counter behind the DMA pieces
    }
    if (s + 1 < total) transform(vn);
  };
  for (int s = 0; s < total; s += 2) {
    stage(s, v0, a0, v1, a1);
    if (s + 1 < total) stage(s + 1, v1, a1, v0, a0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the clamped requests past the last stage
}

template <bool TR, int WNW>
int launch_wino2(const Wino2P& p, hipStream_t s) {
  const long tiles = (long)p.frames * p.tiles_y * p.tiles_x;
  const dim3 grid((unsigned)tiles, (unsigned)(p.cout / (32 * WNW)));
  const bool mean = !TR && (p.kobs > 1 || p.mean_out);
  if (mean) hipLaunchKernelGGL((conv_wino2_kernel<TR, WNW, !TR>), grid, dim3(256 * WNW), 0, s, p);
  else hipLaunchKernelGGL((conv_wino2_kernel<TR, WNW, false>), grid, dim3(256 * WNW), 0, s, p);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

}  // namespace

// Called by conv_wino.hip's entry points (same argument checks, same packed weights).
int nlt_wino2_run(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w, const float* packed,
                  const float* bias, int cout, int tn, float* out, int ldo, float* mean_out, int ldm, int act, float alpha,
                  const float* mask_src, int ld_mask, int accumulate, hipStream_t s) {
  Wino2P p;
  p.zeros = nlt_zero_page();
  if (!p.zeros) return NLT_ERR_LAUNCH;
  p.src = src; p.packed = packed; p.bias = bias; p.out = out; p.mean_out = mean_out;
  p.ld = ld; p.cin = cin; p.frames = frames; p.kobs = kobs; p.h = h; p.w = w;
  p.cout = cout; p.ldo = ldo; p.ldm = ldm; p.nc8 = cin / 8; p.act = act; p.alpha = alpha;
  p.mask_src = mask_src; p.ld_mask = ld_mask; p.accumulate = accumulate;
  p.tiles_y = (h + 2 * BY - 1) / (2 * BY); p.tiles_x = (w + 2 * BX - 1) / (2 * BX);
  if (mode == NLT_CONV_K2S1) return tn == 64 ? launch_wino2<false, 2>(p, s) : launch_wino2<false, 1>(p, s);
  return tn == 64 ? launch_wino2<true, 2>(p, s) : launch_wino2<true, 1>(p, s);
}
