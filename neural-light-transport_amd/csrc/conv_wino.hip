// Winograd F(2x2, 2x2) form of the stride-1 k2 convs (Conv2D k2s1 'same' and its transpose, nlt/networks/elements.py:26-39) on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32).  A 2 x 2 block of outputs of a 2 x 2-tap conv needs 16 multiplies per (input
// channel, output channel) pair as a direct sum and 9 in the minimal-filtering form
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A,     d = the 3 x 3 input window of the block, g = the 2 x 2 taps,
//     B^T = [1 -1 0; 0 1 0; 0 -1 1],  G = [1 0; 1 1; 0 1],  A^T = [1 1 0; 0 1 1]
//
// so the matrix pipe -- the bound of these launches (the MFMA-bound middle of the network) -- does 9/16 of
// the work: nine GEMMs over the input channels (K = Cin instead of 4 Cin), one per position (xi, nu) of the transformed window.
// Only additions are added (12 per window and channel on the input side, 10 per block and output channel on the output side,
// the weight side G g G^T once at pack time); every product is still an exact fp32 MFMA product accumulated in fp32.
//
// One 256-thread workgroup owns 4 x 16 blocks = 8 x 32 output texels x TN output channels.  The K loop runs over 8-channel slabs:
//   V  [position 9][block row 4][channel quad 2][block 16] 16-byte slots: the transformed windows, made by waves 0-1 (one
//      (block, quad) item per thread: nine 16-byte loads, 12 subtractions, nine conflict-free ds_write_b128);
//   U  [position 9][column tile TNT][channel quad 2][output channel 16] slots: the pre-transformed weights (waves 2-3 copy them)
// into a double-buffered LDS stage (loads in flight under the current stage's MFMAs, one barrier per stage).  Weights = MFMA A
// operand, blocks = B operand: lane (kk, j) reads channels (2 kk, 2 kk + 1) of block / output channel j with one ds_read_b64
// (the two 32-lane halves of the instruction each cover 256 contiguous bytes: conflict-free) and feeds two K steps; it ends up
// with 4 consecutive output channels of block j in each of the nine accumulators, so the output transform is lane-local and a
// texel's channels leave as one 16-byte NHWC store.
//
// Transposed form (Conv2DTranspose k2s1 'same' = backward-data of the stride-1 convs): y[i,j] = sum x[i-a,j-b] W[a,b] is the
// same correlation with the taps flipped and the window starting one texel up / left (zero above / left of the image); the flip
// and the (kh,kw,Cout,Cin) indexing are done at pack time.
#include "nlt_common.h"
#include "pack_common.h"

int nlt_wino2_run(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w, const float* packed,
                  const float* bias, int cout, int tn, float* out, int ldo, float* mean_out, int ldm, int act, float alpha,
                  const float* mask_src, int ld_mask, int accumulate, hipStream_t s);      // conv_wino2.hip

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BY = 4, BX = 16;               // blocks per workgroup: 4 rows x 16 columns = 8 x 32 output texels
constexpr int V_SLOTS = 9 * BY * 2 * BX;     // 16-byte slots per stage (8 channels)

struct WinoP {
  const float* src; const float* packed; const float* bias;
  float* out; float* mean_out;
  int ld, cin, frames, kobs, h, w;
  int cout, ldo, ldm;
  int tiles_y, tiles_x, nc8;                 // nc8 = cin / 8
  int act; float alpha;
  const float* mask_src; int ld_mask; int accumulate;      // backward-data epilogue (nlt_conv_wino_backward_data)
};

// One ds_read_b64, never fused with a neighbour into ds_read2*_b64 (16-lane groups over 32 banks: the 16 blocks of a row, 16
// bytes apart, collide two by two -- conv_wino2.hip has the measurement).
__device__ __forceinline__ f32x2 lds_b64(const char* p) {
  typedef const volatile __attribute__((address_space(3))) f32x2 lds_f32x2;
  return *(lds_f32x2*)(__attribute__((address_space(3))) void*)p;
}

__device__ __forceinline__ int xcd_tile_w(int b, int nblocks) {
  return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3);
}

__global__ void pack_wino_kernel(const float* __restrict__ wk, int cin, int cout, int tnt, int full, int lo, int transposed, long total,
                                 float* __restrict__ wp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  wp[idx] = nlt_wino_fragment(wk, idx, cin, cout, tnt, full, lo, transposed != 0);
}

// ABL (measurement only, NLT_WINO_ABL): 1 = the window loads of stage 0 are reused for every stage, 2 = likewise the weight
// loads, 3 = both, 4 = no MFMAs (operands kept live), 5 = no loads and no LDS stores after stage 0 -- wrong results, right timing.
template <bool TR, int TNT, bool MEAN, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(WinoP p) {
  constexpr int RT = 2, CT = TNT / 2;                                  // waves 2 (block rows) x 2 (column tiles)
  constexpr int U_SLOTS = 9 * TNT * 2 * 16;
  constexpr int STAGE = V_SLOTS + U_SLOTS;
  constexpr int NU = (U_SLOTS + 127) / 128;                            // U copy passes of the 128 copying threads
  __shared__ f32x4 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int wn = wave & 1, wm = wave >> 1;
  int tile = xcd_tile_w(blockIdx.x, gridDim.x);
  const int tx0 = (tile % p.tiles_x) * (2 * BX); tile /= p.tiles_x;
  const int ty0 = (tile % p.tiles_y) * (2 * BY);
  const int f = tile / p.tiles_y;
  const int g = blockIdx.y;
  const int total_stages = p.nc8 * p.kobs;
  const long in_frame = (long)p.h * p.w;
  const bool vrole = wave < 2;                                         // waves 0-1: transformed windows; waves 2-3: weights

  // this thread's window (waves 0-1): block (brow, bj), channel quad bq
  const int bj = tid & 15, bq = (tid >> 4) & 1, brow = (tid >> 5) & 3;
  const int wy0 = ty0 + 2 * brow - (TR ? 1 : 0), wx0 = tx0 + 2 * bj - (TR ? 1 : 0);
  unsigned okmask = 0;                                                 // bit r * 3 + s: window texel (r, s) lies inside the image
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (wy0 + r >= 0 && wy0 + r < p.h && wx0 + s >= 0 && wx0 + s < p.w) okmask |= 1u << (r * 3 + s);
  const long wtex = (long)wy0 * p.w + wx0;
  const int v_slot = (brow * 2 + bq) * BX + bj;                        // + position * (BY * 2 * BX)
  const int ut = tid - 128;                                            // copy thread index of waves 2-3

  f32x4 rg[9];                                                         // stage in flight: the window (waves 0-1) / 9 weight slots (waves 2-3)
  auto load_stage = [&](int q) {
    const int i = q / p.nc8, c8 = q - i * p.nc8;
    if (ABL && q > 0 && ((vrole && (ABL == 1 || ABL == 3 || ABL == 5)) || (!vrole && (ABL == 2 || ABL == 3 || ABL == 5)))) return;
    if (vrole) {
      const float* sp = p.src + (long)(f * p.kobs + i) * in_frame * p.ld + c8 * 8 + bq * 4;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const bool ok = (okmask >> (r * 3 + s)) & 1u;
          const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (ok ? wtex + (long)r * p.w + s : 0) * p.ld);
          rg[r * 3 + s] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    } else {
      const f32x4* up = reinterpret_cast<const f32x4*>(p.packed) + ((long)g * p.nc8 + c8) * U_SLOTS;
#pragma unroll
      for (int n = 0; n < NU; ++n) rg[n] = up[(U_SLOTS % 128 == 0 || ut + 128 * n < U_SLOTS) ? ut + 128 * n : ut];
    }
  };
  auto store_stage = [&](int buf) {
    f32x4* base = lds + buf * STAGE;
    if (vrole) {
      // B^T d B: rows first (e0 = d0 - d1, e1 = d1, e2 = d2 - d1), then columns
      f32x4 e[9];
#pragma unroll
      for (int s = 0; s < 3; ++s) { e[s] = rg[s] - rg[3 + s]; e[3 + s] = rg[3 + s]; e[6 + s] = rg[6 + s] - rg[3 + s]; }
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        base[(x * 3 + 0) * (BY * 2 * BX) + v_slot] = e[x * 3] - e[x * 3 + 1];
        base[(x * 3 + 1) * (BY * 2 * BX) + v_slot] = e[x * 3 + 1];
        base[(x * 3 + 2) * (BY * 2 * BX) + v_slot] = e[x * 3 + 2] - e[x * 3 + 1];
      }
    } else {
#pragma unroll
      for (int n = 0; n < NU; ++n)
        if (U_SLOTS % 128 == 0 || ut + 128 * n < U_SLOTS) base[V_SLOTS + ut + 128 * n] = rg[n];
    }
  };

  f32x4 acc[9][RT][CT], mean[MEAN ? RT : 1][MEAN ? CT : 1][4];
#pragma unroll
  for (int ps = 0; ps < 9; ++ps)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ps][rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (MEAN) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int u = 0; u < 4; ++u) mean[rt][ct][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // byte offsets of this lane's fragment reads inside a stage: slot (.., quad kk >> 1, j), floats (kk & 1) * 2 ..+1
  const int frag = ((kk >> 1) * 16 + j) * 16 + (kk & 1) * 8;

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int q = 0; q < total_stages; ++q) {
    if (q + 1 < total_stages) load_stage(q + 1);
    const char* V = reinterpret_cast<const char*>(lds + (q & 1) * STAGE);
    const char* U = V + V_SLOTS * 16;
#pragma unroll
    for (int ps = 0; ps < 9; ++ps) {
      f32x2 bf[RT], af[CT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        bf[rt] = lds_b64(V + ((ps * BY + wm * RT + rt) * 2 * BX) * 16 + frag);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        af[ct] = lds_b64(U + ((ps * TNT + wn * CT + ct) * 2 * 16) * 16 + frag);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            if (ABL == 4) { asm volatile("" :: "v"(af[ct][s]), "v"(bf[rt][s])); acc[ps][rt][ct][0] += 0.f; }
            else acc[ps][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ct][s], bf[rt][s], acc[ps][rt][ct], 0, 0, 0);
    }
    if ((q + 1) % p.nc8 == 0) {                                        // this (observation) frame is complete: A^T M A, epilogue
      const int i = q / p.nc8;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int oc = (g * TNT + wn * CT + ct) * 16 + 4 * kk;
        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + oc) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          f32x4 r0[3], r1[3];
#pragma unroll
          for (int nu = 0; nu < 3; ++nu) {
            r0[nu] = acc[nu][rt][ct] + acc[3 + nu][rt][ct];
            r1[nu] = acc[3 + nu][rt][ct] + acc[6 + nu][rt][ct];
          }
          f32x4 y[4] = {r0[0] + r0[1], r0[1] + r0[2], r1[0] + r1[1], r1[1] + r1[2]};     // (u, v) = (0,0) (0,1) (1,0) (1,1)
#pragma unroll
          for (int ps = 0; ps < 9; ++ps) acc[ps][rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int uv = 0; uv < 4; ++uv) {
            const int gy = ty0 + 2 * (wm * RT + rt) + (uv >> 1), gx = tx0 + 2 * j + (uv & 1);
            const bool in = gy < p.h && gx < p.w;
            f32x4 v = y[uv] + bv;
            const long ot = ((long)(f * p.kobs + i) * p.h + gy) * p.w + gx;
            if (p.mask_src || p.accumulate) {                          // backward-data epilogue
              if (in) {
                f32x4* o = reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc);
                if (p.accumulate) v += *o;
                if (p.mask_src) {
                  const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask_src + ot * p.ld_mask + oc);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] *= (mk[e] > 0.f) ? 1.f : p.alpha;
                }
                *o = v;
              }
              continue;
            }
            if (p.act) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : p.alpha * v[e];
            }
            if (MEAN) mean[rt][ct][uv] += v;
            if (in) {
              if (p.out) *reinterpret_cast<f32x4*>(p.out + ot * p.ldo + oc) = v;
              if (MEAN && p.mean_out && i == p.kobs - 1) {
                const long mt = ((long)f * p.h + gy) * p.w + gx;
                *reinterpret_cast<f32x4*>(p.mean_out + mt * p.ldm + oc) = mean[rt][ct][uv] * (1.f / (float)p.kobs);
              }
            }
          }
        }
      }
    }
    if (q + 1 < total_stages && !(ABL == 5 && q > 0)) store_stage((q + 1) & 1);
    __syncthreads();
  }
}

template <bool TR, int TNT>
int launch_wino(const WinoP& p, hipStream_t s) {
  const long tiles = (long)p.frames * p.tiles_y * p.tiles_x;
  const dim3 grid((unsigned)tiles, (unsigned)(p.cout / (16 * TNT)));
  const bool mean = !TR && TNT == 2 && (p.kobs > 1 || p.mean_out);
  static const int abl = [] { const char* e = getenv("NLT_WINO_ABL"); return e ? atoi(e) : 0; }();
  if (abl && !TR && TNT == 4) {
    if (abl == 1) hipLaunchKernelGGL((conv_wino_kernel<TR, TNT, false, (!TR && TNT == 4) ? 1 : 0>), grid, dim3(256), 0, s, p);
    else if (abl == 2) hipLaunchKernelGGL((conv_wino_kernel<TR, TNT, false, (!TR && TNT == 4) ? 2 : 0>), grid, dim3(256), 0, s, p);
    else if (abl == 3) hipLaunchKernelGGL((conv_wino_kernel<TR, TNT, false, (!TR && TNT == 4) ? 3 : 0>), grid, dim3(256), 0, s, p);
    else if (abl == 4) hipLaunchKernelGGL((conv_wino_kernel<TR, TNT, false, (!TR && TNT == 4) ? 4 : 0>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_wino_kernel<TR, TNT, false, (!TR && TNT == 4) ? 5 : 0>), grid, dim3(256), 0, s, p);
    NLT_CHECK_LAUNCH();
    return NLT_OK;
  }
  if (mean) hipLaunchKernelGGL((conv_wino_kernel<TR, TNT, !TR && TNT == 2>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv_wino_kernel<TR, TNT, false>), grid, dim3(256), 0, s, p);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

// NLT_WINO_V1=1: the first-generation (register-staged) kernel below for every launch (A/B runs); default = conv_wino2.hip
bool wino_v1() {
  static const bool v = [] { const char* e = getenv("NLT_WINO_V1"); return e && e[0] == '1'; }();
  return v;
}

int wino_run(int mode, const WinoP& p, int tn, hipStream_t s) {
  if (!wino_v1())
    return nlt_wino2_run(mode, p.src, p.ld, p.cin, p.frames, p.kobs, p.h, p.w, p.packed, p.bias, p.cout, tn, p.out, p.ldo, p.mean_out,
                         p.ldm, p.act, p.alpha, p.mask_src, p.ld_mask, p.accumulate, s);
  if (mode == NLT_CONV_K2S1) return tn == 64 ? launch_wino<false, 4>(p, s) : launch_wino<false, 2>(p, s);
  return tn == 64 ? launch_wino<true, 4>(p, s) : launch_wino<true, 2>(p, s);
}

}  // namespace

extern "C" long nlt_conv_wino_packed_floats(int mode, int cin, int cout, int tn) {
  if ((mode != NLT_CONV_K2S1 && mode != NLT_DECONV_K2S1) || cin <= 0 || cout <= 0) return -1;
  if ((cin & 7) || (tn != 32 && tn != 64) || cout % tn) return -1;
  return (long)9 * cin * cout;
}

static int pack_wino(int mode, const float* w_keras, int cin, int cout, int tn, int full, int lo, float* packed, void* stream) {
  const long total = nlt_conv_wino_packed_floats(mode, cin, cout, tn);
  if (total <= 0) return NLT_ERR_UNSUPPORTED;
  if (!w_keras || !packed || !nlt_aligned16(packed) || lo < 0 || lo + cout > full) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_keras, cin, cout, tn / 16, full, lo, mode == NLT_DECONV_K2S1 ? 1 : 0, total, packed);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_pack_conv_wino_weights(int mode, const float* w_keras, int cin, int cout, int tn, float* packed, void* stream) {
  return pack_wino(mode, w_keras, cin, cout, tn, cout, 0, packed, stream);
}

extern "C" int nlt_pack_conv_wino_weights_adjoint(int adj_mode, const float* w_keras, int cpre, int cout, int tn, int full, int lo,
                                                  float* packed, void* stream) {
  return pack_wino(adj_mode, w_keras, cpre, cout, tn, full, lo, packed, stream);
}

static int wino_check(int mode, const float* src, int ld, int cin, long long rows, const float* packed, int cout, int tn,
                      const float* out, int ldo) {
  if (!src || !packed || cin <= 0 || cout <= 0 || rows <= 0) return NLT_ERR_BAD_ARG;
  if (nlt_conv_wino_packed_floats(mode, cin, cout, tn) <= 0) return NLT_ERR_UNSUPPORTED;
  if (ld < cin || (ld & 3) || (out && (ldo < cout || (ldo & 3)))) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(src) || !nlt_aligned16(packed) || (out && !nlt_aligned16(out))) return NLT_ERR_BAD_ARG;
  if (rows * (long long)(ld > ldo ? ld : ldo) >= (1ll << 31)) return NLT_ERR_UNSUPPORTED;
  return NLT_OK;
}

extern "C" int nlt_conv_wino_forward(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                                     const float* packed, const float* bias, int cout, int tn,
                                     float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream) {
  if (!bias || (!out && !mean_out) || frames <= 0 || kobs <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  const int rc = wino_check(mode, src, ld, cin, (long long)frames * kobs * h * w, packed, cout, tn, out, ldo);
  if (rc != NLT_OK) return rc;
  if (mean_out && (ldm < cout || (ldm & 3) || !nlt_aligned16(mean_out))) return NLT_ERR_BAD_ARG;
  if (!nlt_aligned16(bias)) return NLT_ERR_BAD_ARG;
  if ((kobs > 1 || mean_out) && mode != NLT_CONV_K2S1) return NLT_ERR_UNSUPPORTED;
  if ((kobs > 1 || mean_out) && tn != 32 && wino_v1()) return NLT_ERR_UNSUPPORTED;     // first generation: the running mean fits at 32 channels only
  WinoP p;
  p.src = src; p.packed = packed; p.bias = bias; p.out = out; p.mean_out = mean_out;
  p.ld = ld; p.cin = cin; p.frames = frames; p.kobs = kobs; p.h = h; p.w = w;
  p.cout = cout; p.ldo = ldo; p.ldm = ldm; p.nc8 = cin / 8; p.act = act; p.alpha = alpha;
  p.mask_src = nullptr; p.ld_mask = 0; p.accumulate = 0;
  p.tiles_y = (h + 2 * BY - 1) / (2 * BY); p.tiles_x = (w + 2 * BX - 1) / (2 * BX);
  return wino_run(mode, p, tn, static_cast<hipStream_t>(stream));
}

extern "C" int nlt_conv_wino_backward_data(int adj_mode, const float* dpre, int ldp, int cpre, int n, int h, int w,
                                           const float* packed, int cout, int tn, float* out, int ldo,
                                           const float* mask_src, int ldm, float mask_alpha, int accumulate, void* stream) {
  if (!out || n <= 0 || h <= 0 || w <= 0) return NLT_ERR_BAD_ARG;
  const int rc = wino_check(adj_mode, dpre, ldp, cpre, (long long)n * h * w, packed, cout, tn, out, ldo);
  if (rc != NLT_OK) return rc;
  if (mask_src && (ldm < cout || (ldm & 3) || !nlt_aligned16(mask_src))) return NLT_ERR_BAD_ARG;
  WinoP p;
  p.src = dpre; p.packed = packed; p.bias = nullptr; p.out = out; p.mean_out = nullptr;
  p.ld = ldp; p.cin = cpre; p.frames = n; p.kobs = 1; p.h = h; p.w = w;
  p.cout = cout; p.ldo = ldo; p.ldm = 0; p.nc8 = cpre / 8; p.act = 0; p.alpha = mask_alpha;
  p.mask_src = mask_src; p.ld_mask = ldm; p.accumulate = accumulate;
  p.tiles_y = (h + 2 * BY - 1) / (2 * BY); p.tiles_x = (w + 2 * BX - 1) / (2 * BX);
  return wino_run(adj_mode, p, tn, static_cast<hipStream_t>(stream));
}
