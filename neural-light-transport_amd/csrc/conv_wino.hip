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
// (The first-generation, register-staged kernel -- every wave fetched its raw tile into registers and the weights once per
// stage behind a workgroup barrier -- lived here until r06; conv_wino2.hip measured 15-25 % faster on every shape that chose a
// Winograd launch and it lost its A/B switch NLT_WINO_V1: profiles/README.md r04.)
int wino_run(int mode, const WinoP& p, int tn, hipStream_t s) {
  return nlt_wino2_run(mode, p.src, p.ld, p.cin, p.frames, p.kobs, p.h, p.w, p.packed, p.bias, p.cout, tn, p.out, p.ldo, p.mean_out,
                       p.ldm, p.act, p.alpha, p.mask_src, p.ld_mask, p.accumulate, s);
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
