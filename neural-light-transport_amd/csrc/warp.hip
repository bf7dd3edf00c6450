// UV -> camera bilinear gather (tfa.image.resampler semantics) for pred/base/fg in one pass,
// and the TF2 half-pixel bilinear resize.
//
// Gather kernel: one LANE per (camera pixel, colour channel) -- a wave covers 21 consecutive pixels x rgb = 63 lanes.
// A tap instruction then reads 12 contiguous bytes per pixel (neighbouring pixels of a chart hit neighbouring texels:
// ~0.5 KB per instruction) instead of one channel of 64 pixels spread over 1.5 KB, the 63 results of a wave are one
// contiguous store, and every lane still adds its four taps in the resampler's order.
#include "nlt_common.h"
#include <hip/hip_fp16.h>

namespace {

constexpr int WARP_PX = 21;      // camera pixels per wave pass (3 lanes each; lane 63 idles)

// Arithmetic is kept un-contracted (no FMA fusion) and in the TFA kernel's order so that the
// fp32 result equals the oracle's NumPy float32 restatement operation for operation.
// STORE = true: base and the uv2cam map are read where the capture lives -- the resident uint8 diffuse store [F,uvh,uvw,3]
// and the fp16 uv2cam store [F,hc,wc,2] (data_gen/util.py:67-70 save_float16_npy), frame ids[f] -- with `_load_data`'s
// conversions in registers (u8_unit; fp16 -> fp32 is exact), so a store-resident batch needs neither the float32 base
// (48 MB written + gathered per 4 frames at 1024^2) nor the float32 copy of the map.  Same arithmetic, same results.
#pragma clang fp contract(off)
template <bool STORE>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ pred, const void* __restrict__ base_v,
                                                   const void* __restrict__ warp_v, const int* __restrict__ ids,
                                                   int uvh, int uvw, int hcwc,
                                                   long total, float* __restrict__ pred_cam,
                                                   float* __restrict__ base_cam, float* __restrict__ fg_cam,
                                                   int* __restrict__ idx_out) {
  const int lane = threadIdx.x & 63;
  const int j = lane / 3, c = lane - 3 * j;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long p = wave * WARP_PX + j;
  if (j >= WARP_PX || p >= total) return;
  const int f = p / hcwc;
  const int fs = STORE ? ids[f] : f;                 // frame of the stores
  float wx, wy;
  if (STORE) {
    const __half2 hv = static_cast<const __half2*>(warp_v)[(long)fs * hcwc + (p - (long)f * hcwc)];
    wx = __low2float(hv); wy = __high2float(hv);
  } else {
    const float* warp = static_cast<const float*>(warp_v);
    wx = warp[p * 2 + 0]; wy = warp[p * 2 + 1];
  }
  const float x = wx * (float)uvw;                   // nlt/models/nlt.py:104-106
  const float y = wy * (float)uvh;
  const bool inside = x > -1.f && y > -1.f && x < (float)uvw && y < (float)uvh;
  const int fx = (int)floorf(x), fy = (int)floorf(y);
  if (idx_out && c == 0) {
    idx_out[p * 4 + 0] = fx; idx_out[p * 4 + 1] = fy; idx_out[p * 4 + 2] = inside ? 1 : 0; idx_out[p * 4 + 3] = 0;
  }
  float op = 0.f, ob = 0.f, og = 0.f;
  if (inside) {
    const int cx = fx + 1, cy = fy + 1;
    const float dx = (float)cx - x, dy = (float)cy - y;
    const float wts[4] = {dx * dy, (1.f - dx) * (1.f - dy), dx * (1.f - dy), (1.f - dx) * dy};
    const int xs[4] = {fx, cx, fx, cx};
    const int ys[4] = {fy, cy, cy, fy};
    float vp[4], vb[4], vg[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {                    // all taps in flight before the first add
      const int xi = xs[t], yi = ys[t];
      const bool ok = xi >= 0 && yi >= 0 && xi <= uvw - 1 && yi <= uvh - 1;
      const bool corner = (xi == 0 && yi == 0);      // set_left_top_corner(., 0): nlt.py:108-110
      const long in_frame = ((long)(ok ? yi : 0)) * uvw + (ok ? xi : 0);
      const long tex = (long)f * uvh * uvw + in_frame;
      vp[t] = (ok && pred) ? pred[tex * 3 + c] : 0.f;
      if (STORE) {
        const unsigned char* bs = static_cast<const unsigned char*>(base_v);
        vb[t] = (ok && !corner && bs && base_cam) ? u8_unit(bs[((long)fs * uvh * uvw + in_frame) * 3 + c]) : 0.f;
      } else {
        const float* base = static_cast<const float*>(base_v);
        vb[t] = (ok && !corner && base && base_cam) ? base[tex * 3 + c] : 0.f;   // (a pred-only launch reads no base)
      }
      vg[t] = (ok && !corner) ? 1.f : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      op = op + wts[t] * vp[t];
      ob = ob + wts[t] * vb[t];
      og = og + wts[t] * vg[t];
    }
  }
  if (pred_cam) pred_cam[p * 3 + c] = op;
  if (base_cam) base_cam[p * 3 + c] = ob;
  if (fg_cam) fg_cam[p * 3 + c] = og;
}

// tfa.image.resampler on a map of c channels (c % 4 == 0): the channel-width stress point of SURVEY.md 8(d) ("1024^2 x
// 64-ch"), same tap rule / order as warp_kernel without the corner mask.  lane = (camera pixel, channel quad): a texel's
// c channels are one contiguous 4c-byte run, so a tap instruction of a wave reads whole texels with 16-byte loads.
__global__ __launch_bounds__(256) void resample_c_kernel(const float* __restrict__ data, const float* __restrict__ warp, int c,
                                                         int h, int w, int hcwc, long total_q, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total_q) return;
  const int cq = c >> 2;
  const long p = i / cq;
  const int q = i - p * cq;
  const int f = p / hcwc;
  const float x = warp[p * 2 + 0], y = warp[p * 2 + 1];   // pixel units already (the raw op)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (x > -1.f && y > -1.f && x < (float)w && y < (float)h) {
    const int fx = (int)floorf(x), fy = (int)floorf(y), cx = fx + 1, cy = fy + 1;
    const float dx = (float)cx - x, dy = (float)cy - y;
    const float wts[4] = {dx * dy, (1.f - dx) * (1.f - dy), dx * (1.f - dy), (1.f - dx) * dy};
    const int xs[4] = {fx, cx, fx, cx};
    const int ys[4] = {fy, cy, cy, fy};
    f32x4 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool ok = xs[t] >= 0 && ys[t] >= 0 && xs[t] <= w - 1 && ys[t] <= h - 1;
      const long tex = ((long)f * h + (ok ? ys[t] : 0)) * w + (ok ? xs[t] : 0);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      v[t] = ok ? *reinterpret_cast<const f32x4*>(data + tex * c + 4 * q) : z;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = acc + wts[t] * v[t];
  }
  *reinterpret_cast<f32x4*>(out + p * c + 4 * q) = acc;
}

__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ x, int n, int h, int w, int c,
                                                     int oh, int ow, long total, float* __restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int ox = p % ow;
  const int oy = (p / ow) % oh;
  const int f = p / ((long)ow * oh);
  const float sy = (float)h / (float)oh, sx = (float)w / (float)ow;
  const float srcy = ((float)oy + 0.5f) * sy - 0.5f;
  const float srcx = ((float)ox + 0.5f) * sx - 0.5f;
  const float fly = floorf(srcy), flx = floorf(srcx);
  const int ylo = max((int)fly, 0), yhi = min((int)ceilf(srcy), h - 1);
  const int xlo = max((int)flx, 0), xhi = min((int)ceilf(srcx), w - 1);
  const float ly = srcy - fly, lx = srcx - flx;
  const float* tl = x + (((long)f * h + ylo) * w + xlo) * c;
  const float* tr = x + (((long)f * h + ylo) * w + xhi) * c;
  const float* bl = x + (((long)f * h + yhi) * w + xlo) * c;
  const float* br = x + (((long)f * h + yhi) * w + xhi) * c;
  for (int ch = 0; ch < c; ++ch) {
    const float top = tl[ch] + (tr[ch] - tl[ch]) * lx;
    const float bot = bl[ch] + (br[ch] - bl[ch]) * lx;
    out[p * c + ch] = top + (bot - top) * ly;
  }
}

}  // namespace

extern "C" int nlt_warp_forward(const float* pred, const float* base, const float* warp,
                                int n, int uvh, int uvw, int hc, int wc,
                                float* pred_cam, float* base_cam, float* fg_cam, int* idx_out, void* stream) {
  if (!warp || n <= 0 || uvh <= 0 || uvw <= 0 || hc <= 0 || wc <= 0) return NLT_ERR_BAD_ARG;
  if (pred_cam && !pred) return NLT_ERR_BAD_ARG;
  if (!pred_cam && !base_cam && !fg_cam && !idx_out) return NLT_ERR_BAD_ARG;
  const long total = (long)n * hc * wc;
  hipLaunchKernelGGL(warp_kernel<false>, dim3((unsigned)((total + 4 * WARP_PX - 1) / (4 * WARP_PX))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pred, base, warp, nullptr, uvh, uvw, hc * wc, total,
                     pred_cam, base_cam, fg_cam, idx_out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_warp_forward_store(const float* pred, const unsigned char* diffuse_store, const unsigned short* uv2cam_store,
                                      const int* ids, int n, int uvh, int uvw, int hc, int wc,
                                      float* pred_cam, float* base_cam, float* fg_cam, int* idx_out, void* stream) {
  if (!uv2cam_store || !ids || n <= 0 || uvh <= 0 || uvw <= 0 || hc <= 0 || wc <= 0) return NLT_ERR_BAD_ARG;
  if (pred_cam && !pred) return NLT_ERR_BAD_ARG;
  if (!pred_cam && !base_cam && !fg_cam && !idx_out) return NLT_ERR_BAD_ARG;
  const long total = (long)n * hc * wc;
  hipLaunchKernelGGL(warp_kernel<true>, dim3((unsigned)((total + 4 * WARP_PX - 1) / (4 * WARP_PX))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pred, diffuse_store, uv2cam_store, ids, uvh, uvw, hc * wc, total,
                     pred_cam, base_cam, fg_cam, idx_out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_resample_forward(const float* data, const float* warp_px, int n, int h, int w, int c, int hc, int wc,
                                    float* out, void* stream) {
  if (!data || !warp_px || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || hc <= 0 || wc <= 0) return NLT_ERR_BAD_ARG;
  if (c % 4 || !nlt_aligned16(data) || !nlt_aligned16(out)) return NLT_ERR_UNSUPPORTED;
  const long total_q = (long)n * hc * wc * (c / 4);
  hipLaunchKernelGGL(resample_c_kernel, dim3((unsigned)((total_q + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     data, warp_px, c, h, w, hc * wc, total_q, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_resize_bilinear_forward(const float* x, int n, int h, int w, int c, int oh, int ow,
                                           float* out, void* stream) {
  if (!x || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || oh <= 0 || ow <= 0) return NLT_ERR_BAD_ARG;
  const long total = (long)n * oh * ow;
  hipLaunchKernelGGL(resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, n, h, w, c, oh, ow, total, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
