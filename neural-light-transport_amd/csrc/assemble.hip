// Texel-buffer assembly (SURVEY.md 8a rows a-B1..a-B6): light/view cosine maps, diffuse base,
// camera<->UV bilinear remap, the integer UV-index (occupancy) map with nearest-sample fill, k-NN
// observation indices, and the uint8 -> float32 batch assembly of the data loader.
//
// Integer / byte work throughout: results are bit-exact against oracle/buffers.py.  The float64
// stages follow the reference's NumPy operation order one IEEE operation at a time; this file is
// compiled with -ffp-contract=off (see Makefile) so no multiply-add is fused behind our back.
#include "nlt_common.h"
#include <hip/hip_fp16.h>

namespace {

typedef unsigned char u8;
typedef unsigned long long u64;

inline unsigned blocks_for(long total, int per_block = 256) { return (unsigned)((total + per_block - 1) / per_block); }

// ---------------------------------------------------------------------------------------
// a-B2  data_gen/render.py:209-228 (view), :231-276 (light), quantisation :164,170
// thread = camera pixel
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double dot3(double ax, double ay, double az, double bx, double by, double bz) {
  return (ax * bx + ay * by) + az * bz;
}

__global__ __launch_bounds__(256) void cosine_kernel(const double* __restrict__ locs, const double* __restrict__ normals,
                                                     const u8* __restrict__ valid, const u8* __restrict__ occluded,
                                                     double sx, double sy, double sz, long pixels,
                                                     double* __restrict__ cos_out, u8* __restrict__ u8_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixels) return;
  double c = 0.0;
  const bool keep = valid[i] != 0 && !(occluded && occluded[i] != 0);
  if (keep) {
    double dx = sx - locs[3 * i], dy = sy - locs[3 * i + 1], dz = sz - locs[3 * i + 2];
    double nx = normals[3 * i], ny = normals[3 * i + 1], nz = normals[3 * i + 2];
    double dl = sqrt(dot3(dx, dy, dz, dx, dy, dz));
    double nl = sqrt(dot3(nx, ny, nz, nx, ny, nz));
    if (!(dl > 0.0)) dl = 1.0;            // mathutils: the zero vector normalises to itself
    if (!(nl > 0.0)) nl = 1.0;
    dx = dx / dl; dy = dy / dl; dz = dz / dl;
    nx = nx / nl; ny = ny / nl; nz = nz / nl;
    c = dot3(dx, dy, dz, nx, ny, nz);
  }
  if (cos_out) cos_out[i] = c;
  if (u8_out) {
    double q = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);   // np.clip(x, 0, 1)
    u8_out[i] = (u8)(int)(q * 255.0);                 // denormalize_float: truncation
  }
}

// ---------------------------------------------------------------------------------------
// a-B3  data_gen/postproc.py:53-76
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void albedo_sum_kernel(const u8* __restrict__ frames, int nframes, long elems,
                                                         double* __restrict__ sum, u64* __restrict__ maxbits) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  if (e < elems) {
    for (int f = 0; f < nframes; ++f) s = s + (double)frames[(long)f * elems + e] / 255.0;   // frame order, as rgb_sum += rgb
    sum[e] = s;
  }
  // s >= 0: the IEEE bit pattern orders like the value -> integer max; wave shuffle, then one atomic per wave
  u64 b = (u64)__double_as_longlong(s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const u64 o = (u64)__shfl_xor((long long)b, off, 64);
    b = o > b ? o : b;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(maxbits, b);
}

__global__ __launch_bounds__(256) void albedo_div_kernel(double* __restrict__ sum, long elems, const u64* __restrict__ maxbits) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  const double m = __longlong_as_double((long long)*maxbits);
  sum[e] = sum[e] / m;
}

__global__ __launch_bounds__(256) void diffuse_kernel(const double* __restrict__ albedo, const u8* __restrict__ lvis,
                                                      long texels, long total, u8* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over frames*texels*3
  if (i >= total) return;
  const long per_frame = texels * 3;
  const long e = i % per_frame;
  const long t = i / 3;                                          // frame*texels + texel
  const double lv = (double)lvis[t] / 255.0;
  double d = albedo[e] * lv;
  d = d < 0.0 ? 0.0 : (d > 1.0 ? 1.0 : d);
  out[i] = (u8)(int)(d * 255.0);
}

// ---------------------------------------------------------------------------------------
// a-B4  data_gen/util.py:45-58 = cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0), src(0,0) forced 0
// thread = output texel
// ---------------------------------------------------------------------------------------
template <int MAPT> __device__ __forceinline__ float map_coord(const void* m, long idx, int size);
template <> __device__ __forceinline__ float map_coord<0>(const void* m, long idx, int size) {      // float64 map
  return (float)(static_cast<const double*>(m)[idx] * (double)size);
}
template <> __device__ __forceinline__ float map_coord<1>(const void* m, long idx, int size) {      // float32 map
  return static_cast<const float*>(m)[idx] * (float)size;
}
template <> __device__ __forceinline__ float map_coord<2>(const void* m, long idx, int size) {      // float16 map (uv2cam.npy)
  const __half v = static_cast<const __half*>(m)[idx];
  return __half2float(__float2half_rn(__half2float(v) * (float)size));   // NumPy multiplies float16 in float16
}

struct RemapCoord { int ix, iy, fx, fy; };

template <int MAPT>
__device__ __forceinline__ RemapCoord remap_coord(const void* mapping, long o, int ldm, int h, int w) {
  const float mx = map_coord<MAPT>(mapping, o * ldm, w);
  const float my = map_coord<MAPT>(mapping, o * ldm + 1, h);
  const int sx = __float2int_rn(mx * 32.f);                    // cvRound(x * INTER_TAB_SIZE): half to even
  const int sy = __float2int_rn(my * 32.f);
  RemapCoord r;
  r.ix = min(max(sx >> 5, -32768), 32767);                     // saturate_cast<short>
  r.iy = min(max(sy >> 5, -32768), 32767);
  r.fx = sx & 31; r.fy = sy & 31;
  return r;
}

template <int MAPT>
__global__ __launch_bounds__(256) void remap_u8_kernel(const u8* __restrict__ src, int h, int w, int c,
                                                       const void* __restrict__ mapping, int ldm, long out_px,
                                                       int force_kbg, u8* __restrict__ out) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= out_px) return;
  const RemapCoord r = remap_coord<MAPT>(mapping, o, ldm, h, w);
  const int wt[4] = {(32 - r.fy) * (32 - r.fx) * 32, (32 - r.fy) * r.fx * 32, r.fy * (32 - r.fx) * 32, r.fy * r.fx * 32};
  long tap[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int x = r.ix + (t & 1), y = r.iy + (t >> 1);
    const bool ok = x >= 0 && x < w && y >= 0 && y < h && !(force_kbg && x == 0 && y == 0);
    tap[t] = ok ? ((long)y * w + x) * c : -1;
  }
  for (int ch = 0; ch < c; ++ch) {
    int acc = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc += (tap[t] >= 0 ? (int)src[tap[t] + ch] : 0) * wt[t];
    out[o * c + ch] = (u8)((acc + (1 << 14)) >> 15);
  }
}

template <int MAPT>
__global__ __launch_bounds__(256) void remap_f32_kernel(const float* __restrict__ src, int h, int w, int c,
                                                        const void* __restrict__ mapping, int ldm, long out_px,
                                                        int force_kbg, float* __restrict__ out) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= out_px) return;
  const RemapCoord r = remap_coord<MAPT>(mapping, o, ldm, h, w);
  const float ax = (float)r.fx * (1.f / 32.f), ay = (float)r.fy * (1.f / 32.f);
  const float wt[4] = {(1.f - ay) * (1.f - ax), (1.f - ay) * ax, ay * (1.f - ax), ay * ax};
  long tap[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int x = r.ix + (t & 1), y = r.iy + (t >> 1);
    const bool ok = x >= 0 && x < w && y >= 0 && y < h && !(force_kbg && x == 0 && y == 0);
    tap[t] = ok ? ((long)y * w + x) * c : -1;
  }
  for (int ch = 0; ch < c; ++ch) {
    float acc = (tap[0] >= 0 ? src[tap[0] + ch] : 0.f) * wt[0];
#pragma unroll
    for (int t = 1; t < 4; ++t) acc = acc + (tap[t] >= 0 ? src[tap[t] + ch] : 0.f) * wt[t];
    out[o * c + ch] = acc;
  }
}

// ---------------------------------------------------------------------------------------
// a-B5  xiuminglib/img.py:289-431 grid_query_unstruct (griddata nearest + L1 trust mask)
//   scatter: integer occupancy indices + per-texel sample lists (lock-free linked lists)
//   query:   trust test on the L1 diamond, then exact nearest sample in a bounded window
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uvmap_scatter_kernel(const double* __restrict__ uvs, long samples, int h, int w,
                                                            u8* __restrict__ occ, int* __restrict__ head, int* __restrict__ next) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= samples) return;
  const double u = uvs[2 * p], v = uvs[2 * p + 1];
  const double rf = (1.0 - v) * (double)(h - 1);               // img.py:389
  const double cf = u * (double)(w - 1);                       // img.py:390
  if (!(rf == rf) || !(cf == cf)) { next[p] = -1; return; }    // NaN coordinates never match anything
  const double lim = 2147483000.0;
  const int ri = (int)(rf > lim ? lim : (rf < -lim ? -lim : rf));     // astype(int): truncation toward zero
  const int ci = (int)(cf > lim ? lim : (cf < -lim ? -lim : cf));
  if (ri >= 0 && ri < h && ci >= 0 && ci < w) occ[(long)ri * w + ci] = 1;   // img.py:391-393
  const int br = min(max(ri, 0), h - 1), bc = min(max(ci, 0), w - 1);       // list bucket (clamped into the canvas)
  next[p] = atomicExch(&head[(long)br * w + bc], (int)p);
}

__global__ __launch_bounds__(256) void uvmap_query_kernel(const double* __restrict__ uvs, const double* __restrict__ values,
                                                          int m, int h, int w, int max_l1, int ry, int rx,
                                                          double step_u, double step_v, double fill,
                                                          const u8* __restrict__ occ, const int* __restrict__ head,
                                                          const int* __restrict__ next, double* __restrict__ out,
                                                          int* __restrict__ index_out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)h * w) return;
  const int i = t / w, j = t - (long)i * w;
  bool trusted = false;                                         // cv2.distanceTransform(L1) <= max_l1 (img.py:394-395)
  for (int di = -max_l1; di <= max_l1 && !trusted; ++di) {
    const int y = i + di;
    if (y < 0 || y >= h) continue;
    const int rem = max_l1 - (di < 0 ? -di : di);
    const int x0 = max(j - rem, 0), x1 = min(j + rem, w - 1);
    for (int x = x0; x <= x1; ++x)
      if (occ[(long)y * w + x]) { trusted = true; break; }
  }
  int best = -1;
  if (trusted) {
    // np.linspace(0, 1, n): arange * step, last element forced to 1 (img.py:366-373: grid_v = 1 - grid_y)
    const double gu = (j == w - 1) ? 1.0 : (double)j * step_u;
    const double gv = 1.0 - ((i == h - 1) ? 1.0 : (double)i * step_v);
    double bd = 0.0;
    const int y0 = max(i - ry, 0), y1 = min(i + ry, h - 1), x0 = max(j - rx, 0), x1 = min(j + rx, w - 1);
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x)
        for (int p = head[(long)y * w + x]; p >= 0; p = next[p]) {
          const double du = gu - uvs[2 * (long)p], dv = gv - uvs[2 * (long)p + 1];
          const double d2 = du * du + dv * dv;
          if (best < 0 || d2 < bd || (d2 == bd && p < best)) { bd = d2; best = p; }
        }
  }
  for (int ch = 0; ch < m; ++ch) out[t * m + ch] = best >= 0 ? values[(long)best * m + ch] : fill;
  if (index_out) index_out[t] = best;
}

// ---------------------------------------------------------------------------------------
// a-B6  data_gen/get_neighbors.py:52-71, k nearest instead of 1.  One WAVE per reference
// position: lanes stride over the candidates, a 64-lane shuffle reduction picks the
// lexicographic minimum of (distance^2, candidate index) above the previous pick.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_kernel(const double* __restrict__ ref, int np, const double* __restrict__ cand,
                                                  int nq, int k, int* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= np) return;
  const double rx = ref[3 * p], ry = ref[3 * p + 1], rz = ref[3 * p + 2];
  double last_d = -1.0; int last_i = -1;
  for (int r = 0; r < k; ++r) {
    double bd = 0.0; int bi = -1;
    for (int q = lane; q < nq; q += 64) {
      const double dx = rx - cand[3 * q], dy = ry - cand[3 * q + 1], dz = rz - cand[3 * q + 2];
      const double d2 = (dx * dx + dy * dy) + dz * dz;
      if (d2 == 0.0 || !(d2 == d2)) continue;                                  // `dist != 0` (get_neighbors.py:63)
      if (d2 < last_d || (d2 == last_d && q <= last_i)) continue;              // already emitted
      if (bi < 0 || d2 < bd) { bd = d2; bi = q; }                              // q ascending: first minimum wins
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double od = __shfl_xor(bd, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (oi >= 0 && (bi < 0 || od < bd || (od == bd && oi < bi))) { bd = od; bi = oi; }
    }
    if (lane == 0) out[(long)p * k + r] = bi;
    if (bi < 0) {                                            // fewer than k qualifying candidates
      for (int rr = r + 1; rr < k; ++rr) if (lane == 0) out[(long)p * k + rr] = -1;
      break;
    }
    last_d = bd; last_i = bi;
  }
}

// ---------------------------------------------------------------------------------------
// a-B1  nlt/datasets/nlt.py:115-184 on a resident uint8 frame store: gather by frame id and
// uint8 -> float64/255 -> float32.  thread = 4 consecutive bytes -> one 16-byte store.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void u8_gather_kernel(const u8* __restrict__ store, const int* __restrict__ ids,
                                                        long per_frame, long quads_per_frame, long total_quads,
                                                        float* __restrict__ out) {
  __shared__ float lut[256];
  lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0);    // normalize_uint then astype(float32)
  __syncthreads();
  const long stride = (long)gridDim.x * blockDim.x;
  for (long qd = (long)blockIdx.x * blockDim.x + threadIdx.x; qd < total_quads; qd += stride) {
    const long f = qd / quads_per_frame;
    const long e = (qd - f * quads_per_frame) * 4;
    const int id = ids ? ids[f] : (int)f;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (id >= 0) {                                            // -1: missing neighbour -> zeros (nlt.py:152-157)
      const uchar4 b = *reinterpret_cast<const uchar4*>(store + (long)id * per_frame + e);
      v = (f32x4){lut[b.x], lut[b.y], lut[b.z], lut[b.w]};
    }
    *reinterpret_cast<f32x4*>(out + f * per_frame + e) = v;
  }
}

// ---------------------------------------------------------------------------------------
// PSNR on luma (xiuminglib/metric.py:105-151, img.py:600-611): float64 throughout, as the reference's
// `im.astype(float)`.  Pass 1: every workgroup adds its pixels' masked squared luma differences in a fixed order
// (thread-strided partial sums, then a tree) and writes (sum, count) to its slot; pass 2: one workgroup adds the slots in
// order.  Deterministic; differs from NumPy's pairwise sum by float64 rounding only.
// ---------------------------------------------------------------------------------------
constexpr int PSNR_BLOCKS = 256;

__device__ __forceinline__ double lum_of(const float* p, int c) {
  if (c == 1) return (double)p[0];
  return 0.2126 * (double)p[0] + 0.7152 * (double)p[1] + 0.0722 * (double)p[2];
}

__global__ __launch_bounds__(256) void psnr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const u8* __restrict__ mask, long pixels, int c,
                                                           double* __restrict__ part) {
  __shared__ double s_se[256], s_n[256];
  double se = 0.0, cnt = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pixels; i += (long)PSNR_BLOCKS * 256) {
    if (mask && !mask[i]) continue;
    const double d = lum_of(a + i * c, c) - lum_of(b + i * c, c);
    se = se + d * d;
    cnt = cnt + 1.0;
  }
  s_se[threadIdx.x] = se; s_n[threadIdx.x] = cnt;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { s_se[threadIdx.x] += s_se[threadIdx.x + w]; s_n[threadIdx.x] += s_n[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s_se[0]; part[2 * blockIdx.x + 1] = s_n[0]; }
}

__global__ void psnr_final_kernel(const double* __restrict__ part, double* __restrict__ out) {
  double se = 0.0, cnt = 0.0;
  for (int i = 0; i < PSNR_BLOCKS; ++i) { se = se + part[2 * i]; cnt = cnt + part[2 * i + 1]; }
  out[0] = se; out[1] = cnt;
}

int launch_gather(const u8* store, const int* ids, int nframes_out, long per_frame, float* out, hipStream_t s) {
  if (per_frame & 3) return NLT_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(store) & 3u) || !nlt_aligned16(out)) return NLT_ERR_BAD_ARG;
  const long qpf = per_frame >> 2, total = qpf * nframes_out;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(u8_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, s, store, ids, per_frame, qpf, total, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

// cv2.resize(arr, (ow, oh)) with the default INTER_LINEAR on the NORMALISED float64 image, as `_load_data` applies it
// when a capture is stored at another resolution than uvh / (imh, imw) (nlt/datasets/nlt.py:138-146; xm.img.resize,
// xiuminglib/img.py:88-118), followed by the astype(float32) of nlt.py:173-181.  OpenCV's table loop restated: per axis
// fx = float((d + 0.5) * (src / dst) - 0.5), s = floor(fx), fx -= s, clamped at both ends with weight 0; tap weights stay
// float32, the horizontal pass runs first, sums are float64 (this file is compiled without fp contraction).
// SRC: 0 = uint8 (/255), 1 = int32 holding 16-bit samples (/65535), 2 = float32 already normalised.
template <int SRC>
__device__ __forceinline__ double cv_sample(const void* src, long i) {
  if (SRC == 0) return (double)static_cast<const unsigned char*>(src)[i] / 255.0;
  if (SRC == 1) return (double)static_cast<const int*>(src)[i] / 65535.0;
  return (double)static_cast<const float*>(src)[i];
}

__device__ __forceinline__ void cv_tap(int d, int src, int dst, int& s, float& f, bool& hi) {
  const double scale = (double)src / (double)dst;
  f = (float)(((double)d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  hi = s >= src - 1;
  if (hi) { s = src - 1; f = 0.f; }
}

template <int SRC>
__global__ __launch_bounds__(256) void resize_cv_kernel(const void* __restrict__ src, int h, int w, int c, int oh, int ow,
                                                        long total, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ch = idx % c;
  const long t = idx / c;
  const int ox = t % ow, oy = (t / ow) % oh;
  const long f = t / ((long)ow * oh);
  int sx, sy; float fx, fy; bool xhi, yhi;
  cv_tap(ox, w, ow, sx, fx, xhi);
  cv_tap(oy, h, oh, sy, fy, yhi);
  const int x1 = sx + 1 < w ? sx + 1 : w - 1, y1 = sy + 1 < h ? sy + 1 : h - 1;
  const double a0 = (double)(1.f - fx), a1 = (double)fx, b0 = (double)(1.f - fy), b1 = (double)fy;
  const long base = f * h * w;
  auto at = [&](int y, int x) { return cv_sample<SRC>(src, ((base + (long)y * w + x) * c) + ch); };
  double r0, r1;
  if (xhi) { r0 = at(sy, sx); r1 = at(y1, sx); }
  else {
    r0 = at(sy, sx) * a0 + at(sy, x1) * a1;
    r1 = at(y1, sx) * a0 + at(y1, x1) * a1;
  }
  out[idx] = (float)(b0 * r0 + b1 * r1);
}

}  // namespace

extern "C" int nlt_resize_cv_linear(const void* src, int src_kind, int n, int h, int w, int c, int oh, int ow, float* out,
                                    void* stream) {
  if (!src || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || oh <= 0 || ow <= 0) return NLT_ERR_BAD_ARG;
  if (src_kind < 0 || src_kind > 2) return NLT_ERR_BAD_ARG;
  const long total = (long)n * oh * ow * c;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(blocks_for(total));
  if (src_kind == 0) hipLaunchKernelGGL(resize_cv_kernel<0>, grid, dim3(256), 0, s, src, h, w, c, oh, ow, total, out);
  else if (src_kind == 1) hipLaunchKernelGGL(resize_cv_kernel<1>, grid, dim3(256), 0, s, src, h, w, c, oh, ow, total, out);
  else hipLaunchKernelGGL(resize_cv_kernel<2>, grid, dim3(256), 0, s, src, h, w, c, oh, ow, total, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_psnr_sums(const float* im1, const float* im2, const unsigned char* mask, long pixels, int channels,
                             double* workspace, double* out2, void* stream) {
  if (!im1 || !im2 || !workspace || !out2 || pixels <= 0) return NLT_ERR_BAD_ARG;
  if (channels != 1 && channels != 3) return NLT_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(psnr_partial_kernel, dim3(PSNR_BLOCKS), dim3(256), 0, s, im1, im2, mask, pixels, channels, workspace);
  hipLaunchKernelGGL(psnr_final_kernel, dim3(1), dim3(1), 0, s, workspace, out2);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_cosine_map(const double* locs, const double* normals, const unsigned char* valid,
                              const unsigned char* occluded, double sx, double sy, double sz, long pixels,
                              double* cos_out, unsigned char* u8_out, void* stream) {
  if (!locs || !normals || !valid || pixels <= 0 || (!cos_out && !u8_out)) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(cosine_kernel, dim3(blocks_for(pixels)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     locs, normals, valid, occluded, sx, sy, sz, pixels, cos_out, u8_out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_albedo(const unsigned char* rgb_frames, int frames, long elems, double* albedo,
                          void* workspace8, void* stream) {
  if (!rgb_frames || !albedo || !workspace8 || frames <= 0 || elems <= 0) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(workspace8, 0, 8, s) != hipSuccess) return NLT_ERR_LAUNCH;
  u64* mx = static_cast<u64*>(workspace8);
  hipLaunchKernelGGL(albedo_sum_kernel, dim3(blocks_for(elems)), dim3(256), 0, s, rgb_frames, frames, elems, albedo, mx);
  hipLaunchKernelGGL(albedo_div_kernel, dim3(blocks_for(elems)), dim3(256), 0, s, albedo, elems, mx);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_diffuse_base(const double* albedo, const unsigned char* lvis, int frames, long texels,
                                unsigned char* diffuse, void* stream) {
  if (!albedo || !lvis || !diffuse || frames <= 0 || texels <= 0) return NLT_ERR_BAD_ARG;
  const long total = (long)frames * texels * 3;
  hipLaunchKernelGGL(diffuse_kernel, dim3(blocks_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     albedo, lvis, texels, total, diffuse);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

#define NLT_REMAP_DISPATCH(KERNEL, ...)                                                                             \
  switch (map_dtype) {                                                                                              \
    case NLT_MAP_F64: hipLaunchKernelGGL(KERNEL<0>, dim3(blocks_for(out_px)), dim3(256), 0, s, __VA_ARGS__); break; \
    case NLT_MAP_F32: hipLaunchKernelGGL(KERNEL<1>, dim3(blocks_for(out_px)), dim3(256), 0, s, __VA_ARGS__); break; \
    case NLT_MAP_F16: hipLaunchKernelGGL(KERNEL<2>, dim3(blocks_for(out_px)), dim3(256), 0, s, __VA_ARGS__); break; \
    default: return NLT_ERR_BAD_ARG;                                                                                \
  }

extern "C" int nlt_remap_bilinear_u8(const unsigned char* src, int h, int w, int c, const void* mapping, int map_dtype,
                                     int ldm, int oh, int ow, int force_kbg, unsigned char* out, void* stream) {
  if (!src || !mapping || !out || h <= 0 || w <= 0 || c <= 0 || oh <= 0 || ow <= 0 || ldm < 2) return NLT_ERR_BAD_ARG;
  if (h > 32767 || w > 32767) return NLT_ERR_UNSUPPORTED;          // cv2.remap's own short-index limit
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long out_px = (long)oh * ow;
  NLT_REMAP_DISPATCH(remap_u8_kernel, src, h, w, c, mapping, ldm, out_px, force_kbg, out)
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_remap_bilinear_f32(const float* src, int h, int w, int c, const void* mapping, int map_dtype,
                                      int ldm, int oh, int ow, int force_kbg, float* out, void* stream) {
  if (!src || !mapping || !out || h <= 0 || w <= 0 || c <= 0 || oh <= 0 || ow <= 0 || ldm < 2) return NLT_ERR_BAD_ARG;
  if (h > 32767 || w > 32767) return NLT_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long out_px = (long)oh * ow;
  NLT_REMAP_DISPATCH(remap_f32_kernel, src, h, w, c, mapping, ldm, out_px, force_kbg, out)
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" long nlt_uv_index_map_workspace_bytes(int h, int w, long samples) {
  if (h <= 0 || w <= 0 || samples <= 0) return -1;
  const long cells = (long)h * w;
  return 4 * cells + 4 * samples + ((cells + 15) / 16) * 16;
}

extern "C" int nlt_uv_index_map(const double* uvs, const double* values, long samples, int m, int h, int w,
                                int max_l1, double fill, void* workspace, double* out, int* index_out, void* stream) {
  if (!uvs || !values || !workspace || !out || samples <= 0 || m <= 0) return NLT_ERR_BAD_ARG;
  if (h < 2 || w < 2 || max_l1 < 0) return NLT_ERR_BAD_ARG;
  if (max_l1 > 64 || samples >= (1l << 31) || (long)h * w >= (1l << 31)) return NLT_ERR_UNSUPPORTED;
  // A trusted texel has an occupied texel within L1 <= max_l1, i.e. a sample within (a+1, b+1) texels with
  // a + b <= max_l1; the nearest sample (Euclidean in uv units) is at most that far -> bounded search window.
  double dmax = 0.0;
  for (int a = 0; a <= max_l1; ++a) {
    const int b = max_l1 - a;
    const double du = (double)(a + 1) / (w - 1), dv = (double)(b + 1) / (h - 1);
    const double d = sqrt(du * du + dv * dv);
    if (d > dmax) dmax = d;
  }
  int rx = (int)(dmax * (w - 1)) + 2, ry = (int)(dmax * (h - 1)) + 2;
  if (rx > w - 1) rx = w - 1;
  if (ry > h - 1) ry = h - 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long cells = (long)h * w;
  int* head = static_cast<int*>(workspace);
  int* next = head + cells;
  u8* occ = reinterpret_cast<u8*>(next + samples);
  if (hipMemsetAsync(head, 0xFF, 4 * cells, s) != hipSuccess) return NLT_ERR_LAUNCH;
  if (hipMemsetAsync(occ, 0, cells, s) != hipSuccess) return NLT_ERR_LAUNCH;
  hipLaunchKernelGGL(uvmap_scatter_kernel, dim3(blocks_for(samples)), dim3(256), 0, s, uvs, samples, h, w, occ, head, next);
  hipLaunchKernelGGL(uvmap_query_kernel, dim3(blocks_for(cells)), dim3(256), 0, s, uvs, values, m, h, w, max_l1, ry, rx,
                     1.0 / (double)(w - 1), 1.0 / (double)(h - 1), fill, occ, head, next, out, index_out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_knn_indices(const double* ref_pos, int np, const double* cand_pos, int nq, int k, int* out, void* stream) {
  if (!ref_pos || !cand_pos || !out || np <= 0 || nq <= 0 || k <= 0) return NLT_ERR_BAD_ARG;
  hipLaunchKernelGGL(knn_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     ref_pos, np, cand_pos, nq, k, out);
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}

extern "C" int nlt_gather_frames_u8(const unsigned char* store, const int* ids, int n, long per_frame, float* out,
                                     void* stream) {
  if (!store || !out || n <= 0 || per_frame <= 0) return NLT_ERR_BAD_ARG;
  return launch_gather(store, ids, n, per_frame, out, static_cast<hipStream_t>(stream));
}

extern "C" int nlt_assemble_batch(const unsigned char* diffuse_store, const unsigned char* rgb_store,
                                  const unsigned char* cvis_store, const unsigned char* lvis_store,
                                  const int* ids, const int* nn_ids, int n, int k, long texels, int test_mode,
                                  float* base, float* cvis, float* lvis, float* rgb, float* nn_base, float* nn_rgb,
                                  void* stream) {
  if (!diffuse_store || !cvis_store || !lvis_store || !ids || !base || !cvis || !lvis || !rgb) return NLT_ERR_BAD_ARG;
  if (n <= 0 || k < 0 || texels <= 0) return NLT_ERR_BAD_ARG;
  if ((k > 0 || !test_mode) && !rgb_store) return NLT_ERR_BAD_ARG;
  if (k > 0 && (!nn_ids || !nn_base || !nn_rgb)) return NLT_ERR_BAD_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc;
  if ((rc = launch_gather(diffuse_store, ids, n, texels * 3, base, s)) != NLT_OK) return rc;
  if ((rc = launch_gather(cvis_store, ids, n, texels, cvis, s)) != NLT_OK) return rc;
  if ((rc = launch_gather(lvis_store, ids, n, texels, lvis, s)) != NLT_OK) return rc;
  if (test_mode) {                                             // nlt.py:126-128: rgb placeholder of zeros
    if (hipMemsetAsync(rgb, 0, (size_t)n * texels * 3 * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
  } else if ((rc = launch_gather(rgb_store, ids, n, texels * 3, rgb, s)) != NLT_OK) return rc;
  if (k > 0) {
    if ((rc = launch_gather(diffuse_store, nn_ids, n * k, texels * 3, nn_base, s)) != NLT_OK) return rc;
    if ((rc = launch_gather(rgb_store, nn_ids, n * k, texels * 3, nn_rgb, s)) != NLT_OK) return rc;
  }
  return NLT_OK;
}
