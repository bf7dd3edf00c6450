// Barron robust image loss as NLT fixes it (alpha = 1 Charbonnier NLL, c = 0.01) on a 5-level
// CDF 9/7 wavelet decomposition of the sYUV residual -- forward value and d(loss)/d(pred).
//   replaces: nlt/losses.py:90-118 -> third_party/robust_loss/adaptive.py:453-538
//             (wavelet.py:164-205,286-334; util.py:96-115; general.py:104-112; distribution.py:181-222)
// Planar working layout [n*3][H][W]; every pass is HBM-bound and tiny next to the network.
#include "nlt_common.h"
#include <math.h>

namespace {

__constant__ float kLO[9] = {+0.037828455507f, -0.023849465020f, -0.110624404418f, +0.377402855613f, +0.852698679009f,
                             +0.377402855613f, -0.110624404418f, -0.023849465020f, +0.037828455507f};
__constant__ float kHI[7] = {+0.064538882629f, -0.040689417609f, -0.418092273222f, +0.788485616406f,
                             -0.418092273222f, -0.040689417609f, +0.064538882629f};
// tf.image.rgb_to_yuv kernel (rows R,G,B -> Y,U,V) times robust_loss' volume-preserving 1.580227820074
__constant__ float kSYUV[9] = {0.299f * 1.580227820074f, -0.14714119f * 1.580227820074f, 0.61497538f * 1.580227820074f,
                               0.587f * 1.580227820074f, -0.28886916f * 1.580227820074f, -0.51496512f * 1.580227820074f,
                               0.114f * 1.580227820074f, 0.43601035f * 1.580227820074f, -0.10001026f * 1.580227820074f};

constexpr float kScale = 0.01f;                 // nlt/losses.py:94
constexpr double kLogZ1 = 1.1854952325;         // log Z(alpha = 1) of partition_spline.npz = log(2 e K1(1))

inline unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256); }

__device__ __forceinline__ int reflect(int j, int n) {          // wavelet.py:138-145
  const int period = max(1, 2 * (n - 1));
  int jm = j % period;
  if (jm < 0) jm += period;
  return min(2 * (n - 1) - jm, jm);
}

__host__ __device__ inline int n_lo(int n) { return (n - 1) / 2 + 1; }
__host__ __device__ inline int n_hi(int n) { return n >= 2 ? (n - 2) / 2 + 1 : 0; }

__global__ __launch_bounds__(256) void residual_syuv_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            int hw, long total, float* __restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int f = p / hw;
  const long pix = p - (long)f * hw;
  const float r0 = gt[p * 3 + 0] - pred[p * 3 + 0], r1 = gt[p * 3 + 1] - pred[p * 3 + 1], r2 = gt[p * 3 + 2] - pred[p * 3 + 2];
#pragma unroll
  for (int oc = 0; oc < 3; ++oc)
    out[((long)f * 3 + oc) * hw + pix] = r0 * kSYUV[0 * 3 + oc] + r1 * kSYUV[1 * 3 + oc] + r2 * kSYUV[2 * 3 + oc];
}

// One separable analysis pass along `axis` (0: rows / H, 1: columns / W) of [P][A][B] planes:
// lo = downsample(x, analysis_lo, shift 0), hi = downsample(x, analysis_hi, shift 1).
__global__ __launch_bounds__(256) void dwt_axis_kernel(const float* __restrict__ in, int A, int B, int axis, long total,
                                                       float* __restrict__ lo, float* __restrict__ hi) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int n = axis == 0 ? A : B;
  const int nl = n_lo(n), nh = n_hi(n);
  const int oA = axis == 0 ? nl : A, oB = axis == 0 ? B : nl;       // lo grid (>= hi grid)
  const int b = idx % oB;
  const int a = (idx / oB) % oA;
  const long pl = idx / ((long)oA * oB);
  const float* x = in + pl * (long)A * B;
  const int i = axis == 0 ? a : b;
  const long stride = axis == 0 ? B : 1;
  const long base = axis == 0 ? b : (long)a * B;
  float sl = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) sl += kLO[t] * x[base + (long)reflect(2 * i + t - 4, n) * stride];
  lo[pl * (long)oA * oB + (long)a * oB + b] = sl;
  if (i < nh) {
    float sh = 0.f;
#pragma unroll
    for (int t = 0; t < 7; ++t) sh += kHI[t] * x[base + (long)reflect(2 * i + 1 + t - 3, n) * stride];
    const int hA = axis == 0 ? nh : A, hB = axis == 0 ? B : nh;
    hi[pl * (long)hA * hB + (long)a * hB + b] = sh;
  }
}

// Adjoint of dwt_axis_kernel: din += D_lo^T dlo + D_hi^T dhi  (din zeroed by the launcher).
__global__ __launch_bounds__(256) void dwt_axis_T_kernel(const float* __restrict__ dlo, const float* __restrict__ dhi,
                                                         int A, int B, int axis, long total, float* din) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int n = axis == 0 ? A : B;
  const int nl = n_lo(n), nh = n_hi(n);
  const int oA = axis == 0 ? nl : A, oB = axis == 0 ? B : nl;
  const int b = idx % oB;
  const int a = (idx / oB) % oA;
  const long pl = idx / ((long)oA * oB);
  float* x = din + pl * (long)A * B;
  const int i = axis == 0 ? a : b;
  const long stride = axis == 0 ? B : 1;
  const long base = axis == 0 ? b : (long)a * B;
  const float gl = dlo[pl * (long)oA * oB + (long)a * oB + b];
#pragma unroll
  for (int t = 0; t < 9; ++t) atomicAdd(x + base + (long)reflect(2 * i + t - 4, n) * stride, kLO[t] * gl);
  if (i < nh) {
    const int hA = axis == 0 ? nh : A, hB = axis == 0 ? B : nh;
    const float gh = dhi[pl * (long)hA * hB + (long)a * hB + b];
#pragma unroll
    for (int t = 0; t < 7; ++t) atomicAdd(x + base + (long)reflect(2 * i + 1 + t - 3, n) * stride, kHI[t] * gh);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// band [n*3 planes][count]: loss[f] += sum rho(w) * inv_total ; band <- rho'(w) * inv_total (in place)
// rho(w) = sqrt((w/c)^2 + 1) - 1, rho'(w) = (w/c^2) / sqrt((w/c)^2 + 1).   blockIdx.y = frame.
__global__ __launch_bounds__(256) void charbonnier_kernel(float* band, long per_frame, float inv_total, int want_grad,
                                                          float* loss) {
  __shared__ float ws[4];
  const int f = blockIdx.y;
  float* w = band + (long)f * per_frame;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_frame; i += (long)gridDim.x * blockDim.x) {
    const float u = w[i] / kScale;
    const float r = sqrtf(u * u + 1.f);
    s += r - 1.f;
    if (want_grad) w[i] = (u / kScale) / r * inv_total;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss + f, (ws[0] + ws[1] + ws[2] + ws[3]) * inv_total);
}

__global__ void add_const_kernel(float* loss, int n, float c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) loss[i] += c;
}

// dpred[f,pix,ic] = - sum_oc dX[f*3+oc][pix] * M[ic][oc]        (r = gt - pred)
__global__ __launch_bounds__(256) void grad_finish_kernel(const float* __restrict__ dx, int hw, long total,
                                                          float* __restrict__ dpred) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int f = p / hw;
  const long pix = p - (long)f * hw;
  const float g0 = dx[((long)f * 3 + 0) * hw + pix], g1 = dx[((long)f * 3 + 1) * hw + pix], g2 = dx[((long)f * 3 + 2) * hw + pix];
#pragma unroll
  for (int ic = 0; ic < 3; ++ic)
    dpred[p * 3 + ic] = -(g0 * kSYUV[ic * 3 + 0] + g1 * kSYUV[ic * 3 + 1] + g2 * kSYUV[ic * 3 + 2]);
}

// ---- fewer, fatter launches (the loss is ~20 tiny passes: launch count, not bytes, is its cost) -------------------------
// Second (column) analysis pass of a level for BOTH row sets in one launch, with the Charbonnier term of every band that
// is final fused in: blockIdx.z = 0: rows of `lo` -> (LL, LH), 1: rows of `hi` -> (HL, HH); blockIdx.y = frame.
// LH / HL / HH are final at every level, LL only at the last one (`ll_final`).  Same sums, same order as dwt_axis_kernel +
// charbonnier_kernel (the per-frame loss is accumulated with float atomics there as well).
__global__ __launch_bounds__(256) void dwt_cols_charb_kernel(const float* __restrict__ lo_rows, const float* __restrict__ hi_rows,
                                                             int hl, int hh, int w, float* LL, float* LH, float* HL, float* HH,
                                                             int ll_final, float inv_total, int want_grad, float* loss) {
  __shared__ float red[4];
  const int z = blockIdx.z, f = blockIdx.y;
  const int A = z == 0 ? hl : hh;
  const float* src = z == 0 ? lo_rows : hi_rows;
  float* out_lo = z == 0 ? LL : HL;
  float* out_hi = z == 0 ? LH : HH;
  const int wl = n_lo(w), wh = n_hi(w);
  const long per_frame = 3l * A * wl;
  float s = 0.f;
  // grid-stride: the launcher caps the workgroups per frame, because each ends in ONE float atomic on the frame's loss and
  // device-scope atomics on one address serialise (~50 ns each: 1500 of them per frame at level 0 were most of this launch)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < per_frame; idx += (long)gridDim.x * blockDim.x) {
    const int b = idx % wl;
    const int a = (idx / wl) % A;
    const long pl = (long)f * 3 + idx / ((long)A * wl);
    const float* x = src + pl * (long)A * w + (long)a * w;
    float sl = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) sl += kLO[t] * x[reflect(2 * b + t - 4, w)];
    if (z == 1 || ll_final) {
      const float u = sl / kScale;
      const float r = sqrtf(u * u + 1.f);
      s += r - 1.f;
      if (want_grad) sl = (u / kScale) / r * inv_total;
    }
    out_lo[pl * (long)A * wl + (long)a * wl + b] = sl;
    if (b < wh) {
      float sh = 0.f;
#pragma unroll
      for (int t = 0; t < 7; ++t) sh += kHI[t] * x[reflect(2 * b + 1 + t - 3, w)];
      const float u = sh / kScale;
      const float r = sqrtf(u * u + 1.f);
      s += r - 1.f;
      if (want_grad) sh = (u / kScale) / r * inv_total;
      out_hi[pl * (long)A * wh + (long)a * wh + b] = sh;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss + f, (red[0] + red[1] + red[2] + red[3]) * inv_total);
}

// Adjoint of one analysis pass in GATHER form: one thread per element of the pass's INPUT adds the (tap, coefficient) pairs
// that touched it -- no zero-fill, no atomics, deterministic.  With n >= 6 every tap position lies in [-4, n + 3] and is
// reflected at most once, so element i collects the virtual positions {i, -i, 2(n-1) - i}.
__device__ __forceinline__ float dwt_T_gather(const float* __restrict__ glo, const float* __restrict__ ghi, long lo_stride,
                                              long hi_stride, int i, int n) {
  const int nl = n_lo(n), nh = n_hi(n);
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    if ((v == 1 && i == 0) || (v == 2 && i == n - 1)) continue;
    const int p = v == 0 ? i : (v == 1 ? -i : 2 * (n - 1) - i);
    if (p < -4 || p > n + 3) continue;
    // lo: position 2 j + t - 4 = p, t in [0, 8]
    int j0 = (p - 4 + 1) >> 1, j1 = (p + 4) >> 1;                 // ceil((p - 4) / 2), floor((p + 4) / 2)  (arithmetic shifts)
    if (j0 < 0) j0 = 0;
    if (j1 > nl - 1) j1 = nl - 1;
    for (int j = j0; j <= j1; ++j) s += kLO[p - 2 * j + 4] * glo[(long)j * lo_stride];
    // hi: position 2 j + 1 + t - 3 = p, t in [0, 6]
    j0 = (p - 4 + 1) >> 1; j1 = (p + 2) >> 1;
    if (j0 < 0) j0 = 0;
    if (j1 > nh - 1) j1 = nh - 1;
    for (int j = j0; j <= j1; ++j) s += kHI[p - 2 * j + 2] * ghi[(long)j * hi_stride];
  }
  return s;
}

// columns: (LL, LH) -> lo rows and (HL, HH) -> hi rows in one launch (blockIdx.z as above)
__global__ __launch_bounds__(256) void dwt_cols_T_kernel(const float* __restrict__ LL, const float* __restrict__ LH,
                                                         const float* __restrict__ HL, const float* __restrict__ HH,
                                                         int hl, int hh, int w, long planes, float* lo_rows, float* hi_rows) {
  const int z = blockIdx.z;
  const int A = z == 0 ? hl : hh;
  const long total = planes * A * w;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int wl = n_lo(w), wh = n_hi(w);
  const int b = idx % w;
  const long row = idx / w;                                          // plane * A + a
  const float* glo = (z == 0 ? LL : HL) + row * wl;
  const float* ghi = (z == 0 ? LH : HH) + row * wh;
  (z == 0 ? lo_rows : hi_rows)[idx] = dwt_T_gather(glo, ghi, 1, 1, b, w);
}

// rows: (lo, hi) -> the level's input image
__global__ __launch_bounds__(256) void dwt_rows_T_kernel(const float* __restrict__ lo_rows, const float* __restrict__ hi_rows,
                                                         int h, int w, long planes, float* din) {
  const long total = planes * h * w;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int hl = n_lo(h), hh = n_hi(h);
  const int b = idx % w;
  const int a = (idx / w) % h;
  const long pl = idx / ((long)h * w);
  din[idx] = dwt_T_gather(lo_rows + pl * (long)hl * w + b, hi_rows + pl * (long)hh * w + b, w, w, a, h);
}

__global__ void init_loss_kernel(float* loss, int n, float c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) loss[i] = c;
}

struct Level { int h, w, hl, hh, wl, wh; long lo, hi, HH, LH, HL, LL; };

constexpr int kLevels = 5;                      // nlt/losses.py:103

long layout(int n, int h, int w, Level* lv, long* x0) {
  const long P = (long)n * 3;
  long off = 0;
  *x0 = off; off += P * h * w;
  for (int l = 0; l < kLevels; ++l) {
    Level& L = lv[l];
    L.h = h; L.w = w; L.hl = n_lo(h); L.hh = n_hi(h); L.wl = n_lo(w); L.wh = n_hi(w);
    L.lo = off; off += P * L.hl * w;
    L.hi = off; off += P * L.hh * w;
    L.HH = off; off += P * L.hh * L.wh;
    L.LH = off; off += P * L.hl * L.wh;
    L.HL = off; off += P * L.hh * L.wl;
    L.LL = off; off += P * L.hl * L.wl;
    h = L.hl; w = L.wl;
  }
  return off;
}

}  // namespace

extern "C" long nlt_barron_workspace_floats(int n, int h, int w) {
  if (n <= 0 || h < 2 || w < 2) return -1;
  Level lv[kLevels]; long x0;
  return layout(n, h, w, lv, &x0);
}

extern "C" int nlt_barron_loss(const float* pred, const float* gt, int n, int h, int w, float* workspace,
                               float* loss, float* dpred_unit, void* stream) {
  if (!pred || !gt || !workspace || !loss || n <= 0) return NLT_ERR_BAD_ARG;
  // wavelet.get_max_num_levels: ceil(log2(min size)) >= 5 levels
  int mn = h < w ? h : w;
  if (mn < 17) return NLT_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Level lv[kLevels]; long x0;
  layout(n, h, w, lv, &x0);
  float* ws = workspace;
  const long P = (long)n * 3;
  const int want_grad = dpred_unit != nullptr;
  const float inv_total = 1.f / ((float)h * (float)w * 3.f);
  // the NLL's constant first (every later pass adds its Charbonnier sums to it): no zero-fill + add-constant pair
  hipLaunchKernelGGL(init_loss_kernel, dim3((n + 63) / 64), dim3(64), 0, s, loss, n, (float)(log((double)kScale) + kLogZ1));
  {
    const long total = (long)n * h * w;
    hipLaunchKernelGGL(residual_syuv_kernel, dim3(blocks_for(total)), dim3(256), 0, s, pred, gt, h * w, total, ws + x0);
  }
  auto charb = [&](long off, long per_plane) {
    const long per_frame = per_plane * 3;
    long bx = (per_frame + 255) / 256; if (bx > 48) bx = 48;
    hipLaunchKernelGGL(charbonnier_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, s, ws + off, per_frame,
                       inv_total, want_grad, loss);
  };
  long xin = x0;
  for (int l = 0; l < kLevels; ++l) {
    const Level& L = lv[l];
    const bool last = l == kLevels - 1;
    hipLaunchKernelGGL(dwt_axis_kernel, dim3(blocks_for(P * L.hl * L.w)), dim3(256), 0, s, ws + xin, L.h, L.w, 0,
                       P * L.hl * L.w, ws + L.lo, ws + L.hi);
    if (L.hh > 0) {
      // rows of `lo` -> (LL, LH), rows of `hi` -> (HL, HH), Charbonnier of the final bands: one launch
      const long per_frame = 3l * L.hl * L.wl;                         // hl >= hh: the z = 1 half exits early
      unsigned bx = blocks_for(per_frame);
      if (bx > 48) bx = 48;
      hipLaunchKernelGGL(dwt_cols_charb_kernel, dim3(bx, (unsigned)n, 2), dim3(256), 0, s, ws + L.lo, ws + L.hi,
                         L.hl, L.hh, L.w, ws + L.LL, ws + L.LH, ws + L.HL, ws + L.HH, last ? 1 : 0, inv_total, want_grad, loss);
    } else {
      hipLaunchKernelGGL(dwt_axis_kernel, dim3(blocks_for(P * L.hl * L.wl)), dim3(256), 0, s, ws + L.lo, L.hl, L.w, 1,
                         P * L.hl * L.wl, ws + L.LL, ws + L.LH);
      if ((long)L.hl * L.wh > 0) charb(L.LH, (long)L.hl * L.wh);
      if (last) charb(L.LL, (long)L.hl * L.wl);
    }
    xin = L.LL;
  }
  NLT_CHECK_LAUNCH();
  if (!want_grad) return NLT_OK;

  // backward: bands now hold d(loss_f)/d(coefficient); run the adjoint transform coarse -> fine
  for (int l = kLevels - 1; l >= 0; --l) {
    const Level& L = lv[l];
    const long dst = l == 0 ? x0 : lv[l - 1].LL;          // gradient w.r.t. this level's input image
    if (L.w >= 6 && L.hh > 0) {
      hipLaunchKernelGGL(dwt_cols_T_kernel, dim3(blocks_for(P * L.hl * L.w), 1, 2), dim3(256), 0, s, ws + L.LL, ws + L.LH,
                         ws + L.HL, ws + L.HH, L.hl, L.hh, L.w, P, ws + L.lo, ws + L.hi);
    } else {
      if (hipMemsetAsync(ws + L.lo, 0, (size_t)P * L.hl * L.w * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
      if (hipMemsetAsync(ws + L.hi, 0, (size_t)P * L.hh * L.w * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
      hipLaunchKernelGGL(dwt_axis_T_kernel, dim3(blocks_for(P * L.hl * L.wl)), dim3(256), 0, s, ws + L.LL, ws + L.LH,
                         L.hl, L.w, 1, P * L.hl * L.wl, ws + L.lo);
      if (L.hh > 0)
        hipLaunchKernelGGL(dwt_axis_T_kernel, dim3(blocks_for(P * L.hh * L.wl)), dim3(256), 0, s, ws + L.HL, ws + L.HH,
                           L.hh, L.w, 1, P * L.hh * L.wl, ws + L.hi);
    }
    if (L.h >= 6) {
      hipLaunchKernelGGL(dwt_rows_T_kernel, dim3(blocks_for(P * L.h * L.w)), dim3(256), 0, s, ws + L.lo, ws + L.hi, L.h, L.w, P,
                         ws + dst);
    } else {
      if (hipMemsetAsync(ws + dst, 0, (size_t)P * L.h * L.w * sizeof(float), s) != hipSuccess) return NLT_ERR_LAUNCH;
      hipLaunchKernelGGL(dwt_axis_T_kernel, dim3(blocks_for(P * L.hl * L.w)), dim3(256), 0, s, ws + L.lo, ws + L.hi,
                         L.h, L.w, 0, P * L.hl * L.w, ws + dst);
    }
  }
  {
    const long total = (long)n * h * w;
    hipLaunchKernelGGL(grad_finish_kernel, dim3(blocks_for(total)), dim3(256), 0, s, ws + x0, h * w, total, dpred_unit);
  }
  NLT_CHECK_LAUNCH();
  return NLT_OK;
}
