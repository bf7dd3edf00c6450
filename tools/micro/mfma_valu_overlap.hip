// Microbenchmark: do MFMA and VALU instructions overlap on a gfx950 SIMD -- inside one wave, and between the two waves of a SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/micro/mfma_valu_overlap.hip && ./mfma_valu_overlap
// Every test launches 256 x 4 workgroups-waves so that each SIMD of the chip gets exactly `wps` resident waves (launch_bounds keeps
// the allocation at 2 waves per SIMD); a wave runs ITER iterations of a body with NM independent MFMAs and NV independent VALU FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int KIND, int NM, int NV, int SPLITROLE>
__global__ __launch_bounds__(512, 1) void body(float* out, int iters, float seed) {
  // KIND 0: bf16 16x16x32 MFMA, 1: f32 16x16x4 MFMA.  SPLITROLE: 0 = every wave runs MFMA + VALU; 1 = waves 0-3 MFMA only, waves 4-7 (the second wave of each SIMD)
  // waves VALU only (same total instruction counts per SIMD when two waves share it)
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
  // r05 (r04 review, item 8): the MFMA operands live in registers of their own (`mb`); the VALU chain owns `v`.  The r04 form fed
  // v[m & 7] to the fp32 MFMA as its B operand while the interleaved v_fma_f32 wrote the same registers -- a RAW / WAR dependence
  // between the two instruction streams, which is why its "interleaved" fp32 figure (1901) exceeded MFMA alone + VALU alone.
  float v[8], mb[8];
  for (int i = 0; i < 8; ++i) { v[i] = seed + i; mb[i] = seed - 0.5f * i; }
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(mb[i]));            // opaque: stays in 8 distinct registers
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  const bool do_m = SPLITROLE == 0 || ((wave >> 2) & 1) == 0;
  const bool do_v = SPLITROLE == 0 || ((wave >> 2) & 1) == 1;
  const int nm = SPLITROLE ? 2 * NM : NM, nv = SPLITROLE ? 2 * NV : NV;
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < (SPLITROLE ? 2 * NM : NM); ++m) {
        if (KIND == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
        else acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, mb[m & 7], acc[m & 3], 0, 0, 0);
        if (SPLITROLE == 0 && NV > 0) {                                      // interleave: NV / NM VALU ops behind each MFMA
#pragma unroll
          for (int q = 0; q < (NM ? NV / NM : 0); ++q) { const int r = (m * (NV / (NM ? NM : 1)) + q) & 7; asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(seed)); }
        }
      }
    }
    if (do_v && (SPLITROLE == 1 || NM == 0)) {
#pragma unroll
      for (int q = 0; q < (SPLITROLE ? 2 * NV : NV); ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[q & 7]) : "v"(seed));
    }
    (void)nm; (void)nv;
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i] + mb[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND, int NM, int NV, int SPLITROLE>
float run(int wps, int iters) {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 64 * 4 * wps;                // wps waves on each of the CU's 4 SIMDs, one workgroup per CU
  body<KIND, NM, NV, SPLITROLE><<<256, threads>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  body<KIND, NM, NV, SPLITROLE><<<256, threads>>>(out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}

int main() {
  const int iters = 20000;
  // cycles per iteration per SIMD at 2.4 GHz = ms * 2.4e6 / iters
  auto cyc = [&](float ms) { return ms * 2.4e6f / iters; };
  printf("per-iteration cycles of a SIMD (2.4 GHz nominal); body = NM MFMAs + NV v_fma_f32 per wave\n");
  printf("bf16 16x16x32, 1 wave/SIMD:  32 MFMA only %7.1f   128 VALU only %7.1f   32 MFMA + 128 VALU interleaved %7.1f\n",
         cyc(run<0, 32, 0, 0>(1, iters)), cyc(run<0, 0, 128, 0>(1, iters)), cyc(run<0, 32, 128, 0>(1, iters)));
  printf("bf16 16x16x32, 2 waves/SIMD: 32 MFMA only %7.1f   128 VALU only %7.1f   32 MFMA + 128 VALU interleaved %7.1f   one wave 64 MFMA, other 256 VALU %7.1f\n",
         cyc(run<0, 32, 0, 0>(2, iters)), cyc(run<0, 0, 128, 0>(2, iters)), cyc(run<0, 32, 128, 0>(2, iters)), cyc(run<0, 32, 128, 1>(2, iters)));
  printf("f32 16x16x4,   1 wave/SIMD:  32 MFMA only %7.1f   128 VALU only %7.1f   32 MFMA + 128 VALU interleaved %7.1f\n",
         cyc(run<1, 32, 0, 0>(1, iters)), cyc(run<1, 0, 128, 0>(1, iters)), cyc(run<1, 32, 128, 0>(1, iters)));
  printf("f32 16x16x4,   2 waves/SIMD: 32 MFMA only %7.1f   128 VALU only %7.1f   32 MFMA + 128 VALU interleaved %7.1f   one wave 64 MFMA, other 256 VALU %7.1f\n",
         cyc(run<1, 32, 0, 0>(2, iters)), cyc(run<1, 0, 128, 0>(2, iters)), cyc(run<1, 32, 128, 0>(2, iters)), cyc(run<1, 32, 128, 1>(2, iters)));
  return 0;
}
