// Tile geometry, packed-blob offsets and small helpers shared by the fused front kernels (fused.hip: first generation,
// also the training variant; front3.hip: LDS-staged raw tiles, float and uint8-store inputs).
#pragma once
#include "nlt_common.h"

namespace {

constexpr int TH = 8, TW = 16;             // half-resolution tile owned by one workgroup
constexpr int HH = TH + 1, HW = TW + 1;    // with the 1-texel halo the stride-1 conv needs
constexpr int HT = HH * HW;                // 153 haloed texels
constexpr int NT = (HT + 15) / 16;         // 10 MFMA column tiles
constexpr int PLT = 160;                   // LDS plane of one channel quad of one path's haloed tile (153 -> 160 slots)
constexpr int PATH = 4 * PLT * 4;          // floats per path: [kk][PLT][4] -- planar by channel quad, so the 16 lanes
                                           // a ds_read_b128 services together hit 16 different 16-byte slots

// packed-blob offsets (floats); written by front_pack_kernel, read by front_kernel
constexpr int OFF_AQ2 = 0;                 // [8][64]     folded q stride-2 conv, MFMA m x lane
constexpr int OFF_AO2 = 512;               // [3][64]     folded obs stride-2 conv
constexpr int OFF_AQ1 = 704;               // [4][64][4]  q stride-1 conv: tap x lane x s4
constexpr int OFF_AO1 = 1728;              // [4][64][4]
constexpr int OFF_BQ2 = 2752, OFF_BO2 = 2768, OFF_BQ1 = 2784, OFF_BO1 = 2800;   // [16] each
constexpr int OFF_WSK = 2816;              // [8][3]  head share of the raw channels
constexpr int OFF_BSK = 2840;              // [3]
constexpr int OFF_WSK8 = 2848;             // [8][3]  the same rows times fl(1 / 255): the uint8-store front kernel feeds raw BYTE values (r05)
constexpr int BLOB = 2880;

__device__ __forceinline__ f32x4 lrelu4(f32x4 v, float alpha) {
  return (f32x4){v[0] > 0.f ? v[0] : alpha * v[0], v[1] > 0.f ? v[1] : alpha * v[1],
                 v[2] > 0.f ? v[2] : alpha * v[2], v[3] > 0.f ? v[3] : alpha * v[3]};
}

// second blob: level 2's stride-2 convs (fused into the front kernel when k <= 4)
constexpr int OFF3_AQ = 0;                 // [2 rt][8 = slab*4 + c4][64][4]  query (2,2,32,32): slab 0 = q1, 1 = mean o1
constexpr int OFF3_AO = 4096;              // [2 rt][4 c4][64][4]             obs   (2,2,16,32)
constexpr int OFF3_BQ = 6144, OFF3_BO = 6176, BLOB3 = 6208;

// XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8; give each XCD a contiguous run
// of tiles so that neighbouring tiles (which share halo lines) meet in the same L2.
__device__ __forceinline__ int xcd_tile(int b, int nblocks) {
  return (nblocks & 7) ? b : (b & 7) * (nblocks >> 3) + (b >> 3);
}

}  // namespace
