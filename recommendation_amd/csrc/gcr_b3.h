// Split-operand ("b3") MFMA helpers shared by the InfoNCE / k-means kernels (gcr_infonce.hip) and the fused
// ranking kernel (gcr_rank.hip).  Included INSIDE each translation unit's anonymous namespace, after
// f32x16 / kTileJ / acc_row are defined there.
#pragma once
// ------------------------------------------------------------------------------------------
// Split-operand engine ("b3").  gfx950's bf16 MFMA (v_mfma_f32_32x32x16_bf16) runs at 16x the rate
// of the f32 MFMA and accumulates in f32.  Every f32 operand x is split ERROR-FREE into three
// bf16 planes x = x1 + x2 + x3 (+ <= 2^-27 |x|): x1 = bf16_rn(x), x2 = bf16_rn(x - x1),
// x3 = bf16_rn(x - x1 - x2), the subtractions being exact in f32; bf16 has the exponent range of
// f32, so this holds for any finite input.  A product x*y is then the six terms
//   x1y1 + (x1y2 + x2y1) + (x2y2 + x1y3 + x3y1),
// the dropped ones (x2y3, x3y2, x3y3) being <= 2^-26 |xy|, below the rounding of the f32
// accumulator that both engines share.  Six bf16 MFMAs of K = 16 replace eight f32 MFMAs of K = 2
// per 16 features: 2.67x fewer matrix-core cycles at f32 accuracy (the parity tests run on both
// engines with the same tolerances).  The split happens once per element: anchors when they are
// loaded, table rows on their way into LDS (three planes, 16-B padded rows, conflict-free b128).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// two f32 -> three packed bf16 pairs (planes 1..3)
__device__ __forceinline__ void split3(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = pack_bf16(a, b);
  a -= bf16_lo(p1);
  b -= bf16_hi(p1);
  p2 = pack_bf16(a, b);
  a -= bf16_lo(p2);
  b -= bf16_hi(p2);
  p3 = pack_bf16(a, b);
}

template <int D>
struct ShapeB3 {
  static constexpr int KH = D / 2;                       // features per lane half
  static constexpr int KC = KH / 8;                      // MFMA k-chunks (8 bf16 per lane) per product term
  static constexpr int ROWB = 2 * D + 16;                // bytes per LDS row and plane (+16 B pad)
  static constexpr int PLANE = kTileJ * ROWB;            // bytes per plane
  static constexpr int NT = D <= 64 ? 2 : 1;             // anchor tiles of 32 per wave
  static constexpr int NLD = (kTileJ * D / 4) / 256;
  static constexpr int ANCHORS_PER_BLOCK = 4 * 32 * NT;
};

template <int D>
__device__ __forceinline__ void stage_store_b3_one(unsigned char* __restrict__ tile, int tid, const float4& v, int u) {
  using S = ShapeB3<D>;
  const int idx = tid + 256 * u;
  const int row = idx / (D / 4), c4 = idx % (D / 4);
  unsigned a1, a2, a3, b1, b2, b3;
  split3(v.x, v.y, a1, a2, a3);
  split3(v.z, v.w, b1, b2, b3);
  unsigned char* p = tile + row * S::ROWB + c4 * 8;
  *reinterpret_cast<uint2*>(p) = make_uint2(a1, b1);
  *reinterpret_cast<uint2*>(p + S::PLANE) = make_uint2(a2, b2);
  *reinterpret_cast<uint2*>(p + 2 * S::PLANE) = make_uint2(a3, b3);
}

template <int D>
__device__ __forceinline__ void stage_store_b3(unsigned char* __restrict__ tile, int tid,
                                               const float4 (&regs)[ShapeB3<D>::NLD]) {
#pragma unroll
  for (int u = 0; u < ShapeB3<D>::NLD; ++u) stage_store_b3_one<D>(tile, tid, regs[u], u);
}

// stationary operand planes: frag[p][c] = 8 consecutive features [h*KH + 8c, +8) of plane p
template <int D>
__device__ __forceinline__ void load_stationary_b3(const float* __restrict__ a, const float* __restrict__ a_scale,
                                                   int64_t m_rows, int64_t row, int h, float mult,
                                                   u32x4 (&frag)[3][ShapeB3<D>::KC]) {
  using S = ShapeB3<D>;
  const bool valid = row < m_rows;
  const float s = valid ? (a_scale != nullptr ? a_scale[row] : 1.0f) * mult : 0.f;
  const float* p = a + (valid ? row : 0) * D + h * S::KH;
#pragma unroll
  for (int c = 0; c < S::KC; ++c) {
    const float4 v0 = valid ? *reinterpret_cast<const float4*>(p + 8 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = valid ? *reinterpret_cast<const float4*>(p + 8 * c + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned q[3][4];
    split3(v0.x * s, v0.y * s, q[0][0], q[1][0], q[2][0]);
    split3(v0.z * s, v0.w * s, q[0][1], q[1][1], q[2][1]);
    split3(v1.x * s, v1.y * s, q[0][2], q[1][2], q[2][2]);
    split3(v1.z * s, v1.w * s, q[0][3], q[1][3], q[2][3]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) frag[pl][c] = (u32x4){q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
  }
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// S^T tile as in score_tile, six bf16 terms per k-chunk, smallest terms first
template <int D, int NT>
__device__ __forceinline__ void score_tile_b3(const unsigned char* __restrict__ tile, int i32, int h,
                                              const u32x4 (&bq)[NT][3][ShapeB3<D>::KC], f32x16 (&acc)[NT]) {
  using S = ShapeB3<D>;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const unsigned char* base = tile + i32 * S::ROWB + h * (S::KH * 2);
#pragma unroll
  for (int c = 0; c < S::KC; ++c) {
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(base + 16 * c);
    const u32x4 a2 = *reinterpret_cast<const u32x4*>(base + S::PLANE + 16 * c);
    const u32x4 a3 = *reinterpret_cast<const u32x4*>(base + 2 * S::PLANE + 16 * c);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a3, bq[t][0][c], acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a1, bq[t][2][c], acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a2, bq[t][1][c], acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a2, bq[t][0][c], acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a1, bq[t][1][c], acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a1, bq[t][0][c], acc[t]);
  }
}

