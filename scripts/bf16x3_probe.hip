// Probe for the next round: accuracy and operand layout of a split-bf16 ("bf16x3") product on
// v_mfma_f32_32x32x16_bf16 against the exact-f32 MFMA and a double reference.   Z[32 x 32] = A[32 x K] * B[K x 32].
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16x3 scripts/bf16x3_probe.hip && /tmp/bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 64;

__device__ inline void split(float a, __bf16 &hi, __bf16 &lo) {
  hi = (__bf16)a;
  lo = (__bf16)(a - (float)hi);
}
__device__ inline void split3(float a, __bf16 &hi, __bf16 &mid, __bf16 &lo) {
  hi = (__bf16)a;
  const float r = a - (float)hi;
  mid = (__bf16)r;
  lo = (__bf16)(r - (float)mid);
}

// A is [32][K] row-major, B is [32][K] (column n of the product is row n here: both operands are K-contiguous)
__global__ void probe(const float *A, const float *B, float *Zf32, float *Zx3, float *Zx1, int terms) {
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  // exact f32: 32x32x2, lane (row = l31, k = h) per step
  f32x16 c = {0};
  for (int s = 0; s < K / 2; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + 2 * s + h], B[l31 * K + 2 * s + h], c, 0, 0, 0);
  // bf16: 32x32x16, lane (row = l31, k = 8 h .. 8 h + 7) per step
  f32x16 c3 = {0}, c1 = {0};
  for (int s = 0; s < K / 16; ++s) {
    bf16x8 ah, al, bh, bl;
    for (int q = 0; q < 8; ++q) {
      __bf16 x, y;
      split(A[l31 * K + 16 * s + 8 * h + q], x, y); ah[q] = x; al[q] = y;
      split(B[l31 * K + 16 * s + 8 * h + q], x, y); bh[q] = x; bl[q] = y;
    }
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c1, 0, 0, 0);
    if (terms == 6) {
      // three-way split, six of the nine partial products (what pmf_fused_sb_kernel's forward uses)
      bf16x8 am, bm;
      for (int q = 0; q < 8; ++q) {
        __bf16 x, y, z;
        split3(A[l31 * K + 16 * s + 8 * h + q], x, y, z); ah[q] = x; am[q] = y; al[q] = z;
        split3(B[l31 * K + 16 * s + 8 * h + q], x, y, z); bh[q] = x; bm[q] = y; bl[q] = z;
      }
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c3, 0, 0, 0);
      continue;
    }
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c3, 0, 0, 0);   // small terms first
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c3, 0, 0, 0);
    if (terms >= 4) {
      // (lo * lo is ~2^-16 of the product: a fourth MFMA buys little unless the split itself is refined)
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, c3, 0, 0, 0);
    }
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c3, 0, 0, 0);
  }
  // accumulator layout (both shapes): lane -> column n = l31; register r -> row m = (r & 3) + 8 (r >> 2) + 4 h
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
    Zf32[m * 32 + l31] = c[r];
    Zx3[m * 32 + l31] = c3[r];
    Zx1[m * 32 + l31] = c1[r];
  }
}

int main() {
  std::vector<float> A(32 * K), B(32 * K);
  srand(7);
  for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float *dA, *dB, *dZ;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dZ, 3 * 1024 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  for (int terms : {3, 4, 6}) {
    probe<<<1, 64>>>(dA, dB, dZ, dZ + 1024, dZ + 2048, terms);
    std::vector<float> Z(3 * 1024);
    hipMemcpy(Z.data(), dZ, Z.size() * 4, hipMemcpyDeviceToHost);
    double e32 = 0, e3 = 0, e1 = 0, zmax = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)A[m * K + k] * (double)B[n * K + k];
        zmax = fmax(zmax, fabs(ref));
        e32 = fmax(e32, fabs(Z[m * 32 + n] - ref));
        e3 = fmax(e3, fabs(Z[1024 + m * 32 + n] - ref));
        e1 = fmax(e1, fabs(Z[2048 + m * 32 + n] - ref));
      }
    printf("K=%d, max|Z|=%.3f: max abs error / max|Z|:  f32 MFMA %.2e   bf16 x%d %.2e   plain bf16 %.2e\n", K, zmax, e32 / zmax,
           terms, e3 / zmax, e1 / zmax);
  }
  return 0;
}
