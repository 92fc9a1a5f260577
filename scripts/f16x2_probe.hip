// Probe (round 3): can the FORWARD product of the K = 128 split kernel run on f16 pairs instead of bf16
// triples?  An f32 operand as hi = f16(a), lo = f16(a - hi): 22 significant bits in two terms (bf16 needs three), i.e.
// 2/3 of the operand bytes / registers and 3-4 MFMAs per k-step instead of 6 -- IF v_mfma_f32_32x32x16_f16 does not
// flush subnormal inputs (lo is subnormal for |a| < 0.12) and the range (|a| < 65504) is handled.
//   Z[32 x 32] = A[32 x K] * B[K x 32];  errors against fp64, relative to max|Z|.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/f16x2 scripts/f16x2_probe.hip && /tmp/f16x2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 128;

// mode 0: 3 products (hh, hl, lh), lo unscaled      mode 1: 4 products (+ ll), lo unscaled
// mode 2: 3 products, lo scaled by 2^11 into a second accumulator      mode 3: bf16 six-term (reference point)
__global__ void probe(const float *A, const float *B, float *Zf32, float *Zx, int mode) {
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  f32x16 c = {0};
  for (int s = 0; s < K / 2; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + 2 * s + h], B[l31 * K + 2 * s + h], c, 0, 0, 0);
  f32x16 c3 = {0}, c2 = {0};
  for (int s = 0; s < K / 16; ++s) {
    if (mode == 3) {
      bf16x8 ah, am, al, bh, bm, bl;
      for (int q = 0; q < 8; ++q) {
        float a = A[l31 * K + 16 * s + 8 * h + q], b = B[l31 * K + 16 * s + 8 * h + q];
        ah[q] = (__bf16)a; float r = a - (float)ah[q]; am[q] = (__bf16)r; al[q] = (__bf16)(r - (float)am[q]);
        bh[q] = (__bf16)b; r = b - (float)bh[q]; bm[q] = (__bf16)r; bl[q] = (__bf16)(r - (float)bm[q]);
      }
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c3, 0, 0, 0);
      continue;
    }
    f16x8 ah, al, bh, bl;
    const float sc = mode == 2 ? 2048.f : 1.f;
    for (int q = 0; q < 8; ++q) {
      const float a = A[l31 * K + 16 * s + 8 * h + q], b = B[l31 * K + 16 * s + 8 * h + q];
      ah[q] = (_Float16)a; al[q] = (_Float16)((a - (float)ah[q]) * sc);
      bh[q] = (_Float16)b; bl[q] = (_Float16)((b - (float)bh[q]) * sc);
    }
    if (mode == 2) {
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c2, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c2, 0, 0, 0);
    } else {
      if (mode == 1) c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c3, 0, 0, 0);
    }
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
    Zf32[m * 32 + l31] = c[r];
    Zx[m * 32 + l31] = c3[r] + c2[r] * (1.f / 2048.f);
  }
}

// does the f16 MFMA flush subnormal inputs?  A = 2^-20 (an f16 subnormal), B = 2^10 in k = 0, zeros elsewhere: Z = 2^-10 if not
__global__ void flush_probe(float *out) {
  const int lane = threadIdx.x;
  f16x8 a = {0}, b = {0};
  if ((lane >> 5) == 0) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.f; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}

int main() {
  float *dA, *dB, *dZ;
  hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dZ, 2 * 1024 * 4);
  flush_probe<<<1, 64>>>(dZ);
  float fz = 0;
  hipMemcpy(&fz, dZ, 4, hipMemcpyDeviceToHost);
  printf("subnormal f16 input 2^-20 x 2^10 through v_mfma_f32_32x32x16_f16: %.6g (expected 2^-10 = %.6g; 0 = flushed)\n", fz, 1.0 / 1024);
  const char *names[4] = {"f16 x2, 3 products", "f16 x2, 4 products", "f16 x2, 3 products, lo * 2^11", "bf16 x3, 6 products"};
  for (float scale : {1.f, 0.1f, 0.01f, 1e-3f, 100.f}) {
    std::vector<float> A(32 * K), B(32 * K);
    srand(7);
    for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f * scale;
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
      probe<<<1, 64>>>(dA, dB, dZ, dZ + 1024, mode);
      std::vector<float> Z(2 * 1024);
      hipMemcpy(Z.data(), dZ, Z.size() * 4, hipMemcpyDeviceToHost);
      double e32 = 0, ex = 0, zmax = 0;
      for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
          double ref = 0;
          for (int k = 0; k < K; ++k) ref += (double)A[m * K + k] * (double)B[n * K + k];
          zmax = fmax(zmax, fabs(ref));
          e32 = fmax(e32, fabs(Z[m * 32 + n] - ref));
          ex = fmax(ex, fabs(Z[1024 + m * 32 + n] - ref));
        }
      printf("scale(A) %-6g K=%d max|Z|=%.3g: f32 MFMA %.2e   %-32s %.2e\n", scale, K, zmax, e32 / zmax, names[mode], ex / zmax);
    }
  }
  return 0;
}
