// Microbenchmark: how much VALU / LDS work fits in the shadow of v_mfma_f32_32x32x16_bf16 when ONE wave per SIMD issues
// both (the situation of pmf_fused_sb2_kernel)?  Development aid; build:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bf16_valu scripts/mfma_bf16_valu.hip && /tmp/mfma_bf16_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_cvt_pk_bf16_f32, 3 ds_write_b64, 4 ds_read_b128, 5 ds_write_b128, 6 ds_write_b32,
//       7 global_load_dwordx4
template <int V, int MODE, int CH>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(threadIdx.x * 0.001f + q); b[q] = (__bf16)1.0f; }
  float x[16];
  for (int q = 0; q < 16; ++q) x[q] = threadIdx.x + q;
  f2 xp[8], bp = {1.0001f, 1.0001f};
  for (int q = 0; q < 8; ++q) xp[q] = f2{(float)q, (float)threadIdx.x};
  unsigned pk[8] = {};
  f4 lq[4] = {};
  const unsigned la = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096, la8 = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 4096,
                 la16 = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  const float *gp = out + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c % CH], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int i = c * V + v;
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 15]) : "v"(bp.x), "v"(bp.y));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[i & 7]) : "v"(bp), "v"(bp));
        if (MODE == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i & 7]) : "v"(x[i & 15]), "v"(x[(i + 1) & 15]));
        if (MODE == 3) asm volatile("ds_write_b64 %0, %1" ::"v"(la8), "v"(xp[i & 7]));
        if (MODE == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(lq[i & 3]) : "v"(la16));
        if (MODE == 5) asm volatile("ds_write_b128 %0, %1" ::"v"(la16), "v"(lq[0]));
        if (MODE == 6) asm volatile("ds_write_b32 %0, %1" ::"v"(la), "v"(x[i & 15]));
        if (MODE == 7) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lq[i & 3]) : "v"(gp));
      }
      if (MODE >= 3 && MODE <= 6) asm volatile("s_waitcnt lgkmcnt(6)");
      if (MODE == 7) asm volatile("s_waitcnt vmcnt(6)");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  float s = 0.f;
  for (int q = 0; q < 4; ++q) s += lq[q].x + lq[q].y + lq[q].z + lq[q].w;
  for (int q = 0; q < 8; ++q) s += xp[q].x + xp[q].y + pk[q];
  for (int q = 0; q < 16; ++q) s += x[q];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int MODE, int CH>
void run(const char *name) {
  float *out;
  hipMalloc(&out, 256 * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<V, MODE, CH><<<256, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<V, MODE, CH><<<256, 256>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 4;
  printf("%-34s %7.3f ms  %6.1f ns per MFMA per SIMD (8 passes = 32 cyc = 13.3 ns at 2.4 GHz)  %7.1f TF/s\n", name, ms, ms * 1e6 / n,
         256.0 * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<0, 0, 4>("MFMA only, 4 chains");
  run<0, 0, 1>("MFMA only, 1 chain (dependent)");
  run<2, 0, 4>("+2 v_fma_f32 per MFMA");
  run<4, 0, 4>("+4 v_fma_f32");
  run<6, 0, 4>("+6 v_fma_f32");
  run<8, 0, 4>("+8 v_fma_f32");
  run<12, 0, 4>("+12 v_fma_f32");
  run<4, 1, 4>("+4 v_pk_fma_f32");
  run<8, 1, 4>("+8 v_pk_fma_f32");
  run<4, 2, 4>("+4 v_cvt_pk_bf16_f32");
  run<8, 2, 4>("+8 v_cvt_pk_bf16_f32");
  run<1, 3, 4>("+1 ds_write_b64");
  run<2, 3, 4>("+2 ds_write_b64");
  run<1, 4, 4>("+1 ds_read_b128");
  run<2, 4, 4>("+2 ds_read_b128");
  run<1, 5, 4>("+1 ds_write_b128");
  run<2, 6, 4>("+2 ds_write_b32");
  run<4, 6, 4>("+4 ds_write_b32");
  run<1, 7, 4>("+1 global_load_dwordx4 (L2 hit)");
  return 0;
}
