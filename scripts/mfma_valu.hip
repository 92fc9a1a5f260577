// Microbenchmark: does VALU / LDS work issued between v_mfma_f32_32x32x2_f32 instructions run in the shadow of the
// matrix pipe or add to it?  (development aid; build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int L, int CH, int MODE = 0>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
  __shared__ float lds[8192];
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0001f;
  float x[16];
  for (int q = 0; q < 16; ++q) x[q] = a + q;
  lds[threadIdx.x] = a; lds[threadIdx.x + 512] = b;
  __syncthreads();
  float lv = 0.f;
  const unsigned la = (threadIdx.x & 63) * 4, la4 = (threadIdx.x & 63) * 16;
  typedef float f2 __attribute__((ext_vector_type(2))); typedef float f4 __attribute__((ext_vector_type(4)));
  f4 lq[4] = {}; f2 l2 = {};
  f2 xp[8], bp = {b, b}; for (int q = 0; q < 8; ++q) xp[q] = f2{a, a + q};
  unsigned xi[9]; for (int q = 0; q < 9; ++q) xi[q] = threadIdx.x + q; unsigned sc = 0;
  const float *gp = out + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c % CH], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(c * V + v) & 15]) : "v"(b), "v"(a));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[(c * V + v) & 7]) : "v"(bp), "v"(bp));
        if (MODE == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(x[(c * V + v) & 15]) : "v"(b));
        if (MODE == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
        if (MODE == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xi[(c * V + v) & 7]) : "v"(xi[8]));
      }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        if (MODE == 0) { float t; asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(t) : "v"(la)); lq[l & 3].x = t; }
        if (MODE == 5) { asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(lq[l & 3]) : "v"(la4)); }
        if (MODE == 6) { asm volatile("ds_write_b32 %0, %1 offset:8192" :: "v"(la), "v"(a)); }
        if (MODE == 7) { asm volatile("ds_write_b128 %0, %1 offset:8192" :: "v"(la4), "v"(lq[0])); }
        if (MODE == 8) { asm volatile("ds_read2_b32 %0, %1 offset0:16 offset1:80" : "=v"(l2) : "v"(la)); }
        if (MODE == 9) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(lq[l & 3]) : "v"(gp)); }
      }
      if (L) asm volatile("s_waitcnt lgkmcnt(8)");
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  float s = lv + l2.x + l2.y + sc; for (int q = 0; q < 4; ++q) s += lq[q].x + lq[q].y + lq[q].z + lq[q].w; for (int q = 0; q < 8; ++q) s += xp[q].x + xp[q].y + xi[q];
  for (int q = 0; q < 16; ++q) s += x[q];
  for (int c = 0; c < CH; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int L, int CH, int MODE = 0>
void run(const char *name, int threads) {
  float *out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<V, L, CH, MODE><<<256, threads>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<V, L, CH, MODE><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 4 * (threads / 64) / 4.0;
  printf("%-28s threads %3d: %.3f ms, %.1f ns/MFMA/SIMD (64 cyc @2.4GHz = 26.7 ns), %.1f TF/s\n", name, threads, ms,
         ms * 1e6 / mfma_per_simd, 256.0 * threads / 64 * iters * 4 * 4096.0 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int th : {512}) {
    run<0, 0, 4>("no extra", th);
    run<8, 0, 4, 0>("8 v_fma_f32", th);
    run<8, 0, 4, 1>("8 v_pk_fma_f32", th);
    run<8, 0, 4, 2>("8 v_mov_b32", th);
    run<8, 0, 4, 4>("8 v_add_u32", th);
    run<8, 0, 4, 3>("8 s_add_u32", th);
    run<0, 2, 4, 0>("2 ds_read_b32", th);
    run<0, 2, 4, 8>("2 ds_read2_b32", th);
    run<0, 2, 4, 5>("2 ds_read_b128", th);
    run<0, 1, 4, 5>("1 ds_read_b128", th);
    run<0, 2, 4, 6>("2 ds_write_b32", th);
    run<0, 1, 4, 7>("1 ds_write_b128", th);
    run<0, 1, 4, 9>("1 global_load_dwordx4 (L2 hit)", th);
  }
  return 0;
}
