// Microbenchmark: HBM read bandwidth of the data pass's D access pattern against a plain linear stream (development aid).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_pattern scripts/hbm_pattern.hip && /tmp/hbm_pattern
// Tile-major D as the library lays it out: tile (rb, cb) at ((cb * nRB) + rb) * 4 KiB; a workgroup of NW waves owns a row
// panel (NW consecutive row blocks) and walks column tiles: NW * 4 KiB contiguous per step, then a stride of nRB * 4 KiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

// mode 0: linear grid-stride stream; mode 1: panel walk (persistent grid, contiguous ranges of (panel, tile) work like the kernel)
template <int NW, int MODE, int DEPTH>
__global__ __launch_bounds__(64 * NW) void k(const f4 *D, long n16, long nRB, long nCT, float *out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride * 4) {
      f4 v0 = __builtin_nontemporal_load(D + i), v1 = {0, 0, 0, 0}, v2 = v1, v3 = v1;
      if (i + stride < n16) v1 = __builtin_nontemporal_load(D + i + stride);
      if (i + 2 * stride < n16) v2 = __builtin_nontemporal_load(D + i + 2 * stride);
      if (i + 3 * stride < n16) v3 = __builtin_nontemporal_load(D + i + 3 * stride);
      acc += v0 + v1 + v2 + v3;
    }
  } else {
    const long n_rp = nRB / NW, total = n_rp * nCT;
    const long b = total * blockIdx.x / gridDim.x, e = total * (blockIdx.x + 1) / gridDim.x;
    // work order: panel-major within the range (tiles of one panel consecutive), DEPTH tiles in flight per wave
    for (long t = b; t < e; t += DEPTH) {
      f4 v[DEPTH][4];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const long tt = t + d < e ? t + d : e - 1;
        const long rp = tt / nCT, ct = tt - rp * nCT;
        const f4 *p = D + ((ct * nRB) + rp * NW + w) * 256 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[d][q] = __builtin_nontemporal_load(p + 64 * q);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += v[d][q];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}

template <int NW, int MODE, int DEPTH>
void run(const char *name, const f4 *D, long nRB, long nCT, float *out, int grid) {
  const long n16 = nRB * nCT * 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NW, MODE, DEPTH><<<grid, 64 * NW>>>(D, n16, nRB, nCT, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k<NW, MODE, DEPTH><<<grid, 64 * NW>>>(D, n16, nRB, nCT, out);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  printf("%-44s grid %5d: %7.3f ms  %7.1f GB/s\n", name, grid, ms, n16 * 16.0 / (ms * 1e-3) / 1e9);
}

int main() {
  const long M = 200000, N = 50000;
  const long nRB = (M + 255) / 256 * 8, nCT = (N + 63) / 64 * 2;   // row blocks padded to whole 256-row panels
  const long bytes = nRB * nCT * 4096;
  f4 *D; float *out;
  if (hipMalloc(&D, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 4);
  hipMemset(D, 0, bytes);
  printf("D: %ld row blocks x %ld column tiles = %.1f GB\n", nRB, nCT, bytes / 1e9);
  run<4, 0, 1>("linear stream, 256 threads", D, nRB, nCT, out, 256 * 8);
  run<4, 0, 1>("linear stream, 256 threads", D, nRB, nCT, out, 256 * 32);
  run<8, 1, 1>("panel walk, 8 waves, 1 tile in flight", D, nRB, nCT, out, 256);
  run<8, 1, 2>("panel walk, 8 waves, 2 tiles in flight", D, nRB, nCT, out, 256);
  run<8, 1, 4>("panel walk, 8 waves, 4 tiles in flight", D, nRB, nCT, out, 256);
  run<4, 1, 2>("panel walk, 4 waves (128 rows), 2 in flight", D, nRB, nCT, out, 256);
  run<4, 1, 4>("panel walk, 4 waves (128 rows), 4 in flight", D, nRB, nCT, out, 256);
  run<8, 1, 2>("panel walk, 8 waves, 2 in flight, 2 WG/CU", D, nRB, nCT, out, 512);
  return 0;
}
