// Microbenchmark: LDS float atomic add throughput (ds_add_f32) vs ds_write_b32 / ds_read_b128
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned seed) {
  extern __shared__ float lds[];   // 128 KB
  const int n = 32768;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7, grp = lane >> 3;
  unsigned s = seed + threadIdx.x / 8 * 2654435761u + blockIdx.x * 40503u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const int row = (s >> 10) & 255;          // 256 rows x 32 floats... use 1024 rows: n/32
    const int r2 = ((s >> 8) & 1023);
    float* p = lds + r2 * 32 + sub * 4;
    if (MODE == 0) {            // 4 atomic adds per lane (128 B per 8-lane group)
      atomicAdd(p + 0, 1.f); atomicAdd(p + 1, 1.f); atomicAdd(p + 2, 1.f); atomicAdd(p + 3, 1.f);
    } else if (MODE == 1) {     // float4 read
      float4 v = *reinterpret_cast<float4*>(p);
      acc += v.x + v.y + v.z + v.w;
    } else if (MODE == 2) {     // 4 plain dword writes
      p[0] = acc; p[1] = acc; p[2] = acc; p[3] = acc + row;
    } else {                    // float4 write
      *reinterpret_cast<float4*>(p) = make_float4(acc, acc, acc, acc + row);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc + lds[seed & 1023];
}
int main() {
  float* out; hipMalloc(&out, 4096);
  const int iters = 20000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"ds_add_f32 x4", "ds_read_b128", "ds_write_b32 x4", "ds_write_b128"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 131072, 0, out, iters, 1u);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 131072, 0, out, iters, 1u);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 131072, 0, out, iters, 1u);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 131072, 0, out, iters, 1u);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 1024 threads * iters * 16 B
    double bytes_per_cu = 1024.0 * iters * 16;
    double gbps_cu = bytes_per_cu / (ms * 1e-3) / 1e9;
    printf("%-16s %.3f ms  %.1f GB/s per CU  (%.1f B/clk at 2.4GHz)  %.2f ns per 128-B row op per CU\n",
           names[mode], ms, gbps_cu, gbps_cu / 2.4, ms * 1e6 / (1024.0 / 8 * iters));
  }
  return 0;
}
