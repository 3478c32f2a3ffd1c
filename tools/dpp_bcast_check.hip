#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../spmf_amd/csrc/common.h"
using namespace spmf;
template <int N> __global__ void k(int* out) {
  int lane = threadIdx.x;
  for (int j = 0; j < N; ++j) out[(blockIdx.x*8 + j) * 64 + lane] = group_bcast<N>(lane * 10, j);
}
int main(){ int* d; hipMalloc(&d, 4*8*64*4); int h[4*8*64];
  hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, d);
  hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d + 8*64);
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d + 16*64);
  hipLaunchKernelGGL(k<16>, dim3(1), dim3(64), 0, 0, d + 24*64);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0; int Ns[4] = {8,4,2,16};
  for (int t = 0; t < 4; ++t) { int N = Ns[t]; for (int j = 0; j < (N>8?8:N); ++j) for (int l = 0; l < 64; ++l) {
    int want = ((l / N) * N + j) * 10; if (h[(t*8+j)*64+l] != want) { if (bad < 10) printf("N=%d j=%d lane=%d got %d want %d\n", N, j, l, h[(t*8+j)*64+l], want); ++bad; } } }
  printf("bad=%d\n", bad); return bad != 0; }
