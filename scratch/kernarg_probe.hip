// Does a kernel take more than 4 KB of arguments on this stack?  (6 KB struct by value)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[1536]; };
__global__ void k(Big b, int* out) { if (threadIdx.x == 0) out[0] = b.v[0] + b.v[1535]; }
int main() {
  Big b; for (int i = 0; i < 1536; ++i) b.v[i] = i;
  int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, b, d);
  hipError_t e = hipDeviceSynchronize();
  int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  printf("launch: %s, result %d (expect 1535)\n", hipGetErrorString(e), h);
  return 0;
}
