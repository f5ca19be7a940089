// Does hipExtAnyOrderLaunch clear the barrier bit on gfx950 (ROCm 7.2)?  Two ~60 us single-workgroup kernels on ONE stream:
// ordered they take ~120 us, any-order ~60 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin_k(long cycles, int* out) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out) out[blockIdx.x] = 1;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  int* d; hipMalloc(&d, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long cyc = 6000;   // wall_clock64 runs at 100 MHz: 60 us
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      for (int i = 0; i < 8; ++i) {
        void* args[] = {(void*)&cyc, (void*)&d};
        const int flags = (mode == 1 && (i & 1)) || (mode == 2 && i > 0) ? hipExtAnyOrderLaunch : 0;
        hipError_t r = hipExtLaunchKernel((const void*)spin_k, dim3(1), dim3(64), args, 0, s, nullptr, nullptr, flags);
        if (r != hipSuccess) { printf("launch error %d\n", (int)r); return 1; }
      }
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): 8 x 60 us kernels took %.1f us\n", mode, mode == 0 ? "ordered" : mode == 1 ? "odd launches any-order" : "all but the first any-order", ms * 1e3);
    }
  }
  return 0;
}
