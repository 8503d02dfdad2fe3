// A kernel that holds `blocks` CU slots for `ticks` of the 100 MHz wall clock with few registers and no LDS: what an
// RCCL all-reduce kernel looks like to the one-block-per-CU MFMA kernels of the training step (a whole-register-file
// block cannot join it on its CU).  scratch/contention_probe.py launches it on a third stream during backward.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcu_hold.so cu_hold.hip
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void hold_kernel(long long ticks) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) { }
}
extern "C" int cu_hold(int blocks, long long ticks, void* stream) {
  hold_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(ticks);
  return (int)hipGetLastError();
}
