// What does a cross-stream dependency cost the PRODUCER queue on MI355X?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o event_gap_probe event_gap_probe.hip && ./event_gap_probe
// The engine's backward issues, per conv, a data gradient on the main stream, then makes the weight-gradient stream
// wait for it (hipEventRecord on main + hipStreamWaitEvent on the side stream), then launches the BatchNorm passes on
// main.  The rocprofv3 timeline shows 6-7 us of idle main queue after every such record.  This probe times N
// repetitions of   main: K (writes 64 MB), [dependency], K     side: small kernel behind the dependency
// with the dependency made four ways:
//   0  none (side stream free-running)                       -- the floor
//   1  hipEventRecord(main) + hipStreamWaitEvent(side)       -- what torch's wait_stream does
//   2  hipExtLaunchKernelGGL(K, ..., stopEvent) + hipStreamWaitEvent(side): the event IS the kernel's completion signal
//   3  as 1 with an event created with hipEventDisableTiming | hipEventReleaseToDevice-style flags
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_kernel(float4* out, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) out[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
}
__global__ void tiny_kernel(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }

int main() {
  const size_t n = (64u << 20) / 16;
  float4 *a, *b;
  float* t;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&t, 4096));
  CK(hipMemset(t, 0, 4096));
  hipStream_t mainq, side;
  CK(hipStreamCreateWithFlags(&mainq, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  const int N = 200;
  hipEvent_t ev[N];
  auto run = [&](int mode) {
    for (int i = 0; i < N; ++i) {
      unsigned flags = hipEventDisableTiming;
      if (mode == 3) flags |= hipEventReleaseToDevice;
      CK(hipEventCreateWithFlags(&ev[i], flags));
    }
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (mode == 2) {
          hipExtLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, mainq, nullptr, ev[i], 0, a, n, (float)i);
        } else {
          hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, mainq, a, n, (float)i);
          if (mode == 1 || mode == 3) CK(hipEventRecord(ev[i], mainq));
        }
        if (mode != 0) CK(hipStreamWaitEvent(side, ev[i], 0));
        hipLaunchKernelGGL(tiny_kernel, dim3(4), dim3(64), 0, side, t);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, mainq, b, n, (float)i);
      }
      CK(hipStreamSynchronize(mainq));
      CK(hipStreamSynchronize(side));
      auto t1 = std::chrono::steady_clock::now();
      const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
      best = us < best ? us : best;
    }
    for (int i = 0; i < N; ++i) CK(hipEventDestroy(ev[i]));
    return best;
  };
  const char* names[4] = {"no dependency", "hipEventRecord + hipStreamWaitEvent", "hipExtLaunchKernelGGL stopEvent + hipStreamWaitEvent",
                          "hipEventRecord (DisableTiming|ReleaseToDevice) + wait"};
  double base = 0;
  for (int mode = 0; mode < 4; ++mode) {
    const double us = run(mode);
    if (mode == 0) base = us;
    printf("mode %d  %-58s %8.2f us per (K, dep, K) pair   (+%.2f us vs no dependency)\n", mode, names[mode], us, us - base);
  }
  return 0;
}
