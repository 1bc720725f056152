// Host cost of a kernel launch on this box: raw hipLaunchKernelGGL enqueue rate (empty kernel, 12 arguments),
// the same through a captured hipGraph, and the device-side cadence of back-to-back tiny kernels.
//   hipcc --offload-arch=gfx950 -O2 -o tools/launch_cost_bin tools/launch_cost.hip && tools/launch_cost_bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <ctime>
#include <unistd.h>
static double cpu() { return (double)clock() / CLOCKS_PER_SEC; }
__global__ void tiny(float *p, int a, int b, int c, int d, int e, int f, int g, int h, int i, int j, float s) {
  if (threadIdx.x == 0 && a < 0) p[0] = s + b + c + d + e + f + g + h + i + j;
}
__global__ void spin(long long ticks) {      // wall_clock64 counts at 100 MHz
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float *p;
  hipMalloc(&p, 1024);
  hipStream_t st;
  hipStreamCreate(&st);
  const int N = 4000;
  for (int rep = 0; rep < 3; ++rep) {
    hipStreamSynchronize(st);
    double c0 = cpu();
    double t0 = now();
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, p, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 1.f);
    double t1 = now();
    hipStreamSynchronize(st);
    double t2 = now();
    printf("eager: %d launches enqueued in %.2f us each; drained after %.2f us each (device cadence); process CPU %.2f us per launch\n", N,
           (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, (cpu() - c0) / N * 1e6);
  }
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, p, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 1.f);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int rep = 0; rep < 3; ++rep) {
    hipStreamSynchronize(st);
    double c0 = cpu();
    double t0 = now();
    hipGraphLaunch(ge, st);
    double t1 = now();
    hipStreamSynchronize(st);
    double t2 = now();
    printf("graph: %d nodes launched in %.2f us each (host); done after %.2f us each; process CPU %.2f us per node\n", N, (t1 - t0) / N * 1e6,
           (t2 - t0) / N * 1e6, (cpu() - c0) / N * 1e6);
  }
  // a graph launched behind 20 ms of queued work: does anybody burn CPU while it waits?
  for (int rep = 0; rep < 2; ++rep) {
    hipStreamSynchronize(st);
    double c0 = cpu(), t0 = now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 2000000LL);
    double t1 = now();
    hipGraphLaunch(ge, st);
    double t2 = now();
    hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, p, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 1.f);
    double t3 = now();
    double c1 = cpu();
    hipStreamSynchronize(st);
    double t4 = now();
    printf("graph behind a 20 ms kernel: spin launch %.1f us, hipGraphLaunch %.1f us, next eager launch %.1f us, CPU until all issued %.2f ms, "
           "wall until done %.2f ms, process CPU until done %.2f ms\n", (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (c1 - c0) * 1e3, (t4 - t0) * 1e3, (cpu() - c0) * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep) {      // the main thread SLEEPS while the device works: CPU burnt meanwhile is other threads'
    hipStreamSynchronize(st);
    double c0 = cpu();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 2000000LL);
    hipGraphLaunch(ge, st);
    double c1 = cpu();
    usleep(40000);
    double c2 = cpu();
    hipStreamSynchronize(st);
    printf("graph behind a 20 ms kernel, main thread asleep for 40 ms: CPU to issue %.2f ms, process CPU during the sleep %.2f ms\n", (c1 - c0) * 1e3, (c2 - c1) * 1e3);
    c0 = cpu();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 2000000LL);
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, p, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 1.f);
    c1 = cpu();
    usleep(40000);
    c2 = cpu();
    hipStreamSynchronize(st);
    printf("eager behind a 20 ms kernel, main thread asleep for 40 ms: CPU to issue %.2f ms, process CPU during the sleep %.2f ms\n", (c1 - c0) * 1e3, (c2 - c1) * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipStreamSynchronize(st);
    double c0 = cpu(), t0 = now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 2000000LL);
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st, p, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 1.f);
    double t3 = now();
    double c1 = cpu();
    hipStreamSynchronize(st);
    double t4 = now();
    printf("eager behind a 20 ms kernel: %d launches issued in %.2f ms, CPU until all issued %.2f ms, wall until done %.2f ms, process CPU until done %.2f ms\n",
           N, (t3 - t0) * 1e3, (c1 - c0) * 1e3, (t4 - t0) * 1e3, (cpu() - c0) * 1e3);
  }
  return 0;
}
