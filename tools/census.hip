// Developer tool: where does the dispatcher put N workgroups of a given shape?
// hipcc --offload-arch=gfx950 -O2 tools/census.hip -o /tmp/census && /tmp/census 525 36864 120
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void census(unsigned *out, int spin) {
  extern __shared__ char smem[];
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = clock64();
  volatile char *p = smem;
  p[threadIdx.x] = 1;
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hwid;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

int main(int argc, char **argv) {
  int n = argc > 1 ? atoi(argv[1]) : 525;
  int lds = argc > 2 ? atoi(argv[2]) : 36864;
  int spin = argc > 3 ? atoi(argv[3]) : 100000;
  unsigned *d;
  hipMalloc(&d, n * 8);
  hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  census<<<n, 256, lds>>>(d, spin);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  for (int i = 0; i < n; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
  }
  std::map<int, int> hist;
  for (auto &kv : per_cu) hist[kv.second]++;
  printf("WGs %d lds %d: distinct CUs used %zu; WGs-per-CU histogram:", n, lds, per_cu.size());
  for (auto &kv : hist) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
