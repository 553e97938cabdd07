// rampbench.hip — what sets the dispatch ramp of a one-wave-per-env launch?  Each wave stamps its entry time
// (s_memrealtime, 100 MHz); variants differ in LDS per workgroup, VGPR allocation and workgroup size.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_rampbench tools/rampbench.hip && ./gpurun_rampbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int VREGS, int WG>
__global__ __launch_bounds__(WG) void ramp_kernel(uint64_t* out, int n_waves) {
  const uint64_t t = __builtin_amdgcn_s_memrealtime();
  __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
  if (LDS_BYTES > 0) lds[threadIdx.x] = (char)t;
  // hold VREGS VGPRs alive
  float acc[VREGS > 0 ? VREGS : 1];
#pragma unroll
  for (int i = 0; i < VREGS; i++) acc[i] = (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < VREGS; i++) asm volatile("" : "+v"(acc[i]));
  float s = 0;
#pragma unroll
  for (int i = 0; i < VREGS; i++) s += acc[i];
  const int wave = (blockIdx.x * WG + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0 && wave < n_waves) out[wave] = t + (s == 12345.f ? 1 : 0) + (LDS_BYTES > 0 ? (lds[0] == 77 && s == 3.f) : 0);
}

template <int LDS_BYTES, int VREGS, int WG>
static void run(const char* name, uint64_t* d, int n_waves) {
  std::vector<uint64_t> h(n_waves);
  const int nb = n_waves * 64 / WG;
  double spread = 0, p50 = 0, p90 = 0;
  const int R = 20;
  for (int r = 0; r < R + 3; r++) {
    hipLaunchKernelGGL((ramp_kernel<LDS_BYTES, VREGS, WG>), dim3(nb), dim3(WG), 0, 0, d, n_waves);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, n_waves * 8, hipMemcpyDeviceToHost);
    if (r < 3) continue;
    std::sort(h.begin(), h.end());
    spread += (h.back() - h.front()) / 100.0;
    p50 += (h[n_waves / 2] - h.front()) / 100.0;
    p90 += (h[n_waves * 9 / 10] - h.front()) / 100.0;
  }
  printf("%-44s waves %5d: start p50 %.2f  p90 %.2f  last %.2f us\n", name, n_waves, p50 / R, p90 / R, spread / R);
}

int main() {
  uint64_t* d;
  hipMalloc((void**)&d, 65536 * 8);
  for (int n : {8192, 4096, 2048}) {
    run<0, 0, 256>("WG256 lds0 vgpr~8", d, n);
    run<10240, 0, 256>("WG256 lds10K vgpr~8", d, n);
    run<8192, 0, 256>("WG256 lds8K vgpr~8", d, n);
    run<0, 40, 256>("WG256 lds0 vgpr~48", d, n);
    run<10240, 40, 256>("WG256 lds10K vgpr~48", d, n);
    run<10240, 40, 64>("WG64  lds10K vgpr~48", d, n);
    run<2560, 40, 64>("WG64  lds2.5K vgpr~48", d, n);
    run<10240, 40, 1024>("WG1024 lds10K vgpr~48", d, n);
    run<18432, 40, 512>("WG512 lds18K vgpr~48", d, n);
    run<0, 0, 512>("WG512 lds0 vgpr~8", d, n);
    run<0, 40, 64>("WG64  lds0 vgpr~48", d, n);
    run<0, 100, 256>("WG256 lds0 vgpr~104", d, n);
  }
  return 0;
}
