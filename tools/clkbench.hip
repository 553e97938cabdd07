// clkbench: what does one instruction cost when W waves share a SIMD?  Every CU runs W waves per SIMD, each executing an unrolled
// chain of one instruction kind; elapsed time from s_memrealtime (100 MHz) inside the kernel.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o clkbench tools/clkbench.hip && ./clkbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
// MODE 0: dependent v_add_u32   1: dependent s_add_u32   2: alternating s_add_u32 / v_add_u32   3: s_add_u32 + s_cmp + s_cselect (typical
// scalar mix)   4: v_add_u32 then v_readfirstlane (VALU->SALU hop)   5: s_lshl_b64/s_and_b64 (64-bit scalar)
template <int MODE>
__global__ __launch_bounds__(256) void chain(uint32_t* out, uint64_t* ticks, int iters) {
  uint32_t a = threadIdx.x, b = blockIdx.x + 1;
  uint32_t s = __builtin_amdgcn_readfirstlane(b), s2 = s + 1;
  uint64_t q = s;
  const uint64_t r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 100; k++) {
      if (MODE == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
      if (MODE == 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
      if (MODE == 2) { if (k & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); else asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc"); }
      if (MODE == 3) { if (k % 3 == 0) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc"); else if (k % 3 == 1) asm volatile("s_cmp_lt_u32 %0, %1" : : "s"(s), "s"(s2) : "scc"); else asm volatile("s_cselect_b32 %0, %1, %0" : "+s"(s2) : "s"(s) : "scc"); }
      if (MODE == 4) { if (k & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "s"(s)); else asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(a)); }
      if (MODE == 5) asm volatile("s_lshl_b64 %0, %0, 1" : "+s"(q) : : "scc");
    }
  }
  const uint64_t r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + s + s2 + (uint32_t)q;
  if (threadIdx.x == 0) ticks[blockIdx.x] = r1 - r0;
}
template <int MODE>
static void run(const char* what, uint32_t* out, uint64_t* ticks) {
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD (256-thread workgroups: one wave per SIMD each)
    const int blocks = 256 * wps, iters = 100;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), ticks, 8 * blocks, hipMemcpyDeviceToHost);
    double rt = 0;
    for (int i = 0; i < blocks; i++) rt += h[i];
    rt /= blocks;
    const double n = 100.0 * iters;
    printf("%-44s waves/SIMD %d: %6.2f ns per instruction per wave, %5.2f ns per instruction per SIMD\n", what, wps, rt * 10.0 / n, rt * 10.0 / n / wps);
  }
}
int main() {
  uint32_t* out;
  uint64_t* ticks;
  hipMalloc(&out, 8192 * 64 * 4 * 4);
  hipMalloc(&ticks, 8 * 65536);
  run<0>("dependent v_add_u32", out, ticks);
  run<1>("dependent s_add_u32", out, ticks);
  run<2>("alternating s_add_u32 / v_add_u32", out, ticks);
  run<3>("s_add / s_cmp / s_cselect", out, ticks);
  run<4>("v_add_u32 <-> v_readfirstlane", out, ticks);
  run<5>("dependent s_lshl_b64", out, ticks);
  return 0;
}
