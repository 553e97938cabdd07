// wcalib.hip — known-byte-count kernels in the step kernel's own access pattern, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE
// (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//   calib_write<sc1|plain>   every wave stores `planes` x 1024 B (16 B per lane, the step kernel's inline-asm store form)
//   calib_read               every wave loads  `planes` x 1024 B (16 B per lane) and folds them into one dword per wave
// Each kernel runs over N "envs" (one wave per env, 8 waves per workgroup, XCD-contiguous block map like arcle_step_kernel).
// usage: wcalib [N] [planes] [reps]      -> prints the exact byte count per launch of every kernel; run under
//        rocprofv3 --pmc WRITE_SIZE (and, separately, --pmc FETCH_SIZE) --kernel-trace and divide.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
#define AS_GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ int wave_of_launch() {
  const uint32_t b = blockIdx.x, nb8 = gridDim.x >> 3;
  return (int)(((b & 7u) * nb8 + (b >> 3)) * 8u + (threadIdx.x >> 6));
}
template <int SC1>
__global__ __launch_bounds__(512) void calib_write(int8_t* base, int planes, int n, uint32_t seed) {
  const int env = wave_of_launch();
  if (env >= n) return;
  const uint32_t lane = threadIdx.x & 63;
  U4 v = {seed + lane, seed ^ (uint32_t)env, lane, seed};
  for (int p = 0; p < planes; p++) {
    int8_t* pl = base + (size_t)p * n * 1024;
    const uint32_t off = (uint32_t)env * 1024u + 16u * lane;
    if (SC1) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(pl) : "memory");
    else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(off), "v"(v), "s"(pl) : "memory");
    v[0] += 1;
  }
}
__global__ __launch_bounds__(512) void calib_read(const int8_t* base, int planes, int n, uint32_t* sink) {
  const int env = wave_of_launch();
  if (env >= n) return;
  const uint32_t lane = threadIdx.x & 63;
  uint32_t acc = 0;
  for (int p = 0; p < planes; p++) {
    const U4 v = *reinterpret_cast<const AS_GLOBAL U4*>((uintptr_t)(base + (size_t)p * n * 1024) + (uint32_t)env * 1024u + 16u * lane);
    acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) sink[env] = acc;  // (never true for the data written above: keeps the loads alive)
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192, planes = argc > 2 ? atoi(argv[2]) : 4, reps = argc > 3 ? atoi(argv[3]) : 20;
  int8_t* buf;
  uint32_t* sink;
  const size_t bytes = (size_t)planes * n * 1024;
  if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, (size_t)n * 4) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes);
  const dim3 g((unsigned)(((n + 7) / 8 + 7) & ~7)), b(512);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(calib_write<1>, g, b, 0, 0, buf, planes, n, (uint32_t)r);
  hipDeviceSynchronize();
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(calib_write<0>, g, b, 0, 0, buf, planes, n, (uint32_t)r + 100u);
  hipDeviceSynchronize();
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(calib_read, g, b, 0, 0, (const int8_t*)buf, planes, n, sink);
  hipDeviceSynchronize();
  printf("wcalib N=%d planes=%d reps=%d: every launch of calib_write<1> (sc1), calib_write<0> (plain) and calib_read moves exactly %zu bytes\n",
         n, planes, reps, bytes);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
