// alignbench.hip — what does a misaligned 16-byte-per-lane load cost on gfx950?  Every lane of every wave loads 16 B at
// base + row * stride + 16 * lane + mis (the access pattern of a full-mask payload row, stride 900, and of a flattened state row's plane
// segment, any byte offset), reduces it and writes one dword per wave.  Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_alignbench
// tools/alignbench.hip ; run: ./gpurun_alignbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
typedef U4 __attribute__((aligned(1))) U4a1;
typedef U4 __attribute__((aligned(4))) U4a4;

template <int MODE>  // 0: aligned dwordx4, 1: dword-aligned dwordx4, 2: byte-aligned dwordx4, 3: aligned chunk + neighbour lane + alignbyte
__global__ __launch_bounds__(256) void k(const char* base, uint32_t* out, int rows, int stride, int mis) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= rows) return;
  const char* p = base + (size_t)wave * stride + 16 * lane + mis;
  U4 v;
  if (MODE == 0) v = *reinterpret_cast<const U4*>(p);
  else if (MODE == 1) v = *reinterpret_cast<const U4a4*>(p);
  else if (MODE == 2) v = *reinterpret_cast<const U4a1*>(p);
  else {
    const uintptr_t a = (uintptr_t)p & ~(uintptr_t)15;
    const uint32_t s = (uint32_t)((uintptr_t)p & 15u);  // (wave-uniform)
    const U4 c = *reinterpret_cast<const U4*>(a);
    U4 n;
    for (int i = 0; i < 4; i++) n[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c[i], 0x130, 0xf, 0xf, true);  // lane + 1's chunk
    uint32_t d[8] = {c[0], c[1], c[2], c[3], n[0], n[1], n[2], n[3]};
    const uint32_t q = __builtin_amdgcn_readfirstlane(s >> 2), r = s & 3u;
    uint32_t e[5];
    switch (q) {
      case 0: for (int i = 0; i < 5; i++) e[i] = d[i]; break;
      case 1: for (int i = 0; i < 5; i++) e[i] = d[i + 1]; break;
      case 2: for (int i = 0; i < 5; i++) e[i] = d[i + 2]; break;
      default: for (int i = 0; i < 5; i++) e[i] = d[i + 3]; break;
    }
    for (int i = 0; i < 4; i++) v[i] = __builtin_amdgcn_alignbyte(e[i + 1], e[i], r);
  }
  uint32_t x = v[0] ^ v[1] ^ v[2] ^ v[3];
  for (int o = 32; o > 0; o >>= 1) x ^= __shfl_xor(x, o);
  if (lane == 0) out[wave] = x;
}

template <int MODE>
float run(const char* base, uint32_t* out, int rows, int stride, int mis) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = (rows * 64 + 255) / 256;
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, base, out, rows, stride, mis);
  hipEventRecord(a);
  for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, base, out, rows, stride, mis);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 100.f;  // us per launch
}

int main() {
  for (int rows : {8192, 131072}) {
    char* base; uint32_t* out;
    const size_t bytes = (size_t)rows * 1024 + 4096;
    hipMalloc(&base, bytes); hipMemset(base, 1, bytes); hipMalloc(&out, (size_t)rows * 4);
    printf("rows %d (%.1f MB per launch)\n", rows, rows * 1024 / 1e6);
    printf("  stride 1024 mis 0 : aligned %.2f us\n", run<0>(base, out, rows, 1024, 0));
    for (int mis : {4, 8, 12}) printf("  stride 1024 mis %2d: dword-aligned x4 %.2f | byte-aligned type %.2f | aligned + dpp + alignbyte %.2f us\n", mis,
                                      run<1>(base, out, rows, 1024, mis), run<2>(base, out, rows, 1024, mis), run<3>(base, out, rows, 1024, mis));
    for (int mis : {1, 2, 6, 14}) printf("  stride 1024 mis %2d: byte-aligned x4 %.2f | aligned + dpp + alignbyte %.2f us\n", mis, run<2>(base, out, rows, 1024, mis),
                                         run<3>(base, out, rows, 1024, mis));
    printf("  stride  900 mis 0 (mask rows): dword-aligned x4 %.2f | aligned + dpp + alignbyte %.2f us\n", run<1>(base, out, rows, 900, 0), run<3>(base, out, rows, 900, 0));
    printf("  stride  902 mis 0 (row segments): byte-aligned x4 %.2f | aligned + dpp + alignbyte %.2f us\n", run<2>(base, out, rows, 902, 0), run<3>(base, out, rows, 902, 0));
    hipFree(base); hipFree(out);
  }
  return 0;
}
