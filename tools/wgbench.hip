// wgbench.hip — what does it cost to DISPATCH one workgroup per env?  A kernel that does next to nothing (one dword in, one out per
// workgroup) launched as N workgroups of T threads with L bytes of dynamic LDS, back to back on one stream: us per launch.  The big-grid step
// kernels launch 16 384 workgroups of 64 … 256 threads; this is the floor under them (profiles/round6_experiments.txt §3).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_wgbench tools/wgbench.hip && ./gpurun_wgbench
#include <hip/hip_runtime.h>
#include <cstdio>

extern __shared__ int dyn_lds[];
__global__ __launch_bounds__(1024) void wg_kernel(const int* in, int* out, int envs_per_wg) {
  // (envs_per_wg > 1: every wavefront group of the workgroup stands for one env)
  const int t = threadIdx.x;
  if (t == 0) dyn_lds[0] = in[blockIdx.x];
  __syncthreads();
  if ((t & (blockDim.x / envs_per_wg - 1)) == 0) out[blockIdx.x * envs_per_wg + t / (blockDim.x / envs_per_wg)] = dyn_lds[0] + t;
}

static float run(int n_wg, int threads, int lds, int epw, int* d_in, int* d_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int R = 200;
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(wg_kernel, dim3(n_wg), dim3(threads), lds, 0, d_in, d_out, epw);
  hipEventRecord(e0, 0);
  for (int i = 0; i < R; i++) hipLaunchKernelGGL(wg_kernel, dim3(n_wg), dim3(threads), lds, 0, d_in, d_out, epw);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / R;
}

int main() {
  int *d_in, *d_out;
  hipMalloc((void**)&d_in, 1 << 20);
  hipMalloc((void**)&d_out, 1 << 20);
  hipMemset(d_in, 0, 1 << 20);
  const int envs = 16384;
  printf("envs %d: us per launch (N workgroups x T threads, L bytes of LDS per workgroup)\n", envs);
  const int lds_per_env[] = {0, 6976, 16704};
  for (int l : lds_per_env)
    for (int tpe : {64, 128, 256})        // threads per env
      for (int epw : {1, 2, 4, 8}) {      // envs per workgroup
        const int T = tpe * epw;
        if (T > 1024 || l * epw > 65536) continue;
        printf("  lds/env %5d  threads/env %3d  envs/wg %d  (N %5d x T %4d, L %5d): %6.2f us\n", l, tpe, epw, envs / epw, T, l * epw, run(envs / epw, T, l * epw, epw, d_in, d_out));
      }
  return 0;
}
