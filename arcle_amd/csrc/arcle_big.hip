// arcle_big.hip — gfx950 kernels for grids beyond ARCLE_MAX_CELLS (H * W > 1024, H, W <= 127): one WORKGROUP of 256 threads per env.
// The bodies live in arcle_big.h (also compiled by the test emulator); this file supplies the workgroup primitives (namespace bx),
// the __global__ wrappers and the host-side launchers arcle_hip.hip routes big handles to.  Second translation unit of
// libarcle_hip.so (arcle_amd/_lib.py compiles both and links them).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ARCLE_BIG_DEV __device__ __forceinline__
#define ARCLE_BIG_HD __host__ __device__

namespace bx {
ARCLE_BIG_DEV int tid() { return (int)threadIdx.x; }
ARCLE_BIG_DEV int nt() { return (int)blockDim.x; }
ARCLE_BIG_DEV void sync() { __syncthreads(); }
// barrier behind which every store of the workgroup is visible at system scope (the row-tail completion signal follows it)
ARCLE_BIG_DEV void sync_release() {
  __threadfence_system();
  __syncthreads();
}
ARCLE_BIG_DEV void lds_or(int32_t* a, int v) { atomicOr(a, v); }
ARCLE_BIG_DEV void lds_add(int32_t* a, int v) { atomicAdd(a, v); }
ARCLE_BIG_DEV void lds_min(int32_t* a, int v) { atomicMin(a, v); }
ARCLE_BIG_DEV void lds_max(int32_t* a, int v) { atomicMax(a, v); }
ARCLE_BIG_DEV void lds_umax(uint32_t* a, uint32_t v) { atomicMax(a, v); }
ARCLE_BIG_DEV void status_or(uint32_t* g, uint32_t v) { atomicOr(g, v); }
ARCLE_BIG_DEV uint64_t brev64(uint64_t x) { return __brevll(x); }
ARCLE_BIG_DEV void release_store_system(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}  // namespace bx

#include "arcle_big.h"

using arcle_big::BigParams;

#define BIG_THREADS 256

extern __shared__ __attribute__((aligned(16))) int8_t arcle_big_lds[];

__global__ __launch_bounds__(BIG_THREADS) void arcle_big_step_kernel(const BigParams p) {
  arcle_big::step_env(p, (int)blockIdx.x, arcle_big_lds);
}
__global__ __launch_bounds__(BIG_THREADS) void arcle_big_reset_kernel(const BigParams p, int mode) {
  arcle_big::reset_env(p, (int)blockIdx.x, mode, arcle_big_lds);
}
__global__ __launch_bounds__(BIG_THREADS) void arcle_big_rows_kernel(const BigParams p, int mode) {
  arcle_big::rows_env(p, (int)blockIdx.x, mode, arcle_big_lds);
}
__global__ __launch_bounds__(BIG_THREADS) void arcle_big_set_rows_kernel(const BigParams p) {
  arcle_big::set_rows_env(p, (int)blockIdx.x, arcle_big_lds);
}

namespace arcle_big {

// the dynamic LDS of a launch: up to 69 KB (127 x 127) — beyond the 64 KB a kernel gets without asking
template <class K>
static int allow_lds(K kernel, int bytes) {
  if (bytes <= 65536) return 0;
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int launch_step(const BigParams& p, void* stream) {
  const int lds = lds_bytes(p.PS);
  if (int rc = allow_lds(arcle_big_step_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_step_kernel, dim3((unsigned)p.n_envs), dim3(BIG_THREADS), (size_t)lds, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
int launch_reset(const BigParams& p, int mode, void* stream) {
  const int lds = lds_bytes(p.PS);
  if (int rc = allow_lds(arcle_big_reset_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_reset_kernel, dim3((unsigned)p.n_envs), dim3(BIG_THREADS), (size_t)lds, (hipStream_t)stream, p, mode);
  return (int)hipGetLastError();
}
int launch_rows(const BigParams& p, int mode, void* stream) {
  const int lds = lds_bytes(p.PS);
  if (int rc = allow_lds(arcle_big_rows_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_rows_kernel, dim3((unsigned)p.n_envs), dim3(BIG_THREADS), (size_t)lds, (hipStream_t)stream, p, mode);
  return (int)hipGetLastError();
}
int launch_set_rows(const BigParams& p, void* stream) {
  const int lds = lds_bytes(p.PS);
  if (int rc = allow_lds(arcle_big_set_rows_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_set_rows_kernel, dim3((unsigned)p.n_envs), dim3(BIG_THREADS), (size_t)lds, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

}  // namespace arcle_big
