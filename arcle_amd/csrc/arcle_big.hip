// arcle_big.hip — gfx950 kernels for grids beyond ARCLE_MAX_CELLS (H * W > 1024, H, W <= 127): one WORKGROUP of 128 … 512 threads per env.
// The bodies live in arcle_big.h (also compiled by the test emulator); this file supplies the workgroup primitives (namespace bx),
// the __global__ wrappers and the host-side launchers arcle_hip.hip routes big handles to.  Second translation unit of
// libarcle_hip.so (arcle_amd/_lib.py compiles both and links them).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define ARCLE_BIG_DEV __device__ __forceinline__
#define ARCLE_BIG_HD __host__ __device__
#define ARCLE_BIG_ROWS 1  // board rows per thread of the flood fill: a workgroup has at least 128 threads (threads_for), a plane at most 127 rows

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
ARCLE_BIG_DEV uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }                                     // v_mul_u32_u24 (full rate)
ARCLE_BIG_DEV uint32_t mul32(uint32_t a, uint32_t b) { return a * b; }
ARCLE_BIG_DEV int dot4_i8(uint32_t v, int acc) { return __builtin_amdgcn_sdot4((int)v, 0x01010101, acc, false); }  // acc + the four int8 of v (v_dot4_i32_i8)
ARCLE_BIG_DEV uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }  // (hi:lo) >> sh, sh < 32
ARCLE_BIG_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// ---- wavefront-wide reductions (every lane of the wavefront active) -----------------------------------------------------------------
enum { HAS_WAVE_OPS = 1 };
ARCLE_BIG_DEV bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
// v op= its neighbours 1, 2, 4 and 8 lanes below inside each row of 16 (lanes without such a neighbour take `id`), then row 0's / row 2's
// lane 15 into rows 1 / 3 and lane 31 into rows 2 - 3: lane 63 holds the reduction of all 64 lanes
#define BIG_WAVE_REDUCE(OP, ID)                                                                      \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(ID), v, 0x111, 0xf, 0xf, false)); /* row_shr:1 */       \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(ID), v, 0x112, 0xf, 0xf, false)); /* row_shr:2 */       \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(ID), v, 0x114, 0xf, 0xf, false)); /* row_shr:4 */       \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(ID), v, 0x118, 0xf, 0xf, false)); /* row_shr:8 */       \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(ID), v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 */    \
  v = OP(v, __builtin_amdgcn_update_dpp((int)(ID), v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 */    \
  return __builtin_amdgcn_readlane(v, 63)
ARCLE_BIG_DEV int op_add(int a, int b) { return a + b; }
ARCLE_BIG_DEV int op_min(int a, int b) { return a < b ? a : b; }
ARCLE_BIG_DEV int op_max(int a, int b) { return a > b ? a : b; }
ARCLE_BIG_DEV int op_umax(int a, int b) { return (uint32_t)a > (uint32_t)b ? a : b; }
ARCLE_BIG_DEV int wave_add(int v) { BIG_WAVE_REDUCE(op_add, 0); }
ARCLE_BIG_DEV int wave_min(int v) { BIG_WAVE_REDUCE(op_min, 0x7fffffff); }
ARCLE_BIG_DEV int wave_max(int v) { BIG_WAVE_REDUCE(op_max, (int)0x80000000); }
ARCLE_BIG_DEV uint32_t wave_umax(uint32_t u) {
  int v = (int)u;
  BIG_WAVE_REDUCE(op_umax, 0);
}
// a dword of a table no kernel writes, at an index every lane holds alike: through the constant address space, i.e. a scalar load even
// behind stores the compiler cannot tell apart from it
ARCLE_BIG_DEV uint32_t sload32(const uint32_t* p, uint32_t i) {
  return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p + i));
}  // a value every lane holds alike, moved to a scalar register
ARCLE_BIG_DEV void release_store_system(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}  // namespace bx

#include "arcle_big.h"

using arcle_big::BigParams;

#define BIG_THREADS 512  // the largest workgroup a launch is ever given (threads_for / lean_threads_for; ARCLE_BIG_THREADS overrides up to it)

extern __shared__ __attribute__((aligned(16))) int8_t arcle_big_lds[];

__global__ __launch_bounds__(BIG_THREADS) void arcle_big_step_kernel(const BigParams p) {
  arcle_big::step_env(p, (int)blockIdx.x, arcle_big_lds);
}
// The LEAN instantiations (arcle_big.h CtxT): the launch's flag set lies within LEAN_FLAGS, W >= 16, no accounting, no scratch rows; CPT: at
// most that many plane chunks per thread.  What ARCVecEnv's plain step calls run; everything else takes the generic kernel above.
// (Instantiations compiled for 64-thread workgroups alone — one wavefront per env, the compiler drops the barriers — measured the same as
// these launched with 64 threads: 25.7 / 25.6 us at 40 x 40 x 16 384, profiles/round6_experiments.txt §2.)
template <int CPT, int ING, int FL>
__global__ __launch_bounds__(arcle_big::LEAN_MAX_THREADS) void arcle_big_step_lean(const BigParams p) {
  arcle_big::step_env_t<arcle_big::CtxT<CPT, true>, ING, FL>(p, (int)blockIdx.x, arcle_big_lds);
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

// the dynamic LDS of a launch: 65 360 bytes at 127 x 127; a plane stride that needs more than the 64 KB a kernel gets without asking
// (a caller-chosen stride with padding) asks for it
template <int ID, class K>
static int allow_lds(K kernel, int bytes) {
  if (bytes <= 65536) return 0;
  static uint64_t done = 0;  // per kernel (ID) and device: the attribute is set once, for the largest plane
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (done & (1ull << (dev & 63))) return 0;
  const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(MAX_PS, 127));
  if (rc == 0) done |= 1ull << (dev & 63);
  return rc;
}

// Threads per workgroup: a thread owns the chunks t, t + NT, ... of a plane.  About one chunk per thread won every sweep on MI355X, with the
// per-cell kernels and with the whole-chunk (SWAR) ones (profiles/round5_experiments.txt §15, §20: 40 x 40 x 16 384 envs 39 us with 128
// threads, 61 with 256, 117 with 512; 127 x 127: 264 us with 512, 312 with 256, 443 with 128) — so: the chunk count rounded up to whole
// wavefronts, at least 128 (the flood fill gives every board row its own thread) and at most 512 (128 threads at 40 x 40, 256 at 64 x 64,
// 512 at 127 x 127).  ARCLE_BIG_THREADS overrides (tuning runs).  (The generic kernel and the reset / row kernels; the LEAN step launches
// size themselves, lean_threads_for below.)
static unsigned threads_for(int PS) {
  static int forced = -1;
  if (forced < 0) {
    const char* s = getenv("ARCLE_BIG_THREADS");
    const int v = s ? atoi(s) : 0;
    forced = (v >= 128 && v <= BIG_THREADS && (v & 63) == 0) ? v : 0;
  }
  if (forced) return (unsigned)forced;
  const int nch = PS >> 4;
  const int t = (nch + 63) & ~63;
  return (unsigned)(t < 128 ? 128 : t > 512 ? 512 : t);
}

int workgroup_threads(int PS) { return (int)threads_for(PS); }
static bool lean_launch(const BigParams& p);
static unsigned lean_threads_for(int PS, int H);
int step_threads(const BigParams& p) { return lean_launch(p) ? (int)lean_threads_for(p.PS, p.H) : (int)threads_for(p.PS); }

template <int ID, int CPT, int ING, int FL = -1>
static int launch_lean(const BigParams& p, unsigned nt, int lds, void* stream) {
  if (int rc = allow_lds<ID>(arcle_big_step_lean<CPT, ING, FL>, lds)) return rc;
  hipLaunchKernelGGL((arcle_big_step_lean<CPT, ING, FL>), dim3((unsigned)p.n_envs), dim3(nt), (size_t)lds, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
// two chunks per thread, the exact ingress form; the flag set AUTORESET | ELIDE_SELECTED (ARCVecEnv(autoreset=True), the benchmark) as a
// compile-time constant, any other LEAN flag set at run time
enum { HOT_FLAGS = ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED };
template <int ID, int ING>
static int launch_lean2(const BigParams& p, unsigned nt, int lds, void* stream) {
  if (p.flags == (uint32_t)HOT_FLAGS) return launch_lean<ID, 2, ING_T_EXACT + ING, HOT_FLAGS>(p, nt, lds, stream);
  return launch_lean<ID + 1, 2, ING_T_EXACT + ING>(p, nt, lds, stream);
}

static int env_int(const char* name) {
  const char* s = getenv(name);
  return s ? atoi(s) : 0;
}
// ARCLE_BIG_GENERIC=1: every step launch takes the generic kernel (A/B runs)
static bool lean_allowed() {
  static int v = -1;
  if (v < 0) v = env_int("ARCLE_BIG_GENERIC") ? 0 : 1;
  return v != 0;
}
// Threads per workgroup of a LEAN step launch: TWO chunks per thread — 64 threads (one wavefront per env) at 40 x 40, 128 at 64 x 64, 512 at
// 127 x 127; at least one thread per plane row (the flood fill's boards).  Same library, same run, C3 mix, 16 384 envs: 40 x 40 29.7 us with
// one chunk per thread, 27.1 with two; 64 x 64 50.9 / 46.3; four: no better at 40 x 40, 57 at 64 x 64 (profiles/round6_experiments.txt §2).
// ARCLE_BIG_CPT = 1 / 2 picks the chunks per thread, ARCLE_BIG_THREADS the size itself (tuning runs).
static unsigned lean_threads_for(int PS, int H) {
  static int cpt = -1, forced = -1;
  if (cpt < 0) {
    const int v = env_int("ARCLE_BIG_CPT"), f = env_int("ARCLE_BIG_THREADS");
    cpt = v == 1 ? 1 : 2;
    forced = (f >= 64 && f <= LEAN_MAX_THREADS && (f & 63) == 0) ? f : 0;
  }
  const int nch = PS >> 4, rows = (H + 63) & ~63;
  int t = forced ? forced : (((nch + cpt - 1) / cpt) + 63) & ~63;
  if (t < rows) t = rows;
  return (unsigned)(t > LEAN_MAX_THREADS ? LEAN_MAX_THREADS : t);
}

// does a step launch with these parameters take a LEAN kernel?  (what they assume, arcle_big.h CtxT; a forced workgroup too small for two
// chunks per thread falls back to the generic kernel)
static bool lean_launch(const BigParams& p) {
  if (!lean_allowed() || (p.flags & ~(uint32_t)LEAN_FLAGS) || p.W < 16 || p.res_rec || p.acct) return false;
  return (unsigned)(p.PS >> 4) <= 2u * lean_threads_for(p.PS, p.H);
}

int launch_step(const BigParams& p0, void* stream) {
  const BigParams p = with_magic(p0);
  const int lds = lds_bytes(p.PS, p.H);
  if (lean_launch(p)) {
    const unsigned nt = lean_threads_for(p.PS, p.H);
    const int need = (int)(((unsigned)(p.PS >> 4) + nt - 1) / nt);
    const bool masks = p.ingress == ING_MASK || p.ingress == ING_BITS;
    // (one chunk per thread — tuning runs, ARCLE_BIG_CPT=1 — by ingress family; the shipped two-chunk launches by exact form)
    if (need <= 1) return masks ? launch_lean<4, 1, ING_T_MASKS>(p, nt, lds, stream) : launch_lean<5, 1, ING_T_TUPLES>(p, nt, lds, stream);
    switch (p.ingress) {
      case ING_MASK: return launch_lean2<6, ING_MASK>(p, nt, lds, stream);
      case ING_BITS: return launch_lean2<8, ING_BITS>(p, nt, lds, stream);
      case ING_BBOX: return launch_lean2<10, ING_BBOX>(p, nt, lds, stream);
      case ING_POINT: return launch_lean2<12, ING_POINT>(p, nt, lds, stream);
      default: return launch_lean2<14, ING_BBOX5>(p, nt, lds, stream);
    }
  }
  const unsigned nt = threads_for(p.PS);
  if (int rc = allow_lds<0>(arcle_big_step_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_step_kernel, dim3((unsigned)p.n_envs), dim3(nt), (size_t)lds, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}
int launch_reset(const BigParams& p0, int mode, void* stream) {
  const BigParams p = with_magic(p0);
  const int lds = lds_bytes(p.PS, p.H);
  if (int rc = allow_lds<1>(arcle_big_reset_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_reset_kernel, dim3((unsigned)p.n_envs), dim3(threads_for(p.PS)), (size_t)lds, (hipStream_t)stream, p, mode);
  return (int)hipGetLastError();
}
int launch_rows(const BigParams& p0, int mode, void* stream) {
  const BigParams p = with_magic(p0);
  const int lds = lds_bytes(p.PS, p.H);
  if (int rc = allow_lds<2>(arcle_big_rows_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_rows_kernel, dim3((unsigned)p.n_envs), dim3(threads_for(p.PS)), (size_t)lds, (hipStream_t)stream, p, mode);
  return (int)hipGetLastError();
}
int launch_set_rows(const BigParams& p0, void* stream) {
  const BigParams p = with_magic(p0);
  const int lds = lds_bytes(p.PS, p.H);
  if (int rc = allow_lds<3>(arcle_big_set_rows_kernel, lds)) return rc;
  hipLaunchKernelGGL(arcle_big_set_rows_kernel, dim3((unsigned)p.n_envs), dim3(threads_for(p.PS)), (size_t)lds, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

}  // namespace arcle_big
