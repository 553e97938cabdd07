// big_emu.cpp — TEST INFRASTRUCTURE: runs the workgroup-per-env kernel bodies of arcle_amd/csrc/arcle_big.h on the CPU, so that their
// logic can be checked against the oracle without a GPU.  A "workgroup" is `nthreads` host threads (>= 16) that walk the envs together;
// the workgroup barrier is a pthread barrier, LDS is one shared buffer, LDS atomics are GCC atomics.  Never part of the product.
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

#define ARCLE_BIG_DEV inline
#define ARCLE_BIG_HD
#define ARCLE_BIG_ROWS 8  // board rows per thread of the flood fill: 127 rows over the 16 threads of the emulated workgroup

namespace bx {
static thread_local int t_tid;
static int g_nt;
static pthread_barrier_t g_bar;
inline int tid() { return t_tid; }
inline int nt() { return g_nt; }
inline void sync() { pthread_barrier_wait(&g_bar); }
inline void sync_release() { sync(); }
inline void lds_or(int32_t* a, int v) { __atomic_fetch_or(a, v, __ATOMIC_RELAXED); }
inline void lds_add(int32_t* a, int v) { __atomic_fetch_add(a, v, __ATOMIC_RELAXED); }
inline void lds_min(int32_t* a, int v) {
  int32_t cur = __atomic_load_n(a, __ATOMIC_RELAXED);
  while (v < cur && !__atomic_compare_exchange_n(a, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
}
inline void lds_max(int32_t* a, int v) {
  int32_t cur = __atomic_load_n(a, __ATOMIC_RELAXED);
  while (v > cur && !__atomic_compare_exchange_n(a, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
}
inline void lds_umax(uint32_t* a, uint32_t v) {
  uint32_t cur = __atomic_load_n(a, __ATOMIC_RELAXED);
  while (v > cur && !__atomic_compare_exchange_n(a, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
  }
}
inline void status_or(uint32_t* g, uint32_t v) { __atomic_fetch_or(g, v, __ATOMIC_RELAXED); }
inline uint64_t brev64(uint64_t x) {
  x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
  return __builtin_bswap64(x);
}
inline void release_store_system(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline int uniform(int v) { return v; }
// (host threads are no wavefront: the kernel body takes its atomic form, arcle_big.h; these only have to exist)
enum { HAS_WAVE_OPS = 0 };
inline bool wave_any(bool b) { return b; }
inline int wave_add(int v) { return v; }
inline int wave_min(int v) { return v; }
inline int wave_max(int v) { return v; }
inline uint32_t wave_umax(uint32_t v) { return v; }
inline uint32_t sload32(const uint32_t* p, uint32_t i) { return p[i]; }
inline uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline uint32_t mul32(uint32_t a, uint32_t b) { return a * b; }
inline int dot4_i8(uint32_t v, int acc) { return acc + (int8_t)(v & 0xff) + (int8_t)((v >> 8) & 0xff) + (int8_t)((v >> 16) & 0xff) + (int8_t)(v >> 24); }
inline uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | (uint64_t)lo) >> (sh & 31u)); }
}  // namespace bx

#include "../../arcle_amd/csrc/arcle_big.h"

using arcle_big::BigParams;

extern "C" int big_emu_params_size(void) { return (int)sizeof(BigParams); }
extern "C" int big_emu_lds_bytes(int PS, int H) { return arcle_big::lds_bytes(PS, H); }

// what: 0 step (generic instantiation), 1 reset (mode 0 / 1 / 2), 2 rows out (mode 0 flat / 1 packed), 3 state rows in,
//       4 step, a LEAN instantiation for this ingress family; mode = its compile-time bound on the chunks per thread (1 / 2: the forms the
//         product launches — 2 by default; 4: the template's general form; 0: the run-time loop, which lets 16 host threads walk a plane of
//         any size)
extern "C" int big_emu_run(int what, const BigParams* p_in, int mode, int nthreads) {
  const BigParams filled = arcle_big::with_magic(*p_in);  // (what the product's launchers do, arcle_big.hip)
  const BigParams* const p = &filled;
  if (nthreads < arcle_big::MIN_THREADS || p->PS > arcle_big::MAX_PS) return -1;
  if (what == 4) {  // what the launcher checks before it picks a LEAN kernel (arcle_big.hip lean_ok)
    if ((p->flags & ~(uint32_t)arcle_big::LEAN_FLAGS) || p->W < 16 || p->res_rec || p->acct) return -3;
    if (mode != 0 && mode != 1 && mode != 2 && mode != 4) return -4;
    if (mode && (p->PS >> 4) > mode * nthreads) return -4;
    if (mode >= 2 && (p->PS >> 4) <= nthreads) return -4;  // (these instantiations take every thread's first chunk for granted)
  }
  const bool masks = p->ingress == arcle_big::ING_MASK || p->ingress == arcle_big::ING_BITS;
  void* lds = nullptr;
  if (posix_memalign(&lds, 64, (size_t)arcle_big::lds_bytes(p->PS, p->H))) return -2;
  memset(lds, 0x5a, (size_t)arcle_big::lds_bytes(p->PS, p->H));  // LDS is not zero at kernel start
  bx::g_nt = nthreads;
  pthread_barrier_init(&bx::g_bar, nullptr, (unsigned)nthreads);
  std::vector<std::thread> th;
  for (int k = 0; k < nthreads; k++)
    th.emplace_back([=]() {
      bx::t_tid = k;
      for (int env = 0; env < p->n_envs; env++) {
        switch (what) {
          case 0: arcle_big::step_env(*p, env, (int8_t*)lds); break;
          case 4:
#define BIG_EMU_LEAN(CPT)                                                                                                   \
  if (masks) arcle_big::step_env_t<arcle_big::CtxT<CPT, true>, arcle_big::ING_T_MASKS>(*p, env, (int8_t*)lds);              \
  else arcle_big::step_env_t<arcle_big::CtxT<CPT, true>, arcle_big::ING_T_TUPLES>(*p, env, (int8_t*)lds)
#define BIG_EMU_EXACT(F)                                                                                                                \
  if (p->flags == 3u) arcle_big::step_env_t<arcle_big::CtxT<2, true>, arcle_big::ING_T_EXACT + arcle_big::F, 3>(*p, env, (int8_t*)lds);  \
  else arcle_big::step_env_t<arcle_big::CtxT<2, true>, arcle_big::ING_T_EXACT + arcle_big::F>(*p, env, (int8_t*)lds)
            if (mode == 1) { BIG_EMU_LEAN(1); }
            else if (mode == 2) {  // (what the product launches by default: the instantiation of the exact ingress form)
              switch (p->ingress) {
                case arcle_big::ING_MASK: BIG_EMU_EXACT(ING_MASK); break;
                case arcle_big::ING_BITS: BIG_EMU_EXACT(ING_BITS); break;
                case arcle_big::ING_BBOX: BIG_EMU_EXACT(ING_BBOX); break;
                case arcle_big::ING_POINT: BIG_EMU_EXACT(ING_POINT); break;
                default: BIG_EMU_EXACT(ING_BBOX5); break;
              }
            }
            else if (mode == 4) { BIG_EMU_LEAN(4); }
            else { BIG_EMU_LEAN(0); }
            break;
          case 1: arcle_big::reset_env(*p, env, mode, (int8_t*)lds); break;
          case 2: arcle_big::rows_env(*p, env, mode, (int8_t*)lds); break;
          default: arcle_big::set_rows_env(*p, env, (int8_t*)lds); break;
        }
        bx::sync();  // the next env's "workgroup" starts after this one has finished
      }
    });
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&bx::g_bar);
  free(lds);
  return 0;
}
