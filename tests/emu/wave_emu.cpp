// wave_emu.cpp — lock-step CPU emulation of ONE 64-lane wavefront, used to run the kernel body of
// arcle_amd/csrc/arcle_wave.h (the very same header hipcc compiles for gfx950) on the host.
//
// TEST INFRASTRUCTURE ONLY: it lets `pytest -m "not gpu"` check the kernel LOGIC against the oracle and
// the golden vectors without a GPU.  It is not a backend: nothing under arcle_amd/ loads it, and the
// product library fails loudly when the HIP device/extension is missing.
//
// Each lane is a ucontext fiber; every cross-lane primitive of namespace xl is a rendezvous of all 64
// fibers.  The scheduler asserts that all lanes reach the same primitive in the same order — i.e. that
// cross-lane traffic only happens in wave-uniform control flow (on the GPU a bpermute from an
// exec-masked lane would silently read garbage) — and that values declared xl::uniform() really are.
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

#define ARCLE_DEV inline

namespace xl {
static int cur_lane;
static uint32_t exch[64];
static int sync_tag[64];
static long sync_seq[64];
static bool finished[64];
static ucontext_t sched_ctx, lane_ctx[64];
static int error_flag;

static void yield(int tag) {
  int me = cur_lane;
  sync_tag[me] = tag;
  sync_seq[me]++;
  swapcontext(&lane_ctx[me], &sched_ctx);
  cur_lane = me;
}
ARCLE_DEV uint32_t shfl(uint32_t v, int src_lane) {
  exch[cur_lane] = v;
  yield(1);
  uint32_t r = exch[src_lane & 63];
  yield(2);
  return r;
}
ARCLE_DEV unsigned long long ballot(bool b) {
  exch[cur_lane] = b ? 1u : 0u;
  yield(3);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) m |= (unsigned long long)(exch[i] & 1u) << i;
  yield(4);
  return m;
}
ARCLE_DEV uint32_t uniform(uint32_t v) {
  exch[cur_lane] = v;
  yield(5);
  for (int i = 0; i < 64; i++)
    if (exch[i] != v) {
      if (!error_flag) fprintf(stderr, "wave_emu: xl::uniform() value differs across lanes (%u vs %u)\n", exch[i], v);
      error_flag |= 2;
    }
  yield(6);
  return v;
}
ARCLE_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3u)));
}
ARCLE_DEV void lds_fence() { yield(7); }
ARCLE_DEV void atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
ARCLE_DEV int lds_idx(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }  // host memory: stay inside the tile
ARCLE_DEV uint64_t clock() { return 0; }
ARCLE_DEV uint32_t lane_prev(uint32_t v) {
  int me = cur_lane;
  uint32_t r = shfl(v, (me + 63) & 63);
  return me == 0 ? 0u : r;
}
ARCLE_DEV uint32_t lane_next(uint32_t v) {
  int me = cur_lane;
  uint32_t r = shfl(v, (me + 1) & 63);
  return me == 63 ? 0u : r;
}
template <class T>
ARCLE_DEV void pin_ptr(T*&) {}
ARCLE_DEV void pin_u32(uint32_t&) {}
ARCLE_DEV void pin_i32(int32_t&) {}
ARCLE_DEV void keep1(uint32_t&) {}
ARCLE_DEV uint32_t readlane(uint32_t v, int lane) { return shfl(v, lane); }
template <class V>
ARCLE_DEV void keep(V&, V&, uint32_t&, int32_t&) {}
template <class V>
ARCLE_DEV void store16(int8_t* ptr, const V& v) { memcpy(ptr, &v, 16); }
}  // namespace xl

#include "../../arcle_amd/csrc/arcle_wave.h"

namespace {
const arcle::StepParams* g_p;
arcle::WaveLDS g_lds;
int g_env, g_kind;
char* g_stacks;
const size_t STACK = 256 * 1024;

int g_tbl;  // arcle::TBL_* of the installed table (emu_run compares it with the canonical decoders)

#define RUN_STEP(I, F)                                                                        \
  do {                                                                                        \
    if (g_tbl == arcle::TBL_O2ARC) arcle::wave_step<I, F, arcle::TBL_O2ARC>(*g_p, &g_lds, g_env, lane);   \
    else if (g_tbl == arcle::TBL_ARC) arcle::wave_step<I, F, arcle::TBL_ARC>(*g_p, &g_lds, g_env, lane);  \
    else if (g_tbl == arcle::TBL_RAW) arcle::wave_step<I, F, arcle::TBL_RAW>(*g_p, &g_lds, g_env, lane);  \
    else arcle::wave_step<I, F, arcle::TBL_LOOKUP>(*g_p, &g_lds, g_env, lane);                            \
  } while (0)
#define RUN_ROLL(I, F)                                                                        \
  do {                                                                                        \
    if (g_tbl == arcle::TBL_O2ARC) arcle::wave_rollout<I, F, arcle::TBL_O2ARC>(*g_p, &g_lds, g_env, lane); \
    else arcle::wave_rollout<I, F, arcle::TBL_LOOKUP>(*g_p, &g_lds, g_env, lane);                          \
  } while (0)

void lane_main(int lane) {
  xl::cur_lane = lane;
  const bool fw = g_p->W >= 16 && g_p->W <= 32;
  if (g_kind == 0) {
    switch (g_p->ingress * 2 + (fw ? 1 : 0)) {  // the same instantiations the HIP library launches
      case 0: RUN_STEP(0, 0); break;
      case 1: RUN_STEP(0, 1); break;
      case 2: RUN_STEP(1, 0); break;
      case 3: RUN_STEP(1, 1); break;
      case 4: RUN_STEP(2, 0); break;
      default: RUN_STEP(2, 1); break;
    }
  }
  else if (g_kind == 2)
    arcle::wave_reset_table(*g_p, &g_lds, g_env, lane);
  else if (g_kind == 3) {
    switch (g_p->ingress * 2 + (fw ? 1 : 0)) {
      case 2: RUN_ROLL(1, 0); break;
      case 3: RUN_ROLL(1, 1); break;
      case 4: RUN_ROLL(2, 0); break;
      default: RUN_ROLL(2, 1); break;
    }
  }
  else
    arcle::wave_reset(*g_p, &g_lds, g_env, lane);
  xl::finished[lane] = true;
  // returning resumes uc_link (the scheduler)
}

void run_wave() {
  for (int l = 0; l < 64; l++) {
    xl::finished[l] = false;
    xl::sync_seq[l] = 0;
    xl::sync_tag[l] = 0;
    getcontext(&xl::lane_ctx[l]);
    xl::lane_ctx[l].uc_stack.ss_sp = g_stacks + (size_t)l * STACK;
    xl::lane_ctx[l].uc_stack.ss_size = STACK;
    xl::lane_ctx[l].uc_link = &xl::sched_ctx;
    makecontext(&xl::lane_ctx[l], (void (*)())lane_main, 1, l);
  }
  for (;;) {
    int alive = 0;
    for (int l = 0; l < 64; l++) {
      if (xl::finished[l]) continue;
      xl::cur_lane = l;
      swapcontext(&xl::sched_ctx, &xl::lane_ctx[l]);
      if (!xl::finished[l]) alive++;
    }
    if (!alive) break;
    // all lanes that are still running must wait at the same primitive, and none may have finished
    int tag = -1;
    long seq = -1;
    for (int l = 0; l < 64; l++) {
      if (xl::finished[l]) {
        if (!(xl::error_flag & 1)) fprintf(stderr, "wave_emu: lane %d returned while others wait at a cross-lane op (env %d)\n", l, g_env);
        xl::error_flag |= 1;
        continue;
      }
      if (tag < 0) {
        tag = xl::sync_tag[l];
        seq = xl::sync_seq[l];
      } else if (tag != xl::sync_tag[l] || seq != xl::sync_seq[l]) {
        if (!(xl::error_flag & 1)) fprintf(stderr, "wave_emu: divergent cross-lane op (lane %d tag %d vs %d, env %d)\n", l, xl::sync_tag[l], tag, g_env);
        xl::error_flag |= 1;
      }
    }
    if (xl::error_flag & 1) {  // cannot continue a diverged wave safely
      return;
    }
  }
}
}  // namespace

// kind: 0 = step, 1 = reset, 2 = reset from the task table, 3 = rollout.  Fills derived fields (P, PS, div_magic, nseg) like arcle_create does.
extern "C" int emu_run(int kind, arcle::StepParams* p) {
  p->P = p->H * p->W;
  p->PS = (p->P + 15) & ~15;
  p->div_magic = 65536u / (uint32_t)p->W + 1u;
  p->nseg = (p->W >= 16) ? 2 : 1 + (15 + p->W - 1) / p->W;
  if (!g_stacks) g_stacks = (char*)malloc(64 * STACK);
  g_p = p;
  g_kind = kind;
  // canonical-table detection, the same rule libarcle_hip uses: the table must equal what decode_op<TBL> computes
  g_tbl = arcle::TBL_LOOKUP;
  if (p->d_ops) {
    for (int t : {arcle::TBL_O2ARC, arcle::TBL_ARC, arcle::TBL_RAW}) {
      const int n = t == arcle::TBL_O2ARC ? 35 : t == arcle::TBL_ARC ? 27 : 12;
      bool same = p->n_ops == n;
      for (int i = 0; same && i < n; i++) {
        uint32_t d = t == arcle::TBL_O2ARC ? arcle::decode_op<arcle::TBL_O2ARC>(*p, i)
                   : t == arcle::TBL_ARC   ? arcle::decode_op<arcle::TBL_ARC>(*p, i)
                                           : arcle::decode_op<arcle::TBL_RAW>(*p, i);
        same = d == p->d_ops[i];
      }
      if (same) g_tbl = t;
    }
  }
  xl::error_flag = 0;
  for (int env = 0; env < p->n_envs; env++) {
    g_env = env;
    memset(&g_lds, 0xA5, sizeof g_lds);  // stale LDS must never matter
    run_wave();
    if (xl::error_flag & 1) return -100 - xl::error_flag;
  }
  return xl::error_flag ? -100 - xl::error_flag : 0;
}
extern "C" int emu_params_size() { return (int)sizeof(arcle::StepParams); }
