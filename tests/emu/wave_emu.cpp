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
#define ARCLE_HD inline

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
template <int K>
ARCLE_DEV uint32_t row_prev(uint32_t v) {  // row_shr:K — lane j-K of the same 16-lane row, else 0
  int me = cur_lane;
  uint32_t r = shfl(v, (me - K) & 63);
  return (me & 15) < K ? 0u : r;
}
template <int K>
ARCLE_DEV uint32_t row_next(uint32_t v) {  // row_shl:K
  int me = cur_lane;
  uint32_t r = shfl(v, (me + K) & 63);
  return (me & 15) + K > 15 ? 0u : r;
}
ARCLE_DEV uint32_t readlane(uint32_t v, int lane) { return shfl(v, lane); }
ARCLE_DEV uint32_t wave_or(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v |= shfl(v, cur_lane ^ o);
  return v;
}
ARCLE_DEV uint32_t wave_add(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v += shfl(v, cur_lane ^ o);
  return v;
}
ARCLE_DEV uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) {
  for (int k = 0; k < 4; k++) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
}
typedef uint32_t U4 __attribute__((vector_size(16)));
typedef uint32_t U2 __attribute__((vector_size(8)));
ARCLE_DEV U4 load16u(const int8_t* p) { U4 v; memcpy(&v, p, 16); return v; }
// wave-uniform scalar loads: every lane reads the same address
ARCLE_DEV uint32_t uload1(const void* p) { uint32_t v; memcpy(&v, p, 4); return v; }
ARCLE_DEV U2 uload2(const void* p) { U2 v; memcpy(&v, p, 8); return v; }
ARCLE_DEV U4 uload4(const void* p) { U4 v; memcpy(&v, p, 16); return v; }
ARCLE_DEV U4 load16(const int8_t* base, uint32_t off) { U4 v; memcpy(&v, base + off, 16); return v; }
ARCLE_DEV void store16(int8_t* base, uint32_t off, const U4& v) { memcpy(base + off, &v, 16); }
ARCLE_DEV void store16_nt(int8_t* base, uint32_t off, const U4& v) { store16(base, off, v); }
ARCLE_DEV void release_store_system(uint32_t* p, uint32_t v) { *p = v; }
ARCLE_DEV void wg_barrier() { yield(8); }
ARCLE_DEV void lanes_converged() { yield(9); }
ARCLE_DEV uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
ARCLE_DEV uint32_t opaque(uint32_t v) { return v; }
ARCLE_DEV int rare_s(int v) { return v; }
ARCLE_DEV int rare_v(int v) { return v; }
ARCLE_DEV uint32_t tov(uint32_t x) { return x; }
ARCLE_DEV uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) {  // v_perm_b32 for selector bytes 0..7
  const uint64_t t = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int k = 0; k < 4; k++) r |= (uint32_t)((t >> (8 * ((sel >> (8 * k)) & 7u))) & 0xffu) << (8 * k);
  return r;
}
template <typename T>
ARCLE_DEV void store_at(void* base, uint32_t off, const T& v) { memcpy((char*)base + off, &v, sizeof(T)); }
ARCLE_DEV uint32_t bfrev(uint32_t v) {
  uint32_t r = 0;
  for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}
#define ARCLE_STOP_AT 0
ARCLE_DEV void sink_s(uint32_t) {}
ARCLE_DEV void own_stores_visible() {}
ARCLE_DEV void sink_v(uint32_t) {}
ARCLE_DEV void arrived(U4&, U2&, uint32_t&, U4&) {}
ARCLE_DEV void arrived3(U4&, U2&, uint32_t&) {}
}  // namespace xl

#include "../../arcle_amd/csrc/arcle_wave.h"

namespace {
const arcle::StepParams* g_p;
arcle::BlockLDS<1> g_lds;
int g_env, g_kind;
char* g_stacks;
const size_t STACK = 256 * 1024;


#define RUN_STEP(I, F)                                                                      \
  do {                                                                                      \
    arcle::Wave w(*g_p, &g_lds.wave[0], nullptr, lane, I, F, false, true, false); /* as the step kernel: no expansion table */ \
    arcle::StepInputs in = arcle::load_inputs<I>(w, g_env);                                 \
    if (g_p->flags & ARCLE_STEP_FEATURE_FLAGS) arcle::wave_step<I, F, 1, 1>(w, g_env, in);           \
    else arcle::wave_step<I, F, 1, 0>(w, g_env, in);                                        \
  } while (0)
#define RUN_TRANS(I, F) arcle::wave_transition_row<I, F>(*g_p, &g_lds.wave[0], g_lds.lut, g_env, lane)
#define RUN_ROLL(I, F) arcle::wave_rollout<I, F>(*g_p, &g_lds.wave[0], g_lds.lut, g_env, lane)

// the same width classes the HIP library launches: FW_FULL when 16 <= W <= 32 and the plane stride is 1024
int width_class() {
  if (g_p->W < 16 || g_p->W > 32) return arcle::FW_GENERIC;
  return g_p->PS == 1024 ? arcle::FW_FULL : arcle::FW_FAST;
}

void lane_main(int lane) {
  xl::cur_lane = lane;
  const int fw = width_class();
  arcle::lut_init(g_lds.lut, lane, 64);
  xl::wg_barrier();
  if (g_kind == 0) {
    switch (g_p->ingress * 3 + fw) {
      case 0: RUN_STEP(0, 0); break;
      case 1: RUN_STEP(0, 1); break;
      case 2: RUN_STEP(0, 2); break;
      case 3: RUN_STEP(1, 0); break;
      case 4: RUN_STEP(1, 1); break;
      case 5: RUN_STEP(1, 2); break;
      case 6: RUN_STEP(2, 0); break;
      case 7: RUN_STEP(2, 1); break;
      case 8: RUN_STEP(2, 2); break;
      case 9: RUN_STEP(3, 0); break;
      case 10: RUN_STEP(3, 1); break;
      case 11: RUN_STEP(3, 2); break;
      case 12: RUN_STEP(4, 0); break;
      case 13: RUN_STEP(4, 1); break;
      default: RUN_STEP(4, 2); break;
    }
  }
  else if (g_kind == 6)
    arcle::wave_set_state_row(*g_p, &g_lds.wave[0], g_lds.lut, g_env, lane);
  else if (g_kind == 7) {
    const int f = fw ? 1 : 0;  // (as the library: FW_FAST code for FW_FULL)
    switch (g_p->ingress * 2 + f) {
      case 0: RUN_TRANS(0, 0); break;
      case 1: RUN_TRANS(0, 1); break;
      case 2: RUN_TRANS(1, 0); break;
      case 3: RUN_TRANS(1, 1); break;
      case 4: RUN_TRANS(2, 0); break;
      default: RUN_TRANS(2, 1); break;
    }
  }
  else if (g_kind == 8) {  // arcle_pack_mask_bits
    arcle::Wave w(*g_p, &g_lds.wave[0], g_lds.lut, lane, arcle::INGRESS_MASK, arcle::FW_GENERIC, false);
    const arcle::U4 v = arcle::load_payload(w, g_env, 0, g_p->sel);
    const uint32_t m = arcle::nz16(v) & w.valid16;
    uint8_t* bits = reinterpret_cast<uint8_t*>(g_p->flat_out) + (size_t)g_env * ARCLE_BITS_STRIDE + 2 * lane;
    bits[0] = (uint8_t)m;
    bits[1] = (uint8_t)(m >> 8);
  }
  else if (g_kind == 2)
    arcle::wave_reset_table(*g_p, &g_lds.wave[0], g_lds.lut, g_env, lane);
  else if (g_kind == 3) {
    const int f = fw ? 1 : 0;  // (the library launches FW_FAST rollouts for FW_FULL too)
    switch (g_p->ingress * 2 + f) {
      case 0: RUN_ROLL(0, 0); break;
      case 1: RUN_ROLL(0, 1); break;
      case 2: RUN_ROLL(1, 0); break;
      case 3: RUN_ROLL(1, 1); break;
      case 4: RUN_ROLL(2, 0); break;
      case 5: RUN_ROLL(2, 1); break;
      case 8: RUN_ROLL(4, 0); break;
      default: RUN_ROLL(4, 1); break;
    }
  }
  else if (g_kind == 4)
    arcle::wave_flatten(*g_p, &g_lds.wave[0], g_lds.lut, g_env, lane);
  else if (g_kind == 5)
    arcle::wave_pack_obs(*g_p, &g_lds.wave[0], g_lds.lut, g_env, lane);
  else
    arcle::wave_reset(*g_p, &g_lds.wave[0], g_lds.lut, g_env, lane);
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

// kind: 0 = step, 1 = reset, 2 = reset from the task table (task_idx NULL: device-drawn), 3 = rollout, 4 = flatten, 5 = pack,
// 6 = set_state_rows, 7 = transition_rows (n_envs = rows, n_resident = envs), 8 = pack_mask_bits (flat_out = the bit rows).  Fills derived fields (P, div_magic, nseg) like
// arcle_create does; PS (plane stride) comes from the caller (0 = default).
extern "C" int emu_run(int kind, arcle::StepParams* p) {
  p->P = p->H * p->W;
  if (p->PS == 0) p->PS = ARCLE_DEFAULT_PLANE_STRIDE(p->P);
  p->div_magic = 65536u / (uint32_t)p->W + 1u;
  p->nseg = (p->W >= 16) ? 2 : 1 + (15 + p->W - 1) / p->W;
  if (!g_stacks) g_stacks = (char*)malloc(64 * STACK);
  g_p = p;
  g_kind = kind;
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
