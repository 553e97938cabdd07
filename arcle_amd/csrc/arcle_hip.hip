// arcle_hip.hip — gfx950 kernels + the C ABI of include/arcle_hip.h  (libarcle_hip.so)
//
// Kernels (bodies in arcle_wave.h; one wavefront per env everywhere)
//   arcle_step_kernel          one step() of every env; 4 or 8 waves per workgroup (plan_launch); the launches of the standard batch order
//                              themselves inside groups of 32 envs (the GROUPED block)
//   arcle_transition_rows_kernel  the stateless transition(state, action) on flattened state rows
//   arcle_rollout_kernel       n_steps step()s per launch with the env state resident in registers
//   arcle_reset[_table]_kernel init_state for (masked) envs, optionally from the device task table / device-drawn tasks
//   arcle_flatten_kernel       flattened observation rows (also an epilogue of the step kernel: ARCLE_STEP_FLAT_OBS)
//   arcle_pack_kernel          packed per-step observation rows for the multi-GPU gather
// Launch geometry: grid = ceil(N / waves per workgroup) rounded up to a multiple of 8.  Workgroup b is observed
// to run on XCD b%8 (MI355X_MICROARCH.md §Workgroup dispatch); the block->env map below gives each XCD
// one contiguous range of envs, so an env's state stays in one XCD's L2 / Infinity-Cache slice from step to step.
// That is an affinity choice only — correctness never depends on placement (envs share no
// data and no workgroup communicates with another).
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#define ARCLE_DEV __device__ __forceinline__
#define ARCLE_HD __host__ __device__ __forceinline__

namespace xl {  // cross-lane / memory primitives of one 64-lane wavefront
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
typedef uint32_t U2 __attribute__((ext_vector_type(2)));
#define ARCLE_AS_GLOBAL __attribute__((address_space(1)))
#define ARCLE_AS_CONST __attribute__((address_space(4)))
ARCLE_DEV uint32_t shfl(uint32_t v, int src_lane) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
ARCLE_DEV unsigned long long ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
ARCLE_DEV uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
ARCLE_DEV uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// orders this wave's LDS traffic (each wave owns a private LDS tile: no workgroup barrier needed)
ARCLE_DEV void lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
ARCLE_DEV void wg_barrier() { __syncthreads(); }
// a wavefront executes in lock step: nothing to do (the CPU emulator runs lanes as fibers and needs a rendezvous here)
ARCLE_DEV void lanes_converged() {}
ARCLE_DEV void atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
// LDS reads outside the workgroup's allocation return 0 and reads inside it but outside this wave's tile
// return bytes every caller masks away, so tile indices are not clamped on the GPU
ARCLE_DEV int lds_idx(int i, int /*n*/) { return i; }
// Wave-uniform loads of per-env scalars (record, op index, counters, bbox / point payload): the address is uniform and
// the location is not written by anyone else during the launch, so they go through the scalar cache straight into
// SGPRs (s_load_dword[x2|x4]) — no VGPRs, no v_readfirstlane, handled by the scalar ALU afterwards.
#ifdef ARCLE_VECTOR_INPUTS  // A/B: the same values through the vector memory path (global_load + v_readfirstlane)
ARCLE_DEV uint32_t uload1(const void* p) { return uniform(*reinterpret_cast<const ARCLE_AS_GLOBAL uint32_t*>((uintptr_t)p)); }
ARCLE_DEV U2 uload2(const void* p) {
  U2 v = *reinterpret_cast<const ARCLE_AS_GLOBAL U2*>((uintptr_t)p);
  v[0] = uniform(v[0]);
  v[1] = uniform(v[1]);
  return v;
}
ARCLE_DEV U4 uload4(const void* p) {
  U4 v = *reinterpret_cast<const ARCLE_AS_GLOBAL U4*>((uintptr_t)p);
  for (int i = 0; i < 4; i++) v[i] = uniform(v[i]);
  return v;
}
#else
ARCLE_DEV uint32_t uload1(const void* p) { return *reinterpret_cast<const ARCLE_AS_CONST uint32_t*>((uintptr_t)p); }
ARCLE_DEV U2 uload2(const void* p) { return *reinterpret_cast<const ARCLE_AS_CONST U2*>((uintptr_t)p); }
ARCLE_DEV U4 uload4(const void* p) { return *reinterpret_cast<const ARCLE_AS_CONST U4*>((uintptr_t)p); }
#endif
// 16 B plane load: SGPR base + 32-bit VGPR byte offset (global_load_dwordx4 v, v_off, s[base])
ARCLE_DEV U4 load16(const int8_t* base, uint32_t off) {
  return *reinterpret_cast<const ARCLE_AS_GLOBAL U4*>((uintptr_t)base + off);
}
// DPP quad_perm: every lane of a quad takes the value of the quad's lane K / lanes 0-1 take lane 1's, lanes 2-3 lane 3's
template <int K>
ARCLE_DEV uint32_t quad_bcast(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, true); }
ARCLE_DEV uint32_t quad_bcast_odd(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xf5, 0xf, 0xf, true); }
ARCLE_DEV U2 load8(const void* base, uint32_t off) { return *reinterpret_cast<const ARCLE_AS_GLOBAL U2*>((uintptr_t)base + off); }
ARCLE_DEV uint32_t load32(const void* base, uint32_t off) {  // one dword per lane: SGPR base + 32-bit VGPR byte offset
  return *reinterpret_cast<const ARCLE_AS_GLOBAL uint32_t*>((uintptr_t)base + off);
}
// 16 B plane store, write-through (`sc1`): the planes written by a step are only read again by the NEXT launch,
// and per-XCD L2s are written back at every kernel boundary anyway; writing through lets that traffic overlap
// the kernel instead of being flushed at its end (profiles/archive/round1_store_policy_ab.txt: plain 11.46 us, nt 11.08,
// sc1 9.98 per launch of the C3 mix).  For state far beyond the 256 MiB Infinity Cache `nt` streams better
// (tools/membench.hip, N = 131072: nt 43 us vs sc1 62 us), selectable at build time.
// The trailing s_nop covers the ">64-bit VMEM store data" hazard: hipcc's hazard recogniser does not see into
// inline asm and may overwrite the data VGPRs in the very next instruction (it did: parity caught it).
#ifndef ARCLE_STORE_POLICY
#define ARCLE_STORE_POLICY "sc1"
#endif
// The leading s_nop 4 covers "VALU writes an SGPR (v_readlane of a spilled pointer, v_readfirstlane) -> VMEM reads it as
// saddr" (5 wait states on gfx9): for its own instructions the compiler inserts them, inside inline asm it cannot — the
// register-starved rollout instantiations reloaded the plane pointer with v_readlane right before the store and wrote to
// a stale address (GPU memory fault; tools/dbg_rollout.py).  It only idles this wave's issue slot.
ARCLE_DEV void store16(int8_t* base, uint32_t off, const U4& v) {
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 " ARCLE_STORE_POLICY "\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
}
// Non-temporal forms for the streaming instantiations (ARCLE_STEPX_STORE_NT / _EARLY_NT; which batch sizes take which is the launcher's
// table, measured in profiles/round4_experiments.txt).  The load is only used for the speculative grid request the KERNEL issues:
// a per-instantiation hint on Wave::load_hbm does not survive the optimiser (the method is simplified as a function of its own before it
// is inlined, and two arms loading the same address become ONE load without the hint).
ARCLE_DEV U4 load16_nt(const int8_t* base, uint32_t off) {
  return __builtin_nontemporal_load(reinterpret_cast<const ARCLE_AS_GLOBAL U4*>((uintptr_t)base + off));
}
ARCLE_DEV void store16_nt(int8_t* base, uint32_t off, const U4& v) {
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
}
// lane-0 stores of the step outputs: uniform base + byte offset held in a VGPR -> `global_store v_off, data, s[base]` (the saddr
// form).  With the offset in an SGPR the compiler forms the 64-bit address with s_add_u32 / s_addc_u32 per store — scalar
// instructions, the one kind this kernel is short of (profiles/round3_experiments.txt: 8-11 ns per launch each)
template <typename T>
ARCLE_DEV void store_at(void* base, uint32_t off, const T& v) {
  asm volatile("" : "+v"(off));  // (the offset lives in a VGPR from here on)
  typedef std::conditional_t<sizeof(T) == 16, U4, std::conditional_t<sizeof(T) == 8, U2, std::conditional_t<sizeof(T) == 4, uint32_t, uint8_t>>> R;
  static_assert(sizeof(R) == sizeof(T), "16 / 8 / 4 / 1 byte values");
  // (plain write-back stores: written THROUGH (`sc1`) these few bytes per wave are one fabric write each and the kernel runs 3.8 -> 5.05 us,
  // while the ~1 us between two launches does not shrink — profiles/round5_launch_trace.txt)
  *reinterpret_cast<ARCLE_AS_GLOBAL R*>((uintptr_t)base + off) = __builtin_bit_cast(R, v);
}
// release at SYSTEM scope + the store: everything this wave stored before (vmcnt is per wave: all lanes' stores) is visible to the host
// when it reads `v` — s_waitcnt vmcnt(0), L2 write-back of non-coherent lines, then the store itself
ARCLE_DEV void release_store_system(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
ARCLE_DEV uint64_t clock() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz constant clock
// neighbouring lane's value through DPP wave shifts (no LDS): lane j-1 / lane j+1, 0 at the wave boundary
ARCLE_DEV uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
ARCLE_DEV uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
// lane j-K / j+K inside the lane's 16-lane DPP row (row_shr:K / row_shl:K), 0 where the source falls outside the row
template <int K>
ARCLE_DEV uint32_t row_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + K, 0xf, 0xf, true); }
template <int K>
ARCLE_DEV uint32_t row_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + K, 0xf, 0xf, true); }
// OR over the 64 lanes (DPP: row_shr 1, 2, 4, 8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31 across them), uniform result
ARCLE_DEV uint32_t wave_or(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// sum over the 64 lanes (same DPP ladder as wave_or), uniform result
ARCLE_DEV uint32_t wave_add(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// v_dot4_u32_u8: sum of the four byte products a.b[k] * b.b[k], plus c — turns four per-byte flags into a nibble in ONE
// instruction (weights 1, 2, 4, 8), see arcle::flags16
ARCLE_DEV uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// 16 B load from an arbitrary byte address (flattened state rows: plane segments start at any offset of the row); global memory
// takes unaligned vector accesses
ARCLE_DEV U4 load16u(const int8_t* p) {
  typedef U4 __attribute__((aligned(1))) U4a1;
  return *reinterpret_cast<const ARCLE_AS_GLOBAL U4a1*>((uintptr_t)p);
}
ARCLE_DEV U4 load16u_at(const void* base, uint32_t off) {  // (SGPR base + 32-bit VGPR byte offset)
  typedef U4 __attribute__((aligned(1))) U4a1;
  return *reinterpret_cast<const ARCLE_AS_GLOBAL U4a1*>((uintptr_t)base + off);
}
ARCLE_DEV uint32_t bfrev(uint32_t v) { return __builtin_bitreverse32(v); }  // v_bfrev_b32
ARCLE_DEV uint32_t readlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
ARCLE_DEV uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }  // full-rate 24-bit multiply
// signed 24-bit multiply of a scalar by a vector value, as the ONE instruction it is (the compiler masks operands it cannot bound)
ARCLE_DEV int mul24s(uint32_t s, int v) {
  int r;
  asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(v));
  return r;
}
ARCLE_DEV uint32_t opaque(uint32_t v) {  // hides a value's origin from the optimiser (no instruction)
  asm volatile("" : "+v"(v));
  return v;
}
// "these scalar-loaded values are needed now": one s_waitcnt for all of them instead of one per first use, and it keeps
// the compiler from sinking some of the loads below a branch on another (a second dependent latency)
ARCLE_DEV void arrived(U4& a, U2& b, uint32_t& c, U4& d) { asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d)); }
ARCLE_DEV void arrived3(U4& a, U2& b, uint32_t& c) { asm volatile("" : "+s"(a), "+s"(b), "+s"(c)); }
#ifndef ARCLE_STOP_AT
#define ARCLE_STOP_AT 0
#endif
#ifndef ARCLE_PACK_SPEC
#define ARCLE_PACK_SPEC 0  // 1: packed-row instantiations also request the grid plane beside the per-env scalar loads — every wave needs it for
                           // its row, yet it loses (c4 6.57 -> 6.62 us, hinted 6.21 -> 6.50: profiles/round4_experiments.txt); kept as a knob
#endif
#ifndef ARCLE_SPEC_SMALL_MAX
#define ARCLE_SPEC_SMALL_MAX 2048  // (0 = off) batches up to this size take the speculative grid load as well: one latency chain per launch
#endif
#ifndef ARCLE_STREAM_MIN_ENVS
#define ARCLE_STREAM_MIN_ENVS 34816  // batches from this size on (state 8 x N x 1 KiB beyond the 256 MiB Infinity Cache) take a streaming instantiation
#endif
// Reading back what this wave stored earlier needs no cache maintenance and no wait: a wave's vector memory operations reach its
// write-through L1 in issue order, so a load issued after a store to the same address returns the stored data (the guarantee
// every `a[i] = x; y = a[i];` relies on).  An agent-scope invalidate (`buffer_inv sc1`) here cost 50 us per launch: it empties the
// CU's L1 under the 31 other resident waves.  The asm is a compiler-level fence only (the plane stores are inline asm).
ARCLE_DEV void own_stores_visible() { asm volatile("" ::: "memory"); }
// "uniform value, vector register": v_perm_b32 with the identity selector has no scalar form, so the (still wave-uniform) result
// lives in a VGPR and everything computed from it is vector ALU work; a branch on such a value is v_cmp + s_cbranch_vccnz — no
// scalar instruction at all.  The CU's one scalar unit is what the step kernel saturates (profiles/round3_experiments.txt: a
// scalar instruction costs a launch 8-11 ns, the first ~40 extra vector instructions per wave nothing), so the per-env
// arithmetic that only feeds cell masks and LDS addresses is moved over with this.
ARCLE_DEV uint32_t tov(uint32_t x) { return __builtin_amdgcn_perm(x, x, 0x03020100u); }
// v_perm_b32: byte k of the result = byte sel.b[k] of the 8-byte table {hi, lo} (selector values 0-3: lo, 4-7: hi)
ARCLE_DEV uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// a volatile no-op on a scalar: the optimiser can neither speculate it nor fold the branch it sits in into a select — keeps a rare
// case a BRANCH, so that its arithmetic stays off the common path
ARCLE_DEV int rare_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
ARCLE_DEV int rare_v(int v) {  // (the same for a value held in a vector register)
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ void sink_s(uint32_t v) { asm volatile("" ::"s"(v)); }
__device__ __forceinline__ void sink_v(uint32_t v) { asm volatile("" ::"v"(v)); }
}  // namespace xl

#include "arcle_wave.h"
#include "arcle_big_params.h"  // grids beyond ARCLE_MAX_CELLS: one workgroup per env (arcle_big.hip)

using arcle::StepParams;
using arcle::WaveLDS;

#ifndef ARCLE_WAVES_PER_WG
#define ARCLE_WAVES_PER_WG 8  // 512-thread workgroups: 7.3 vs 7.7 us per launch of the C3 mix against 256 (in-box A/B, round 2)
#endif
static constexpr int WAVES_PER_WG = ARCLE_WAVES_PER_WG;
#ifndef ARCLE_SGPR_CAP
#define ARCLE_SGPR_CAP 80  // 8 waves per SIMD need <= 80 SGPRs per wave (MI355X_MICROARCH.md, residency)
#endif
typedef arcle::BlockLDS<WAVES_PER_WG> BlockLDS;

// wave index of the launch with XCD-contiguous ranges: workgroup b runs on XCD b%8 (observed), so
// vb = (b&7)*(nb/8) + (b>>3) gives each XCD one contiguous range of waves (affinity only)
// (waves_per_wg comes from the caller, not from blockDim: the workgroup size sits in the hidden kernel arguments as a 16-bit
// field, which costs a VECTOR load and its latency before the wave can even compute which env it owns)
__device__ __forceinline__ int wave_of_launch(int waves_per_wg = WAVES_PER_WG, uint32_t nb8 = gridDim.x >> 3, uint32_t boff = 0) {
  const uint32_t b = blockIdx.x - boff;  // the grid is a multiple of 8 workgroups; nb8 = (gridDim.x - boff) / 8
  const uint32_t vb = (b & 7u) * nb8 + (b >> 3);
  return __builtin_amdgcn_readfirstlane((int)(vb * (uint32_t)waves_per_wg + (threadIdx.x >> 6)));
}

// ---- dispatch order ---------------------------------------------------------------------------------------------------------------
// A launch of one wave per env ends with its last object operation: the hardware starts the 8192 waves over ~2 us (every XCD works
// through its workgroups in index order) and a Move / Rotate / Flip wave lives ~1.2 us longer than the others, so a launch whose late
// slots hold object ops ends ~0.6 us after one whose EARLY slots hold them (tools/lptbench.py: 5.33 vs 4.71 us).  Rounds 3-4 sorted the
// NEXT step's slots from a table written by the front workgroups of the previous launch (arcle_step_many, or single steps after
// arcle_hint_next_ops: 4.93 us, but only for callers who know their ops one step ahead).  Round 5: the launch orders ITSELF inside groups
// of ARCLE_GROUP_SIZE envs (the GROUPED block of arcle_step_kernel below: 4.90 us, no table, no hint) and the table form is gone.
#ifndef ARCLE_GROUP_SIZE
#define ARCLE_GROUP_SIZE 32  // envs (= dispatch strata) per group of a self-ordering launch
#endif

// ING: selection ingress form; FW: arcle::FW_* grid-width class;
// ACCT: 1 = add the step's algorithmic bytes to p.acct[env]; FEAT: 1 = carries the ARCLE_STEP_FEATURE_FLAGS code
// WC: 30 = the launch's grid is the standard 30 x 30 (H, W, P, plane stride and the division constant are compile-time
// constants: every clamp, row/column split and rectangle mask folds), 0 = read from the arguments
template <int ING, int FW, int ACCT, int FEAT, int FL = -1, int WC = 0>
__global__ __launch_bounds__(64 * WAVES_PER_WG) __attribute__((amdgpu_num_sgpr(ARCLE_SGPR_CAP))) void arcle_step_kernel(
    const int8_t* rec, const int32_t* cnt, const int32_t* op, const void* sel, const uint32_t* order, int n_envs, int wpw_front, uint32_t nb8,
    const StepParams pa) {
  // (wpw_front: waves per workgroup; self-ordering launches pass its log2)
  const int wpw = wpw_front;
  StepParams p = pa;  // (a register-promoted copy: only the fields a path reads are ever fetched)
  if (WC == 30) {
    p.H = p.W = 30;
    p.P = 900;
    p.PS = ARCLE_MAX_CELLS;
    p.div_magic = 65536u / 30u + 1u;
    p.nseg = 2;
  }
  if (FL >= 0) {  // the launch's flag set is a compile-time constant of this instantiation (the launcher checked that it applies)
    p.flags = (uint32_t)FL & 0xffffu;
    if (FL & ARCLE_STEP_FLAT_OBS) {  // ... and so are the shape of the fused observation rows: every segment offset folds
      p.flat_filter = (FL & ARCLE_STEPX_FLAT_FILTERED) ? 1 : 0;
      p.flat_tail = 0;
      if (WC == 30) p.flat_stride = (FL & ARCLE_STEPX_FLAT_FILTERED) ? ARCLE_ROW30_FILTERED_STRIDE : ARCLE_ROW30_FULL_STRIDE;
    }
  }
  // (leading scalar arguments = what a wave needs to find and request its env's inputs; built with -amdgpu-kernarg-preload-count they
  // are in SGPRs at wave start.  They repeat p.rec / p.cnt / p.op / p.sel / p.n_envs / p.wpw.)
  // (at the FRONT of the grid: the copy waves start first and their PCIe round trips run under the whole launch — 6.9 us per step at
  // 8192 envs; placed at the end of the grid they start last and the launch waits for them: 8.1 us, profiles/round3_experiments.txt)
  const bool pf_role = ING == arcle::INGRESS_BBOX5_PF && blockIdx.x < ARCLE_PF_BLOCKS;
  const uint32_t pf_first = 0, pf_off = ING == arcle::INGRESS_BBOX5_PF ? ARCLE_PF_BLOCKS : 0u;
  if (pf_role) {
    // The first workgroups of the launch are a copy engine: they move the NEXT step's action records from pinned host memory into the
    // device staging buffer that step will read (arcle_step_many over host-resident records).  The PCIe round trips of these few
    // waves run under the whole launch; the env waves of the next launch then find their records in device memory.
    const uint32_t n16 = ((uint32_t)n_envs * 20u) >> 4;  // whole 16-byte chunks (n_envs % 4 == 0, checked by the launcher)
    const uint32_t tid = (blockIdx.x - pf_first) * blockDim.x + threadIdx.x, nthr = ARCLE_PF_BLOCKS * blockDim.x;
    for (uint32_t i = tid; i < n16; i += 4u * nthr) {
      xl::U4 v[4];
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i + (uint32_t)k * nthr < n16) v[k] = xl::load16(reinterpret_cast<const int8_t*>(pa.next_sel), 16u * (i + (uint32_t)k * nthr));
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i + (uint32_t)k * nthr < n16) *reinterpret_cast<ARCLE_AS_GLOBAL xl::U4*>((uintptr_t)pa.stage_out + 16u * (i + (uint32_t)k * nthr)) = v[k];
    }
    return;
  }
  // one private 2 KiB tile pair per wave, sized by the launch (dynamic LDS: the step kernel has no workgroup-wide LDS state since the
  // expansion table went, so any workgroup size keeps the occupancy)
  extern __shared__ __attribute__((aligned(16))) unsigned char step_lds[];
  WaveLDS* const tiles = reinterpret_cast<WaveLDS*>(step_lds);
#ifdef ARCLE_TRACE_WAVES
  const uint64_t t_entry = xl::clock();
#endif
  if constexpr (FL >= 0 && (FL & ARCLE_STEPX_GROUPED) != 0) {
    // ---- the launch that orders itself (round 5) ------------------------------------------------------------------------------------
    // An XCD starts the workgroups of its slot range in index order, so the range falls into GS = 32 strata of G = rs / 32 consecutive slots
    // that start one after the other.  Group g of the XCD = the 32 slots {g + j G, j = 0..31} (one per stratum) and the 32 CONTIGUOUS envs
    // xbase + 32 g .. + 31.  Every wave of the group loads the inputs of ALL 32 envs — records, counters, tuples, op indices: one request of
    // the whole wave per array, 1.4 KB — ballots which ops are object operations (m, L = popc(m)) and applies one rule inside
    // the group: by default position j steps env j; the k-th object op found in a position >= L trades places with the k-th other op found
    // in a position < L.  All 32 waves compute the same permutation from the same 32 ops, so every env is stepped exactly once whatever the
    // ops are, and the slot's env is then picked out of the lanes with v_readlane: no table, no hint, no second round trip, no barrier.
    // Measured (profiles/round5_experiments.txt): 8192 envs 5.37 -> 4.90 us per launch (the table form with hints: 4.93; ops dealt in the
    // ideal order: 4.72); groups of 16: 5.00, of 64: 4.92; a group of 16 whose traded slots re-request their inputs (scalar loads): 5.25.
    // Preloaded arguments of these launches: `order` carries the op table's 64-bit object-op mask, `n_envs` the reciprocal of G
    // (floor(2^32 / G) + 1; every slot holds an env: n_envs % 256 == 0, checked by the launcher), `nb8` the slots per XCD and `wpw_front`
    // the log2 of the waves per workgroup.
    const uint64_t long_mask = (uint64_t)reinterpret_cast<uintptr_t>(order);
    const uint32_t magic = (uint32_t)n_envs, rs = nb8;
    constexpr uint32_t GS = ARCLE_GROUP_SIZE;
    static_assert(GS == 32, "the lane layout below is written for groups of 32");
    // (wave-uniform arithmetic on the vector ALUs — xl::tov — the CU's scalar unit is the short resource)
    const uint32_t vb = xl::tov(blockIdx.x);
    const uint32_t s_local = ((vb >> 3) << (uint32_t)wpw_front) + (threadIdx.x >> 6);
    const uint32_t j = __umulhi(s_local, magic);  // stratum of this slot = its position in the group
    // first env of the group: xcd rs + GS (s_local - j G), G = rs / GS
    const uint32_t gfirst = (uint32_t)xl::mul24s(rs, (int)(vb & 7u) - (int)j) + (s_local << 5);
    const uint32_t js = xl::uniform(j), gfirst_s = xl::uniform(gfirst);
    arcle::Wave w(p, &tiles[threadIdx.x >> 6], nullptr, (int)(threadIdx.x & 63), ING, FW, false, ACCT != 0, false);
    static_assert(ING != arcle::INGRESS_BBOX5_PF, "grouped launches: tuples, 5-tuple records, masks — not the record-prefetching form");
    constexpr bool REC5 = ING == arcle::INGRESS_BBOX5, CELLS = arcle::is_cells(ING);  // (masks / bit-packed masks: the payload is per cell — fetched for the slot's env once it is known)
    constexpr bool BY_LIMIT = (FL & ARCLE_STEP_TRUNCATE) != 0;  // (the research step: an env about to be re-initialised counts as long: its auto-reset is that kernel's longest wave)
    // the group's inputs: lanes 0-31 read the 32 op indices (the upper half repeats them); records (16 B per env), bbox tuples (16 B), point
    // tuples and counters (8 B) as ONE contiguous block per array spread over the 64 lanes — env e's item in lanes 2 e, 2 e + 1
    const uint32_t lane = threadIdx.x & 63u, e = lane & 31u;
    const uint32_t vop = REC5 ? xl::load32(sel, 20u * (gfirst + e) + 16u) : xl::load32(op, (gfirst + e) << 2);
    const xl::U2 vrec = xl::load8(rec, (gfirst << 4) + (lane << 3));
    xl::U4 vsel = {0u, 0u, 0u, 0u};
    if constexpr (CELLS) {
    } else if constexpr (REC5) vsel = xl::load16u_at(sel, 20u * (gfirst + e));  // (records are only dword aligned; per lane e)
    else if constexpr (ING == arcle::INGRESS_BBOX) {
      const xl::U2 t = xl::load8(sel, (gfirst << 4) + (lane << 3));
      vsel[0] = t[0], vsel[1] = t[1];
    } else vsel[0] = xl::load32(sel, (gfirst << 3) + (lane << 2));
    xl::U2 vcnt = {0u, 0u};
    if constexpr (BY_LIMIT) vcnt = xl::load8(cnt, (gfirst + e) << 3);  // (per lane e: the classification reads env e's step counter)
    else vcnt[0] = xl::load32(cnt, (gfirst << 3) + (lane << 2));
    // (the fetch of the argument block — plane bases, op table — is issued HERE, beside the loads above, not behind the wait for the group's ops)
    asm volatile("" ::"s"(pa.plane[ARCLE_PL_GRID]), "s"(pa.d_ops));
    bool lg = ((long_mask >> __builtin_elementwise_min(vop, 63u)) & 1ull) != 0ull;
    if constexpr (BY_LIMIT) lg = lg || (pa.step_limit > 0 && (int32_t)vcnt[0] == pa.step_limit - 1);
    const uint64_t m = xl::ballot(lg) & 0xffffffffull;
    const uint64_t hi = xl::ballot(lane >= (uint32_t)__builtin_popcountll(m));  // positions >= L, as a lane compare
    const uint64_t late_long = m & hi, early_other = ~(m | hi);                 // the two sides of the trade, k-th with k-th
    const uint32_t ra = __builtin_amdgcn_mbcnt_lo((uint32_t)late_long, 0u), rb = __builtin_amdgcn_mbcnt_lo((uint32_t)early_other, 0u);
    // code: what a position is (0x40 | rank: a late object op, 0x80 | rank: an early other op, 0x100 | lane: it keeps its env); want: the code of
    // the position whose env it steps (the k-th of the other side, or itself) — found with ONE ballot, no branch
    uint32_t code = 0x100u | lane;
    code = __builtin_amdgcn_inverse_ballot_w64(late_long) ? (0x40u | ra) : code;
    code = __builtin_amdgcn_inverse_ballot_w64(early_other) ? (0x80u | rb) : code;
    const uint32_t want = (code & 0x100u) ? code : (code ^ 0xc0u);
    const int pos = __builtin_ctzll(xl::ballot(code == xl::readlane(want, (int)js)));
    const int my_env = (int)gfirst_s + pos;
    // the env's scalars out of the lanes that hold them; an item that spans two lanes has its upper words moved to the even lane first (DPP),
    // so that ONE lane index serves every v_readlane of the array (no scalar index arithmetic)
    arcle::StepInputs in;
    const int h = pos << 1;
    in.rec[0] = xl::readlane(vrec[0], h);
    in.rec[1] = xl::readlane(vrec[1], h);
    in.rec[2] = xl::readlane(xl::quad_bcast_odd(vrec[0]), h);
    in.rec[3] = xl::readlane(xl::quad_bcast_odd(vrec[1]), h);
    if constexpr (CELLS) {
      in.payload = arcle::load_payload(w, my_env, 0, sel);
    } else if constexpr (REC5) {
#pragma unroll
      for (int k = 0; k < 4; k++) in.payload[k] = xl::readlane(vsel[k], pos);
    } else if constexpr (ING == arcle::INGRESS_BBOX) {
      in.payload[0] = xl::readlane(vsel[0], h);
      in.payload[1] = xl::readlane(vsel[1], h);
      in.payload[2] = xl::readlane(xl::quad_bcast_odd(vsel[0]), h);
      in.payload[3] = xl::readlane(xl::quad_bcast_odd(vsel[1]), h);
    } else {
      in.payload = arcle::u4_zero();
      in.payload[0] = xl::readlane(vsel[0], h);
      in.payload[1] = xl::readlane(xl::quad_bcast_odd(vsel[0]), h);
    }
    if constexpr (BY_LIMIT) {
      in.cnt[0] = xl::readlane(vcnt[0], pos);
      in.cnt[1] = xl::readlane(vcnt[1], pos);
    } else {
      in.cnt[0] = xl::readlane(vcnt[0], h);
      in.cnt[1] = xl::readlane(xl::quad_bcast_odd(vcnt[0]), h);
    }
    in.op = xl::readlane(vop, pos);
    arcle::wave_step<ING, FW, ACCT, FEAT, FL>(w, my_env, in, 0, 0, false, arcle::u4_zero());
#ifdef ARCLE_TRACE_WAVES  // (diagnostic builds, tools/launchtrace.py: when this wave entered and left, into the launch's half of the trace buffer)
    if (pa.acct && (threadIdx.x & 63u) == 0) {
      uint64_t* tr = reinterpret_cast<uint64_t*>(pa.acct) + 2 * ((size_t)(pa.n_steps & 1) * (size_t)pa.n_envs + (size_t)my_env);
      tr[0] = t_entry;
      tr[1] = xl::clock();
    }
#endif
    return;
  }
  const int wv = wave_of_launch(wpw, nb8, pf_off);
  const bool valid = wv < n_envs;  // (every wave of the workgroup reaches the barrier below)
  const int env = (int)__builtin_elementwise_min((uint32_t)wv, (uint32_t)n_envs - 1u);  // (surplus waves load env N-1's inputs and leave)
  arcle::Wave w(p, &tiles[threadIdx.x >> 6], nullptr, (int)(threadIdx.x & 63), ING, FW, false, ACCT != 0, false);  // (no expansion table: Wave::expand16)
  constexpr bool STREAM = FL >= 0 && (FL & ARCLE_STEPX_STREAM) != 0;
  w.store_nt = FL >= 0 && (FL & ARCLE_STEPX_STORE_NT) != 0;
  arcle::StepInputs in = arcle::load_inputs<ING>(w, env, rec, cnt, op, sel);  // in flight while the expansion table is built
  // streaming regime: the env's grid plane is requested NOW, beside the scalar inputs (its address needs nothing but the env index; the
  // plane's base travels in the preloaded `order` argument, which these launches do not use otherwise) — two of three steps of the O2ARC
  // mix read it, and with every wave waiting on HBM the bytes in flight per resident wave are what sets the rate
  arcle::U4 early_grid = arcle::u4_zero();
  bool early = false;
  if (STREAM) {
    const uint32_t goff = (uint32_t)env * (uint32_t)ARCLE_MAX_CELLS + 16u * (threadIdx.x & 63u);
    early_grid = (FL & ARCLE_STEPX_EARLY_NT) ? xl::load16_nt(reinterpret_cast<const int8_t*>(order), goff) : xl::load16(reinterpret_cast<const int8_t*>(order), goff);
    early = true;
  } else if (ARCLE_PACK_SPEC && FL >= 0 && (FL & ARCLE_STEP_PACK_OBS) && WC == 30) {
    // fused packed rows: EVERY wave needs the grid plane at its end (the row it packs), so the speculative request is never wasted —
    // and the epilogue no longer waits for a read-back
    early_grid = xl::load16(pa.plane[ARCLE_PL_GRID], (uint32_t)env * (uint32_t)ARCLE_MAX_CELLS + 16u * (threadIdx.x & 63u));
    early = true;
  } else if (WC == 0 && !ACCT && !FEAT && ING != arcle::INGRESS_BBOX5_PF) {
    // small batches of other grid shapes (at most a wave or two per SIMD: the launch is one wave's latency chain, nothing competes for
    // the memory pipes): the same speculative request, decided by the launcher (StepParams::spec_grid)
    if (pa.spec_grid) {
      if (w.live) early_grid = xl::load16(pa.plane[ARCLE_PL_GRID], (uint32_t)env * (uint32_t)pa.PS + 16u * (threadIdx.x & 63u));
      early = true;
    }
  }
  // (round 4: no expansion table, hence no workgroup barrier — every wave goes on as soon as its own inputs arrive: ordered launches
  // 5.04 -> 4.95 us, research step 10.0 -> 9.85; the always-true scalar test keeps a block boundary behind the loads, as the barrier did)
  if (wpw > 0) asm volatile("" ::: "memory");
  if (!valid) return;
#if ARCLE_STOP_AT == 1
  return;
#endif
#ifdef ARCLE_TRACE_WAVES
  arcle::wave_step<ING, FW, ACCT, FEAT, FL>(w, env, in, t_entry, xl::clock());
  if (!ACCT && pa.acct && (threadIdx.x & 63u) == 0) {  // (the lean kernels: entry / exit of the wave, as in the self-ordering branch)
    uint64_t* tr = reinterpret_cast<uint64_t*>(pa.acct) + 2 * ((size_t)(pa.n_steps & 1) * (size_t)pa.n_envs + (size_t)env);
    tr[0] = t_entry;
    tr[1] = xl::clock();
  }
#else
  arcle::wave_step<ING, FW, ACCT, FEAT, FL>(w, env, in, 0, 0, early, early_grid);
#endif
}

#ifndef ARCLE_ROLLOUT_WAVES
#define ARCLE_ROLLOUT_WAVES 4  // waves per SIMD the rollout kernels are compiled for (the register allocator's target: 512 / waves VGPRs)
#endif
template <int ING, int FW, int WC = 0, int FL = -1>
__global__ __launch_bounds__(64 * WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(ARCLE_ROLLOUT_WAVES))) void arcle_rollout_kernel(const StepParams pa) {
  StepParams p = pa;
  if (FL >= 0) p.flags = (uint32_t)FL;  // (the launcher checked that the launch's flags are this instantiation's constant)
  if (WC == 30) {  // the standard 30 x 30 grid: dimensions as compile-time constants (as in arcle_step_kernel)
    p.H = p.W = 30;
    p.P = 900;
    p.div_magic = 65536u / 30u + 1u;
    p.nseg = 2;
  }
  __shared__ BlockLDS lds;
  arcle::lut_init(lds.lut, (int)threadIdx.x);
  xl::wg_barrier();
  const int env = wave_of_launch();
  if (env >= p.n_envs) return;
  arcle::wave_rollout<ING, FW, FL>(p, &lds.wave[threadIdx.x >> 6], lds.lut, env, (int)(threadIdx.x & 63));
}

__global__ __launch_bounds__(64 * WAVES_PER_WG) void arcle_flatten_kernel(const StepParams p) {
  __shared__ BlockLDS lds;
  const int env = wave_of_launch();
  if (env >= p.n_envs) return;
  arcle::wave_flatten(p, &lds.wave[threadIdx.x >> 6], lds.lut, env, (int)(threadIdx.x & 63));
}

__global__ __launch_bounds__(64 * WAVES_PER_WG) void arcle_pack_kernel(const StepParams p) {
  __shared__ BlockLDS lds;
  const int env = wave_of_launch();
  if (env >= p.n_envs) return;
  arcle::wave_pack_obs(p, &lds.wave[threadIdx.x >> 6], lds.lut, env, (int)(threadIdx.x & 63));
}

// int8 [N][P] selection masks -> bit-packed uint8 [N][128] rows (the INGRESS_BITS form): bit f of a row = cell f truthy
__global__ __launch_bounds__(64 * WAVES_PER_WG) void arcle_pack_bits_kernel(const StepParams p, uint8_t* bits) {
  __shared__ BlockLDS lds;
  const int env = wave_of_launch();
  if (env >= p.n_envs) return;
  arcle::Wave w(p, &lds.wave[threadIdx.x >> 6], lds.lut, (int)(threadIdx.x & 63), arcle::INGRESS_MASK, arcle::FW_GENERIC, false);
  const arcle::U4 v = arcle::load_payload(w, env, 0, p.sel);
  const uint32_t m = arcle::nz16(v) & w.valid16;
  *reinterpret_cast<uint16_t*>(bits + (size_t)env * ARCLE_BITS_STRIDE + 2 * w.lane) = (uint16_t)m;
}

__global__ __launch_bounds__(64 * WAVES_PER_WG) void arcle_set_state_rows_kernel(const StepParams p) {
  __shared__ BlockLDS lds;
  const int env = wave_of_launch();
  if (env >= p.n_envs) return;
  arcle::wave_set_state_row(p, &lds.wave[threadIdx.x >> 6], lds.lut, env, (int)(threadIdx.x & 63));
}

template <int ING, int FW>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void arcle_transition_rows_kernel(const StepParams p) {
  __shared__ BlockLDS lds;
  arcle::lut_init(lds.lut, (int)threadIdx.x);
  xl::wg_barrier();
  const int row = wave_of_launch();
  if (row >= p.n_envs) return;
  arcle::wave_transition_row<ING, FW>(p, &lds.wave[threadIdx.x >> 6], lds.lut, row, (int)(threadIdx.x & 63));
}

__global__ __launch_bounds__(64 * WAVES_PER_WG) void arcle_reset_kernel(const StepParams p) {
  __shared__ BlockLDS lds;
  const int env = wave_of_launch();
  if (env >= p.n_envs) return;
  arcle::wave_reset(p, &lds.wave[threadIdx.x >> 6], lds.lut, env, (int)(threadIdx.x & 63));
}

__global__ __launch_bounds__(64 * WAVES_PER_WG) void arcle_reset_table_kernel(const StepParams p) {
  __shared__ BlockLDS lds;
  arcle::lut_init(lds.lut, (int)threadIdx.x);
  xl::wg_barrier();
  const int env = wave_of_launch();
  if (env >= p.n_envs) return;
  arcle::wave_reset_table(p, &lds.wave[threadIdx.x >> 6], lds.lut, env, (int)(threadIdx.x & 63));
}

// ------------------------------------------------------------------------------------------------
// host side: the C ABI
// ------------------------------------------------------------------------------------------------
// entry points run on the handle's device and leave the caller's current device untouched
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

struct LaunchPlan {  // how a step launch runs (see plan_launch)
  int policy;   // 0 | 'A' | 'B' | 'H' | 'J' (StepParams::spec_grid)
  int wpw;      // waves per workgroup
  int grouped;  // 1: a launch that orders itself (ARCLE_STEPX_GROUPED)
};

struct arcle_env {
  int big;  // H * W > ARCLE_MAX_CELLS: every launch of this handle goes to the workgroup-per-env kernels of arcle_big.hip
  int8_t* big_scratch;      // ... arcle_transition_rows of such a handle: scratch envs (planes, records, counters) for big_scratch_rows rows
  int32_t big_scratch_rows;
  arcle_config cfg;
  arcle_buffers bufs;
  bool owns_bufs;
  StepParams base;
  uint32_t* d_status;
  uint32_t* d_ops;
  uint64_t acct_extra = 0;  // algorithmic bytes of launches without a per-env counter (flattened observation rows)
  uint32_t ops_host[ARCLE_MAX_OPS];
  int8_t* flat_out;  // ARCLE_STEP_FLAT_OBS destination (arcle_set_flat_output)
  int8_t* pack_out;  // ARCLE_STEP_PACK_OBS destination (arcle_set_packed_output)
  int32_t flat_stride;
  int flat_filtered;
  int flat_tail;
  int flat_seq;  // arcle_set_flat_seq: the next rows' tails carry this sequence number behind a system-scope release (0: off)
  uint32_t* retired_ops[64];  // op tables replaced by arcle_set_op_table: launches in flight (and captured graphs) may still read them
  int n_retired;
  int32_t* d_stage;           // int32 [2][n_envs][5]: staging of host-resident action records (arcle_step_many), allocated on first use
  const int32_t* pf_next;     // set by arcle_step_many around a launch: the next step's host records / the staging buffer to fill
  int32_t* pf_stage;
  int pf_active;              // ... inside such a call (its first and last launches read records no front workgroup staged)
  int order_enabled;          // arcle_set_dispatch_order (default 1): launches of the standard batch order themselves (see the kernel)
  int stream_min;             // batches of at least this many envs take the streaming instantiations (ARCLE_STREAM_MIN_ENVS / env override)
  int spec_small_max;         // batches of at most this many envs request the grid plane speculatively (ARCLE_SPEC_SMALL_MAX env override)
  int stream_policy_override; // tuning runs: ARCLE_STREAM_POLICY = 0 | A | B | H | J for every batch size
  int group_enabled;          // tuning runs: ARCLE_GROUPED = 0 switches the self-ordering launches off for handles created under it
  int group_min, group_max;   // ... for batches of group_min .. group_max envs (ARCLE_GROUP_MIN / ARCLE_GROUP_MAX)
  int group_wpw;              // ... in workgroups of this many waves (ARCLE_GROUP_WPW)
  const LaunchPlan* forced;   // arcle_autotune timing a candidate
  int tuned_valid, tuned_ingress;   // arcle_autotune's choice for (ingress, flags) launches of this handle
  uint32_t tuned_flags;
  LaunchPlan tuned;
#ifdef ARCLE_TRACE_WAVES
  uint64_t* d_trace;          // diagnostic builds: uint64 [2][n_envs][2] entry / exit clocks of the waves of the last two launches
  int trace_seq;
#endif
  const void* ptr_seen[4];    // on_device(): the last action arrays asked about, and the answers
  bool ptr_dev[4];
  unsigned ptr_next;
  int dense_cache_live;       // the dense-pair cache may hold pairs: 1 a dense step ran since it was last dropped, 2 always assume so (a captured dense step)
  int wpw_override;           // tuning runs: waves per workgroup of the step launches (ARCLE_WPW = 1, 2, 4 or 8), 0 = the library's choice
  int32_t* d_dense_cache;     // int32 [n_envs][2]: dense pair of every env's current grid (allocated with the first dense output)
  uint32_t* d_acct;
  uint64_t acct_steps;
  int device;
  char err[256];
};

#define HIP_TRY(env, call)                                                                        \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      snprintf((env)->err, sizeof((env)->err), "%s failed: %s", #call, hipGetErrorString(e_));    \
      return ARCLE_ERR_HIP;                                                                       \
    }                                                                                             \
  } while (0)

static int fail(arcle_env* e, int code, const char* msg) {
  if (e) snprintf(e->err, sizeof(e->err), "%s", msg);
  return code;
}

extern "C" int arcle_abi_version(void) { return ARCLE_ABI_VERSION; }

extern "C" const char* arcle_last_error(const arcle_env* env) { return env ? env->err : "null handle"; }

extern "C" int arcle_create(const arcle_config* cfg, const arcle_buffers* bufs, arcle_env** out) {
  if (!cfg || !out) return ARCLE_ERR_ARG;
  *out = nullptr;
  if (cfg->n_envs <= 0 || cfg->H <= 0 || cfg->W <= 0 || cfg->H > 127 || cfg->W > 127 ||
      cfg->max_trial < -128 || cfg->max_trial > 127 ||
      (uint64_t)cfg->n_envs * (uint64_t)ARCLE_MAX_CELLS >= (1ull << 32))  // 32-bit plane offsets (one-wavefront kernels)
    return ARCLE_ERR_CONFIG;
  // more than ARCLE_MAX_CELLS cells (the reference takes any max_grid_size, base.py:37-49; dims are int8 there, hence <= 127): the
  // handle is served by the workgroup-per-env kernels (arcle_big.hip) — same ABI, the entry points below say which they do not have
  const bool big = cfg->H * cfg->W > ARCLE_MAX_CELLS;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ARCLE_ERR_NO_DEVICE;
  arcle_env* e = new (std::nothrow) arcle_env();
  if (!e) return ARCLE_ERR_ARG;
  memset(e, 0, sizeof(*e));
  e->order_enabled = 1;
  e->stream_min = ARCLE_STREAM_MIN_ENVS;
  if (const char* sm = getenv("ARCLE_STREAM_MIN_ENVS")) e->stream_min = atoi(sm);  // (tuning runs)
  if (const char* sp = getenv("ARCLE_STREAM_POLICY")) {
    if (sp[0] == '0' || sp[0] == 'A' || sp[0] == 'B' || sp[0] == 'H' || sp[0] == 'J') e->stream_policy_override = sp[0];
  }
  e->group_enabled = 1;
  e->group_min = 2049;
  e->group_max = 10240;
  e->group_wpw = 4;
  if (const char* gs = getenv("ARCLE_GROUPED")) e->group_enabled = atoi(gs) != 0;
  if (const char* gs = getenv("ARCLE_GROUP_MIN")) e->group_min = atoi(gs);
  if (const char* gs = getenv("ARCLE_GROUP_MAX")) e->group_max = atoi(gs);
  if (const char* gs = getenv("ARCLE_GROUP_WPW")) {
    const int v = atoi(gs);
    if (v == 1 || v == 2 || v == 4 || v == 8) e->group_wpw = v;
  }
  e->spec_small_max = ARCLE_SPEC_SMALL_MAX;
  if (const char* ss = getenv("ARCLE_SPEC_SMALL_MAX")) e->spec_small_max = atoi(ss);
  if (const char* wp = getenv("ARCLE_WPW")) {
    const int v = atoi(wp);
    if (v == 1 || v == 2 || v == 4 || v == 8) e->wpw_override = v;
  }
  e->cfg = *cfg;
  e->big = big ? 1 : 0;
  int caller_dev = 0;
  (void)hipGetDevice(&caller_dev);
  e->device = cfg->device >= 0 ? cfg->device : caller_dev;
  if (e->device >= ndev) {
    delete e;
    return ARCLE_ERR_NO_DEVICE;
  }
  DeviceGuard guard(e->device);  // allocations below land on the handle's device; the caller's device is restored
  StepParams& b = e->base;
  b.n_envs = cfg->n_envs;
  b.H = cfg->H;
  b.W = cfg->W;
  b.P = cfg->H * cfg->W;
  b.PS = cfg->plane_stride ? cfg->plane_stride : ARCLE_DEFAULT_PLANE_STRIDE(b.P);
  if ((b.PS & 15) || b.PS < b.P || b.PS > (big ? (int)arcle_big::MAX_PS : ARCLE_MAX_CELLS)) {
    delete e;
    return ARCLE_ERR_CONFIG;
  }
  b.max_trial = cfg->max_trial;
  b.div_magic = 65536u / (uint32_t)cfg->W + 1u;
  for (uint32_t n = 0; !big && n < ARCLE_MAX_CELLS + 16; n++)  // flat cell indices the kernel divides
    if (((n * b.div_magic) >> 16) != n / (uint32_t)cfg->W) {
      delete e;
      return ARCLE_ERR_CONFIG;
    }
  b.nseg = (cfg->W >= 16) ? 2 : 1 + (15 + cfg->W - 1) / cfg->W;
  const size_t plane_bytes = (size_t)cfg->n_envs * b.PS;
  if (bufs) {
    e->bufs = *bufs;
    e->owns_bufs = false;
    if (!bufs->plane[ARCLE_PL_INPUT] || !bufs->plane[ARCLE_PL_GRID] || !bufs->plane[ARCLE_PL_ANSWER] || !bufs->rec ||
        !bufs->cnt) {
      delete e;
      return ARCLE_ERR_ARG;
    }
    for (int i = 0; i < ARCLE_N_PLANES; i++)
      if ((reinterpret_cast<uintptr_t>(bufs->plane[i]) & 15) != 0) {
        delete e;
        return ARCLE_ERR_ARG;  // planes must be 16-byte aligned
      }
    if ((reinterpret_cast<uintptr_t>(bufs->rec) & 15) != 0) {
      delete e;
      return ARCLE_ERR_ARG;
    }
  } else {
    e->owns_bufs = true;
    for (int i = 0; i < ARCLE_N_PLANES; i++) {
      if (hipMalloc((void**)&e->bufs.plane[i], plane_bytes) != hipSuccess ||
          hipMemset(e->bufs.plane[i], 0, plane_bytes) != hipSuccess) {
        arcle_destroy(e);
        return ARCLE_ERR_HIP;
      }
    }
    if (hipMalloc((void**)&e->bufs.rec, (size_t)cfg->n_envs * ARCLE_REC_BYTES) != hipSuccess ||
        hipMemset(e->bufs.rec, 0, (size_t)cfg->n_envs * ARCLE_REC_BYTES) != hipSuccess ||
        hipMalloc((void**)&e->bufs.cnt, (size_t)cfg->n_envs * 8) != hipSuccess ||
        hipMemset(e->bufs.cnt, 0, (size_t)cfg->n_envs * 8) != hipSuccess) {
      arcle_destroy(e);
      return ARCLE_ERR_HIP;
    }
  }
  if (hipMalloc((void**)&e->d_status, 8) != hipSuccess || hipMemset(e->d_status, 0, 8) != hipSuccess ||
      hipMalloc((void**)&e->d_ops, sizeof(uint32_t) * (ARCLE_MAX_OPS + 1)) != hipSuccess ||  // (+1: slot n_ops is always an empty one)
      hipMemset(e->d_ops, 0, sizeof(uint32_t) * (ARCLE_MAX_OPS + 1)) != hipSuccess) {
    arcle_destroy(e);
    return ARCLE_ERR_HIP;
  }
  for (int i = 0; i < ARCLE_N_PLANES; i++) b.plane[i] = e->bufs.plane[i];
  b.rec = e->bufs.rec;
  b.cnt = e->bufs.cnt;
  b.status = e->d_status;
  b.d_ops = e->d_ops;
  b.n_ops = 0;
  *out = e;
  return ARCLE_OK;
}

extern "C" int arcle_destroy(arcle_env* e) {
  if (!e) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  if (e->owns_bufs) {
    for (int i = 0; i < ARCLE_N_PLANES; i++)
      if (e->bufs.plane[i]) (void)hipFree(e->bufs.plane[i]);
    if (e->bufs.rec) (void)hipFree(e->bufs.rec);
    if (e->bufs.cnt) (void)hipFree(e->bufs.cnt);
  }
  if (e->d_status) (void)hipFree(e->d_status);
  if (e->d_ops) (void)hipFree(e->d_ops);
  for (int i = 0; i < e->n_retired; i++) (void)hipFree(e->retired_ops[i]);
  if (e->d_dense_cache) (void)hipFree(e->d_dense_cache);
  if (e->d_stage) (void)hipFree(e->d_stage);
  if (e->d_acct) (void)hipFree(e->d_acct);
  if (e->big_scratch) (void)hipFree(e->big_scratch);
  delete e;
  return ARCLE_OK;
}

extern "C" int arcle_get_buffers(const arcle_env* e, arcle_buffers* out) {
  if (!e || !out) return ARCLE_ERR_ARG;
  *out = e->bufs;
  return ARCLE_OK;
}

// ---- handles of more than ARCLE_MAX_CELLS cells --------------------------------------------------------------------------------------
static arcle_big::BigParams big_params(const arcle_env* e) {
  const StepParams& b = e->base;
  arcle_big::BigParams q;
  memset(&q, 0, sizeof(q));
  for (int i = 0; i < ARCLE_N_PLANES; i++) q.plane[i] = b.plane[i];
  q.rec = b.rec;
  q.cnt = b.cnt;
  q.n_envs = b.n_envs;
  q.H = b.H;
  q.W = b.W;
  q.P = b.P;
  q.PS = b.PS;
  q.n_ops = b.n_ops;
  q.max_trial = b.max_trial;
  q.step_limit = b.step_limit;
  q.status = b.status;
  q.d_ops = b.d_ops;
  q.trunc = b.trunc;
  q.tbl_in = b.tbl_in;
  q.tbl_ans = b.tbl_ans;
  q.tbl_in_dim = b.tbl_in_dim;
  q.tbl_ans_dim = b.tbl_ans_dim;
  q.n_tasks = b.n_tasks;
  q.seed = b.seed;
  q.env_base = b.env_base;
  q.episode = b.episode;
  q.cur_task = b.cur_task;
  q.pair_off = b.pair_off;
  q.pair_cnt = b.pair_cnt;
  q.n_problems = b.n_problems;
  q.aug_flags = b.aug_flags;
  return q;
}
static int big_done(arcle_env* e, int hip_rc, const char* what) {
  if (hip_rc != 0) {
    snprintf(e->err, sizeof(e->err), "%s failed: %s", what, hipGetErrorString((hipError_t)hip_rc));
    return ARCLE_ERR_HIP;
  }
  return ARCLE_OK;
}
extern "C" int arcle_set_op_table(arcle_env* e, const uint32_t* descs, int32_t n_ops) {
  if (!e || !descs) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  if (n_ops <= 0 || n_ops > ARCLE_MAX_OPS) return fail(e, ARCLE_ERR_CONFIG, "n_ops out of range");
  for (int i = 0; i < n_ops; i++) {
    const uint32_t k = ARCLE_OP_KIND(descs[i]), f = ARCLE_OP_FLAGS(descs[i]), a = ARCLE_OP_ARG(descs[i]);
    if (k >= ARCLE_N_OP_KINDS) return fail(e, ARCLE_ERR_CONFIG, "unknown op kind");
    const bool need_sel = (f & (ARCLE_OPF_RESET_SEL | ARCLE_OPF_KEEP_SEL)) != 0;
    const bool need_obj = (k == ARCLE_OP_MOVE || k == ARCLE_OP_ROTATE || k == ARCLE_OP_FLIP);
    const bool need_clip = (k == ARCLE_OP_COPY || k == ARCLE_OP_PASTE);
    if ((need_sel || need_obj) && !e->bufs.plane[ARCLE_PL_SELECTED])
      return fail(e, ARCLE_ERR_CONFIG, "op table needs the `selected` plane");
    if (need_obj && !(e->bufs.plane[ARCLE_PL_OBJECT] && e->bufs.plane[ARCLE_PL_OBJECT_SEL] &&
                      e->bufs.plane[ARCLE_PL_BACKGROUND]))
      return fail(e, ARCLE_ERR_CONFIG, "op table needs the object planes");
    if (need_clip && !e->bufs.plane[ARCLE_PL_CLIP]) return fail(e, ARCLE_ERR_CONFIG, "op table needs the `clip` plane");
    if ((k == ARCLE_OP_MOVE && a > 3) || (k == ARCLE_OP_ROTATE && (a < 1 || a > 3)) || (k == ARCLE_OP_FLIP && a > 3) ||
        (k == ARCLE_OP_COPY && a > 1) || (k == ARCLE_OP_PASTE && a > 1))
      return fail(e, ARCLE_ERR_CONFIG, "op argument out of range");
  }
  // Every launch receives its parameters BY VALUE (StepParams is a kernel argument), the table pointer included: launches already
  // enqueued — and hipGraphs already captured — keep reading the table they were launched / captured with.  So a new table goes into
  // a NEW device buffer and the old one is only retired (freed in arcle_destroy); no synchronisation, nothing in flight is disturbed.
  uint32_t* fresh = nullptr;
  if (e->base.n_ops > 0) {
    if (e->n_retired >= (int)(sizeof(e->retired_ops) / sizeof(e->retired_ops[0]))) {
      HIP_TRY(e, hipDeviceSynchronize());  // (64 replacements later: recycle — nothing can still be reading the oldest ones)
      for (int i = 0; i < e->n_retired; i++) (void)hipFree(e->retired_ops[i]);
      e->n_retired = 0;
    }
    HIP_TRY(e, hipMalloc((void**)&fresh, sizeof(uint32_t) * (ARCLE_MAX_OPS + 1)));
  }
  memset(e->ops_host, 0, sizeof(e->ops_host));
  memcpy(e->ops_host, descs, sizeof(uint32_t) * (size_t)n_ops);
  uint32_t staged[ARCLE_MAX_OPS + 1];
  memset(staged, 0, sizeof(staged));  // (slot n_ops .. ARCLE_MAX_OPS stay empty)
  memcpy(staged, descs, sizeof(uint32_t) * (size_t)n_ops);
  uint32_t* dst = fresh ? fresh : e->d_ops;
  HIP_TRY(e, hipMemcpy(dst, staged, sizeof(staged), hipMemcpyHostToDevice));  // synchronous: `staged` may go out of scope
  if (fresh) {
    e->retired_ops[e->n_retired++] = e->d_ops;
    e->d_ops = fresh;
    e->base.d_ops = fresh;
  }
  e->base.n_ops = n_ops;
  e->base.long_mask = 0;  // the operations whose waves run longest (self-ordering launches deal them to the slots that start first)
  for (int i = 0; i < n_ops && i < 63; i++) {
    const uint32_t k = ARCLE_OP_KIND(descs[i]);
    if (k == ARCLE_OP_MOVE || k == ARCLE_OP_ROTATE || k == ARCLE_OP_FLIP) e->base.long_mask |= 1ull << i;
  }
  return ARCLE_OK;
}

extern "C" int arcle_set_task_table(arcle_env* e, const int8_t* in_planes, const int8_t* in_dims,
                                    const int8_t* ans_planes, const int8_t* ans_dims, int32_t n_tasks) {
  if (!e || !in_planes || !in_dims || !ans_planes || !ans_dims) return ARCLE_ERR_ARG;
  if (n_tasks <= 0) return fail(e, ARCLE_ERR_CONFIG, "empty task table");
  if ((reinterpret_cast<uintptr_t>(in_planes) & 15) || (reinterpret_cast<uintptr_t>(ans_planes) & 15))
    return fail(e, ARCLE_ERR_ARG, "task table planes must be 16-byte aligned");
  // (an entry's two dims are read with ONE aligned scalar dword load, arcle::load_task: the dims arrays start on a 4-byte boundary and their
  // allocation covers 2 * n_tasks rounded up to a multiple of 4 bytes)
  if ((reinterpret_cast<uintptr_t>(in_dims) & 3) || (reinterpret_cast<uintptr_t>(ans_dims) & 3))
    return fail(e, ARCLE_ERR_ARG, "task table dims arrays must be 4-byte aligned (and allocated in whole dwords)");
  // (the table stays caller-owned and must outlive its last use; what CAN be checked here: all four are device-accessible memory)
  for (const void* ptr : {(const void*)in_planes, (const void*)in_dims, (const void*)ans_planes, (const void*)ans_dims}) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)) {
      (void)hipGetLastError();
      return fail(e, ARCLE_ERR_ARG, "task table arrays must be device memory");
    }
  }
  e->base.tbl_in = in_planes;
  e->base.tbl_in_dim = in_dims;
  e->base.tbl_ans = ans_planes;
  e->base.tbl_ans_dim = ans_dims;
  e->base.n_tasks = n_tasks;
  return ARCLE_OK;
}

extern "C" int arcle_can_elide_selected(const arcle_env* e) {
  if (!e || e->base.n_ops <= 0) return 0;
  for (int i = 0; i < e->base.n_ops; i++)
    if (ARCLE_OP_FLAGS(e->ops_host[i]) & ARCLE_OPF_KEEP_SEL) return 0;
  return e->bufs.plane[ARCLE_PL_SELECTED] != nullptr;
}

static dim3 grid_for(int n_envs, int waves_per_wg = WAVES_PER_WG);

extern "C" int arcle_reset_from_table(arcle_env* e, const int32_t* task_idx, const uint8_t* mask, void* stream) {
  if (!e || !task_idx) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  if (e->base.n_tasks <= 0) return fail(e, ARCLE_ERR_CONFIG, "no task table installed (arcle_set_task_table)");
  if (e->big) {
    arcle_big::BigParams q = big_params(e);
    q.rmask = mask;
    q.task_idx = task_idx;
    return big_done(e, arcle_big::launch_reset(q, 1, stream), "arcle_reset_from_table");
  }
  StepParams p = e->base;
  p.rmask = mask;
  p.task_idx = task_idx;
  hipLaunchKernelGGL(arcle_reset_table_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, (hipStream_t)stream, p);
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

static dim3 grid_for(int n_envs, int waves_per_wg) {
  unsigned nb = (unsigned)((n_envs + waves_per_wg - 1) / waves_per_wg);
  nb = (nb + 7u) & ~7u;
  return dim3(nb);
}

extern "C" int arcle_reset(arcle_env* e, const uint8_t* mask, void* stream) {
  if (!e) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  if (e->big) {
    arcle_big::BigParams q = big_params(e);
    q.rmask = mask;
    return big_done(e, arcle_big::launch_reset(q, 0, stream), "arcle_reset");
  }
  StepParams p = e->base;
  p.rmask = mask;
  hipLaunchKernelGGL(arcle_reset_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, (hipStream_t)stream, p);
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

static int launch_flatten(arcle_env* e, int8_t* out, int32_t out_stride, int filtered, hipStream_t st);

// ---- instantiation dispatch: (ingress, width class, flags, accounting) -> kernel --------------------------------------
static int width_class(const StepParams& p) {
  if (p.W < 16 || p.W > 32) return arcle::FW_GENERIC;
  return p.PS == ARCLE_MAX_CELLS ? arcle::FW_FULL : arcle::FW_FAST;
}
#define STEP_ARGS (const int8_t*)p.rec, (const int32_t*)p.cnt, p.op, p.sel, (const uint32_t*)nullptr, p.n_envs, p.wpw, g.x >> 3, p
#define STEP_LDS(b) ((size_t)((b).x / 64u) * sizeof(arcle::WaveLDS))
#define LAUNCH_STEP(...) hipLaunchKernelGGL((arcle_step_kernel<__VA_ARGS__>), g, b, STEP_LDS(b), st, STEP_ARGS)
// a launch that orders itself: the preloaded `order` argument carries the object-op mask, `n_envs` the reciprocal of the group count
#define LAUNCH_GROUPED(INGV, FWV, FLSET, FEATV)                                                                                             \
  hipLaunchKernelGGL((arcle_step_kernel<INGV, FWV, 0, FEATV, (FLSET) | ARCLE_STEPX_GROUPED, 30>), g, b, STEP_LDS(b), st, (const int8_t*)p.rec,    \
                     (const int32_t*)p.cnt, p.op, p.sel, reinterpret_cast<const uint32_t*>((uintptr_t)p.long_mask), (int)p.group_magic, __builtin_ctz((unsigned)p.wpw), \
                     (uint32_t)p.n_envs >> 3, p)
// the flag combination ARCVecEnv steps with (next-step autoreset, elided zero-fill of `selected`) has its own instantiation
// with the flags as a compile-time constant
static constexpr int HOT_FLAGS = ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED;
// ... and the same plus the fused packed row (what a ShardedVecEnv steps with): lean 30 x 30 instantiation
static constexpr int HOT_PACK_FLAGS = HOT_FLAGS | ARCLE_STEP_PACK_OBS;
// ... and the research env's step (agents/env.py:23-58 + agents/train.py:61-68 as ARCVecEnv(autoreset="resample", dense_reward,
// max_episode_steps) runs it): episode end -> new device-drawn task, TimeLimit, dense reward pair, fused FilterO2ARC rows
static constexpr int RESEARCH_FLAGS = ARCLE_STEP_ELIDE_SELECTED | ARCLE_STEP_TRUNCATE | ARCLE_STEP_RESAMPLE | ARCLE_STEP_DENSE | ARCLE_STEP_FLAT_OBS;
static constexpr int RESEARCH_FL = RESEARCH_FLAGS | ARCLE_STEPX_FLAT_FILTERED;
static constexpr int RESEARCH_INC_FL = RESEARCH_FL | ARCLE_STEP_ROWS_INCREMENTAL;  // ... with incremental rows (what ARCVecEnv runs)
static bool research_shape(const StepParams& p, uint32_t extra = 0) {
  return p.flags == ((uint32_t)RESEARCH_FLAGS | extra) && p.flat_filter == 1 && p.flat_tail == 0 && p.flat_stride == ARCLE_ROW30_FILTERED_STRIDE;
}
#ifdef ARCLE_FAST_BUILD  // development builds: only the benchmark's instantiations exist (seconds instead of minutes)
template <int ING>
static int launch_step_ing(int, bool acct, bool feat, dim3 g, dim3 b, hipStream_t st, const StepParams& p) {
  if (ING != arcle::INGRESS_BBOX || width_class(p) != arcle::FW_FULL || p.H != 30 || p.W != 30) return ARCLE_ERR_CONFIG;
  if (feat || acct) {
    if (!acct && research_shape(p)) LAUNCH_STEP(arcle::INGRESS_BBOX, arcle::FW_FULL, 0, 1, RESEARCH_FL, 30);
    else if (!acct && research_shape(p, ARCLE_STEP_ROWS_INCREMENTAL)) LAUNCH_STEP(arcle::INGRESS_BBOX, arcle::FW_FULL, 0, 1, RESEARCH_INC_FL, 30);
    else LAUNCH_STEP(arcle::INGRESS_BBOX, arcle::FW_FULL, 1, 1);
  } else if (p.flags == (uint32_t)HOT_FLAGS && p.group_magic) {
    LAUNCH_GROUPED(arcle::INGRESS_BBOX, arcle::FW_FULL, HOT_FLAGS, 0);
  } else if (p.flags == (uint32_t)HOT_FLAGS && p.spec_grid) {
#define LAUNCH_STREAM(BITS)                                                                                                                    \
  hipLaunchKernelGGL((arcle_step_kernel<arcle::INGRESS_BBOX, arcle::FW_FULL, 0, 0, HOT_FLAGS | ARCLE_STEPX_STREAM | (BITS), 30>), g, b, STEP_LDS(b), st, (const int8_t*)p.rec, \
                     (const int32_t*)p.cnt, p.op, p.sel, (const uint32_t*)p.plane[ARCLE_PL_GRID], p.n_envs, p.wpw, g.x >> 3, p)
    switch (p.spec_grid) {
      case 'B': LAUNCH_STREAM(ARCLE_STEPX_STORE_NT); break;
      case 'H': LAUNCH_STREAM(ARCLE_STEPX_EARLY_NT); break;
      case 'J': LAUNCH_STREAM(ARCLE_STEPX_STORE_NT | ARCLE_STEPX_EARLY_NT); break;
      default: LAUNCH_STREAM(0); break;
    }
#undef LAUNCH_STREAM
  } else if (p.flags == (uint32_t)HOT_FLAGS) LAUNCH_STEP(arcle::INGRESS_BBOX, arcle::FW_FULL, 0, 0, HOT_FLAGS, 30);
  else LAUNCH_STEP(arcle::INGRESS_BBOX, arcle::FW_FULL, 0, 0, -1, 30);
  return ARCLE_OK;
}
#else
template <int ING, int FW>
static void launch_step_tbl(bool acct, bool feat, dim3 g, dim3 b, hipStream_t st, const StepParams& p) {
  if constexpr (FW == arcle::FW_FULL) {  // the standard 30 x 30 grid: lean instantiations with the dimensions as compile-time constants
    if (p.H == 30 && p.W == 30 && !acct) {
      if constexpr (ING == arcle::INGRESS_BBOX5) {
        if (p.flags == (uint32_t)HOT_FLAGS && p.next_sel && p.wpw == WAVES_PER_WG) {  // records prefetched by the launch's front workgroups
          const dim3 gp(g.x + ARCLE_PF_BLOCKS);
          hipLaunchKernelGGL((arcle_step_kernel<arcle::INGRESS_BBOX5_PF, FW, 0, 0, HOT_FLAGS, 30>), gp, b, STEP_LDS(b), st, (const int8_t*)p.rec,
                             (const int32_t*)p.cnt, p.op, p.sel, (const uint32_t*)nullptr, p.n_envs, p.wpw, g.x >> 3, p);
          return;
        }
      }
      if (p.group_magic && p.flags == (uint32_t)HOT_FLAGS) { LAUNCH_GROUPED(ING, FW, HOT_FLAGS, 0); return; }  // (every ingress form)
      if constexpr (ING == arcle::INGRESS_BBOX || ING == arcle::INGRESS_BBOX5 || ING == arcle::INGRESS_POINT) {
        // ... and without the auto-reset: what ARCVecEnv steps with by default (the reference has none: a terminated env keeps being mutated)
        if (p.group_magic && p.flags == (uint32_t)ARCLE_STEP_ELIDE_SELECTED) { LAUNCH_GROUPED(ING, FW, ARCLE_STEP_ELIDE_SELECTED, 0); return; }
      }
      if constexpr (ING == arcle::INGRESS_BBOX || ING == arcle::INGRESS_BBOX5) {
        if (p.group_magic && p.flags == (uint32_t)HOT_PACK_FLAGS) { LAUNCH_GROUPED(ING, FW, HOT_PACK_FLAGS, 0); return; }
        if (p.group_magic && research_shape(p, ARCLE_STEP_ROWS_INCREMENTAL)) { LAUNCH_GROUPED(ING, FW, RESEARCH_INC_FL, 1); return; }
      }
      if (p.flags == (uint32_t)HOT_PACK_FLAGS) { LAUNCH_STEP(ING, FW, 0, 0, HOT_PACK_FLAGS, 30); return; }
      if constexpr (ING == arcle::INGRESS_BBOX || ING == arcle::INGRESS_BBOX5) {
        if (p.flags == (uint32_t)HOT_FLAGS && p.spec_grid) {  // speculative grid request: the plane's base rides in the preloaded `order` argument
#define LAUNCH_STREAM(BITS)                                                                                                                    \
  hipLaunchKernelGGL((arcle_step_kernel<ING, FW, 0, 0, HOT_FLAGS | ARCLE_STEPX_STREAM | (BITS), 30>), g, b, STEP_LDS(b), st, (const int8_t*)p.rec, (const int32_t*)p.cnt, \
                     p.op, p.sel, (const uint32_t*)p.plane[ARCLE_PL_GRID], p.n_envs, p.wpw, g.x >> 3, p)
          switch (p.spec_grid) {
            case 'B': LAUNCH_STREAM(ARCLE_STEPX_STORE_NT); break;
            case 'H': LAUNCH_STREAM(ARCLE_STEPX_EARLY_NT); break;
            case 'J': LAUNCH_STREAM(ARCLE_STEPX_STORE_NT | ARCLE_STEPX_EARLY_NT); break;
            default: LAUNCH_STREAM(0); break;
          }
#undef LAUNCH_STREAM
          return;
        }
      }
      if (p.flags == (uint32_t)HOT_FLAGS) { LAUNCH_STEP(ING, FW, 0, 0, HOT_FLAGS, 30); return; }
      if (research_shape(p)) { LAUNCH_STEP(ING, FW, 0, 1, RESEARCH_FL, 30); return; }
      if (research_shape(p, ARCLE_STEP_ROWS_INCREMENTAL)) { LAUNCH_STEP(ING, FW, 0, 1, RESEARCH_INC_FL, 30); return; }
      if (!feat) { LAUNCH_STEP(ING, FW, 0, 0, -1, 30); return; }
    }
  }
  // the feature instantiation also carries the byte accounting (it adds up scalars; stored only when the handle has a counter buffer)
  if (feat || acct) { LAUNCH_STEP(ING, FW, 1, 1); return; }
  if constexpr (FW != arcle::FW_GENERIC) {  // (generic widths have no compile-time-flag twin: one instantiation less per ingress form)
    if (p.flags == (uint32_t)HOT_FLAGS) { LAUNCH_STEP(ING, FW, 0, 0, HOT_FLAGS); return; }
  }
  LAUNCH_STEP(ING, FW, 0, 0);
}
template <int ING>
static int launch_step_ing(int fw, bool acct, bool feat, dim3 g, dim3 b, hipStream_t st, const StepParams& p) {
  if (fw == arcle::FW_FULL) launch_step_tbl<ING, arcle::FW_FULL>(acct, feat, g, b, st, p);
  else if (fw == arcle::FW_FAST) launch_step_tbl<ING, arcle::FW_FAST>(acct, feat, g, b, st, p);
  else launch_step_tbl<ING, arcle::FW_GENERIC>(acct, feat, g, b, st, p);
  return ARCLE_OK;
}
#endif

// which (ingress, flag set) combinations have a self-ordering instantiation (30 x 30, FW_FULL, no accounting)
static bool grouped_instantiation(int ingress, const StepParams& p) {
#ifdef ARCLE_FAST_BUILD
  return ingress == arcle::INGRESS_BBOX && p.flags == (uint32_t)HOT_FLAGS;
#else
  const bool tuple5 = ingress == arcle::INGRESS_BBOX || ingress == arcle::INGRESS_BBOX5;
  if (p.flags == (uint32_t)HOT_FLAGS) return true;  // (tuples, records, int8 and bit-packed masks)
  if (p.flags == (uint32_t)ARCLE_STEP_ELIDE_SELECTED) return tuple5 || ingress == arcle::INGRESS_POINT;  // (ARCVecEnv without autoreset)
  if (p.flags == (uint32_t)HOT_PACK_FLAGS) return tuple5;
  return tuple5 && research_shape(p, ARCLE_STEP_ROWS_INCREMENTAL);
#endif
}

// ... and whether a launch of this handle with these parameters (flags, row shape already filled in) CAN take it (whether it does: plan_launch)
static bool grouped_applies(const arcle_env* e, int ingress, const StepParams& p) {
  return e->group_enabled && e->order_enabled && p.H == 30 && p.W == 30 && p.PS == ARCLE_MAX_CELLS && !e->d_acct && e->base.long_mask != 0 &&
         (p.n_envs % (8 * ARCLE_GROUP_SIZE)) == 0 && p.n_envs >= 16 * ARCLE_GROUP_SIZE && grouped_instantiation(ingress, p);
}

// A self-ordering launch reads the actions of 32 envs per wave: fine from device memory (one wave's request serves the whole group out of
// L2), 32 x the PCIe traffic for a payload in pinned host memory (ARCVecEnv.step_bbox5 accepts one) — those keep the scalar per-env loads
// (asked once per array: a loop that steps out of the same action buffers finds its last answers in a four-entry cache of the handle; a
// wrong answer — an address freed and handed out again as the other kind of memory — costs speed, never correctness: the kernels read both)
static bool on_device(arcle_env* e, const void* ptr) {
  for (int i = 0; i < 4; i++)
    if (e->ptr_seen[i] == ptr && ptr) return e->ptr_dev[i];
  hipPointerAttribute_t attr;
  bool dev = false;
  if (hipPointerGetAttributes(&attr, ptr) == hipSuccess) dev = attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
  else (void)hipGetLastError();
  e->ptr_seen[e->ptr_next & 3] = ptr;
  e->ptr_dev[e->ptr_next & 3] = dev;
  e->ptr_next++;
  return dev;
}

// Which batches request the grid plane speculatively, and with which cache policies (profiles/round4_experiments.txt; sweeps in
// profiles/round4_stream_policy_sweep*.txt: us per launch of the C3 mix, same box, action stream cache-resident, plain kernel -> policy):
//   'A' spec, write-through stores          N <= 2048: the launch is ONE wave's latency chain          3.82 -> 3.68 (1024 envs)
//   'B' spec, non-temporal stores            state beyond the 256 MiB Infinity Cache                    19.0 -> 17.1 (36 864), 29.2 -> 21.2 (49 152), 37.4 -> 26.2 (65 536),
//                                                                                                       55.4 -> 40.9 (98 304)
//   'H' non-temporal spec, write-through     ~1 GB of state                                             65.5 -> 57.2 (114 688), 77.3 -> 67.0 (131 072)
//   'J' non-temporal spec, nt stores         multi-GB state                                             97.2 -> 82.9 (163 840), 117 -> 99.2 (196 608), 228 -> 199 (393 216)
// 4096-32768 envs (state inside the cache: what a launch writes through is what the next one finds there) keep the plain kernel — every
// variant loses there (32 768: 14.0 plain, 15.5 B, 16.5 H).  With an action stream that itself streams from HBM (hundreds of distinct
// action batches) the windows shift (B already wins at 32 768, H from 81 920): the table is set for a policy that writes one batch per step.
static int stream_policy(const arcle_env* e, int n) {
  if (e->stream_policy_override) return e->stream_policy_override == '0' ? 0 : e->stream_policy_override;
  if (n <= e->spec_small_max) return 'A';
  // tables without object operations (ARCEnv / RawARCEnv: Color, FloodFill, Copy, Paste, Submit ...): nearly every op reads the grid, so
  // the request is almost never wasted, and no long Move / Rotate / Flip wave hides it — c5 (4096 envs, 70 % flood fills) 5.81 -> 5.59 us
  if (e->base.long_mask == 0 && n <= 8192) return 'A';
  if (e->stream_min <= 0 || n < e->stream_min) return 0;
  if (n < 110592) return 'B';
  if (n < 155648) return 'H';
  return 'J';
}

// workgroups of 8 waves while the batch is one occupancy round or two (7.3 vs 7.7 us per launch at 8192 envs), 4 waves in the streaming
// regime (76-79 vs 85-88 us at 131072 envs; in-box A/B, profiles/archive/round2_experiments.txt) and for batches of at most two waves per SIMD
// (256-thread workgroups spread them over twice the CUs: c2 3.85 -> 3.69 us, profiles/round4_experiments.txt)
static int launch_wpw(const arcle_env* e) {
  if (e->wpw_override) return e->wpw_override;
  return (e->cfg.n_envs >= 65536 || e->cfg.n_envs <= e->spec_small_max) ? 4 : WAVES_PER_WG;
}

// How a step launch of this handle runs: cache policy of the speculative grid request, workgroup size, self-ordering or not.  The library's
// tables (stream_policy, launch_wpw, the grouping window) were measured on one box with one action-stream regime; arcle_autotune replaces
// them, per handle, by what it measured on THIS handle's size, box and action residency.
// p: the launch's parameters with flags and row shape filled in; device_payload: the actions live in device memory
static LaunchPlan plan_launch(const arcle_env* e, int ingress, const StepParams& p, bool device_payload) {
  const uint32_t flags = p.flags;
  const bool std30 = p.H == 30 && p.W == 30 && p.PS == ARCLE_MAX_CELLS;
  const bool tuple5 = ingress == arcle::INGRESS_BBOX || ingress == arcle::INGRESS_BBOX5;
  // which choices exist for this launch at all
  const bool can_group = device_payload && !e->pf_active && grouped_applies(e, ingress, p);
  const bool any_policy = std30 && flags == (uint32_t)HOT_FLAGS && tuple5;                      // the lean streaming instantiations
  const bool policy_a = !std30 && !(flags & ARCLE_STEP_FEATURE_FLAGS) && ingress != arcle::INGRESS_MASK;  // other shapes: the run-time request
  LaunchPlan pl;
  if (e->forced) {  // (arcle_autotune timing a candidate)
    pl = *e->forced;
  } else if (e->tuned_valid && e->tuned_ingress == ingress && e->tuned_flags == flags) {
    pl = e->tuned;
  } else {
    pl.policy = stream_policy(e, p.n_envs);
    pl.wpw = launch_wpw(e);
    // ... and for a batch of at most one occupancy round whose launch has no front workgroups to carry (no records to prefetch): 5.40 -> 5.35 us
    // at 8192 envs, 4.28 -> 4.25 at 4096; from 16384 envs on 8 waves are the better shape (profiles/round4_experiments.txt §9)
    if (!e->wpw_override && pl.wpw == WAVES_PER_WG && p.n_envs <= 8192 && !e->pf_next) pl.wpw = 4;
    // ... round 5, the lean bbox / record kernel between 8192 and 65536 envs as well: 4-wave workgroups 2-4 % ahead of 8 at 12 288 … 49 152
    // envs whatever the policy (profiles/round5_experiments.txt §4c: e.g. 32 768 envs plain 13.2 vs 13.7 us, 40 960 B 17.4 vs 17.6)
    if (!e->wpw_override && any_policy && !e->pf_next) pl.wpw = 4;
    // self-ordering launches inside the window they were measured to win in with a cache-resident action stream (profiles/round5_experiments.txt)
    pl.grouped = p.n_envs >= e->group_min && p.n_envs <= e->group_max;
    if (pl.grouped && can_group && !e->wpw_override) pl.wpw = e->group_wpw;
  }
  if (!can_group) pl.grouped = 0;
  if (pl.grouped) pl.policy = 0;
  else if (!(any_policy || (policy_a && pl.policy == 'A'))) pl.policy = 0;
  if (e->pf_next && pl.wpw != WAVES_PER_WG) pl.wpw = WAVES_PER_WG;  // (the record-prefetching launch is written for 8-wave workgroups)
  if (pl.wpw & (pl.wpw - 1)) pl.grouped = 0;  // (a self-ordering launch rebuilds its slot from log2 of the workgroup's waves: powers of two only)
  return pl;
}

// (env kinds without a `selected` plane: the zero-fill elision is vacuous, see launch_step — the flag set a launch really runs with)
static uint32_t effective_flags(const arcle_env* e, uint32_t flags) {
  if (!e->bufs.plane[ARCLE_PL_SELECTED] && (flags & ARCLE_STEP_AUTORESET) && !(flags & ~(uint32_t)(ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED | ARCLE_STEP_PACK_OBS)))
    flags |= ARCLE_STEP_ELIDE_SELECTED;
  return flags;
}

static int launch_step(arcle_env* e, int ingress, const void* sel, const int32_t* op, int32_t* reward, uint8_t* term,
                       uint32_t flags, void* stream) {
  if (!e || !sel || (!op && ingress != arcle::INGRESS_BBOX5) || !reward || !term) return ARCLE_ERR_ARG;
  if (e->base.n_ops <= 0) return fail(e, ARCLE_ERR_CONFIG, "no op table installed (arcle_set_op_table)");
  if ((flags & ARCLE_STEP_TRUNCATE) && !e->base.trunc) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_TRUNCATE without arcle_set_truncation");
  if ((flags & ARCLE_STEP_RESAMPLE) && e->base.n_problems <= 0) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_RESAMPLE without arcle_set_sampler");
  if ((flags & ARCLE_STEP_DENSE) && (!e->base.dense || !e->bufs.plane[ARCLE_PL_ANSWER])) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_DENSE without arcle_set_dense_output");
  if ((flags & ARCLE_STEP_CONTINUE_RULE) && (!arcle::is_cells(ingress) || !e->bufs.plane[ARCLE_PL_SELECTED]))
    return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_CONTINUE_RULE needs mask ingress and the `selected` plane");
  if (flags & ~0x3ffu) return fail(e, ARCLE_ERR_ARG, "unknown step flag");
  if ((flags & ARCLE_STEP_ROWS_INCREMENTAL) && !(flags & ARCLE_STEP_FLAT_OBS)) return fail(e, ARCLE_ERR_ARG, "ARCLE_STEP_ROWS_INCREMENTAL without ARCLE_STEP_FLAT_OBS");
  DeviceGuard guard(e->device);
  if (e->big) {
    // one workgroup per env (arcle_big.hip).  Flags: AUTORESET, ELIDE_SELECTED, TRUNCATE, RESAMPLE (without augmentation), CONTINUE_RULE,
    // RESET_ON_SUBMIT, FLAT_OBS (+ tail / completion signal), PACK_OBS; ROWS_INCREMENTAL rewrites the rows in full (identical bytes)
    arcle_big::BigParams q = big_params(e);
    q.ingress = ingress;
    q.sel = sel;
    q.op = op;
    q.reward = reward;
    q.term = term;
    q.flags = flags;
    q.dense = e->base.dense;
    q.acct = e->d_acct;
    if (e->d_acct) e->acct_steps += (uint64_t)q.n_envs;
    if (flags & ARCLE_STEP_FLAT_OBS) {
      if (!e->flat_out) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_FLAT_OBS without arcle_set_flat_output");
      q.flat_out = e->flat_out;
      q.flat_stride = e->flat_stride;
      q.flat_filter = e->flat_filtered ? 1 : 0;
      q.flat_tail = e->flat_tail ? 1 : 0;
      q.flat_seq = e->flat_tail ? e->flat_seq : 0;
    }
    if (flags & ARCLE_STEP_PACK_OBS) {
      if (!e->pack_out) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_PACK_OBS without arcle_set_packed_output");
      q.pack_out = reinterpret_cast<uint8_t*>(e->pack_out);
    }
    return big_done(e, arcle_big::launch_step(q, stream), "arcle_step");
  }
  // A step without ARCLE_STEP_DENSE on a handle that keeps dense pairs may move grids the cache still describes: drop the entries first
  // (stream-ordered; handles that always step with the flag never get here).  Only needed while the cache may hold pairs: a host flag,
  // set by dense steps — and stuck at "always" once a dense step was captured into a hipGraph, whose replays fill the cache unseen.
  if (e->d_dense_cache) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (flags & ARCLE_STEP_DENSE) {
      e->dense_cache_live = capturing ? 2 : (e->dense_cache_live == 2 ? 2 : 1);
    } else if (capturing || e->dense_cache_live) {
      HIP_TRY(e, hipMemsetAsync(e->d_dense_cache, 0, (size_t)e->cfg.n_envs * 8, (hipStream_t)stream));
      if (!capturing && e->dense_cache_live == 1) e->dense_cache_live = 0;
    }
  }
  // env kinds without a `selected` plane (ARCEnv, RawARCEnv: no table of theirs can hold a reset_sel-wrapped op — arcle_set_op_table
  // rejects it): the zero-fill elision is vacuous there, so an auto-resetting step of such a handle takes the same lean instantiations
  // as the O2ARC batch (ARCVecEnv's flag set) instead of the runtime-flag kernel
  flags = effective_flags(e, flags);
  StepParams p = e->base;
  p.ingress = ingress;
  p.sel = sel;
  p.op = op;
  p.reward = reward;
  p.term = term;
  p.flags = flags;
  p.acct = e->d_acct;
#ifdef ARCLE_TRACE_WAVES
  if (!e->d_acct && e->d_trace) {
    p.acct = reinterpret_cast<uint32_t*>(e->d_trace);
    p.n_steps = e->trace_seq++;
  }
#endif
  p.rmask = nullptr;
  p.next_sel = e->pf_next;
  p.stage_out = e->pf_stage;
  if (flags & ARCLE_STEP_FLAT_OBS) {
    if (!e->flat_out) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_FLAT_OBS without arcle_set_flat_output");
    p.flat_out = e->flat_out;
    p.flat_stride = e->flat_stride;
    p.flat_filter = e->flat_filtered ? 1 : 0;
    p.flat_tail = e->flat_tail ? 1 : 0;
    p.flat_seq = e->flat_tail ? e->flat_seq : 0;
  }
  if (flags & ARCLE_STEP_PACK_OBS) {
    if (!e->pack_out) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_PACK_OBS without arcle_set_packed_output");
    p.pack_out = e->pack_out;
  }
  // (a self-ordering launch reads the actions of 32 envs per wave: only from device memory, see on_device)
  const bool dev_payload = grouped_applies(e, ingress, p) && on_device(e, sel) && (ingress == arcle::INGRESS_BBOX5 || on_device(e, op));
  const LaunchPlan pl = plan_launch(e, ingress, p, dev_payload);
  p.spec_grid = pl.policy;
  p.wpw = pl.wpw;
  p.group_magic = pl.grouped ? (uint32_t)(0x100000000ull / (uint64_t)(p.n_envs / (8 * ARCLE_GROUP_SIZE))) + 1u : 0u;  // (G = groups per XCD >= 2)
  const dim3 g = grid_for(p.n_envs, pl.wpw), b(64 * pl.wpw);
  hipStream_t st = (hipStream_t)stream;
  const int fw = width_class(p);
  const bool acct = e->d_acct != nullptr;
  const bool feat = (flags & ARCLE_STEP_FEATURE_FLAGS) != 0;
  int rc;
  switch (ingress) {
    case arcle::INGRESS_BBOX: rc = launch_step_ing<arcle::INGRESS_BBOX>(fw, acct, feat, g, b, st, p); break;
    case arcle::INGRESS_POINT: rc = launch_step_ing<arcle::INGRESS_POINT>(fw, acct, feat, g, b, st, p); break;
    case arcle::INGRESS_MASK: rc = launch_step_ing<arcle::INGRESS_MASK>(fw, acct, feat, g, b, st, p); break;
    case arcle::INGRESS_BBOX5: rc = launch_step_ing<arcle::INGRESS_BBOX5>(fw, acct, feat, g, b, st, p); break;
    case arcle::INGRESS_BITS: rc = launch_step_ing<arcle::INGRESS_BITS>(fw, acct, feat, g, b, st, p); break;
    default: return fail(e, ARCLE_ERR_ARG, "unknown ingress form");
  }
  if (rc != ARCLE_OK) return fail(e, rc, "this build of libarcle_hip has no kernel for the configuration");
  HIP_TRY(e, hipGetLastError());
  if (e->d_acct) e->acct_steps += (uint64_t)p.n_envs;
  return ARCLE_OK;
}

static size_t payload_bytes(const arcle_env* e, int ingress);

extern "C" int arcle_launch_info(arcle_env* e, int ingress, uint32_t flags, int32_t* out4) {
  if (e && out4 && e->big) {  // one workgroup per env: no plan to choose; [2] = its wavefronts for this flag set / ingress form
    arcle_big::BigParams q = big_params(e);
    q.flags = flags;
    q.ingress = ingress;
    q.acct = e->d_acct;
    out4[0] = 0;
    out4[1] = 0;
    out4[2] = arcle_big::step_threads(q) / 64;
    out4[3] = 0;
    return ARCLE_OK;
  }
  if (!e || !out4) return ARCLE_ERR_ARG;
  if (ingress < 0 || ingress > arcle::INGRESS_BITS) return fail(e, ARCLE_ERR_ARG, "unknown ingress form");
  flags = effective_flags(e, flags);
  StepParams p = e->base;
  p.flags = flags;
  p.flat_stride = e->flat_stride;
  p.flat_filter = e->flat_filtered ? 1 : 0;
  p.flat_tail = e->flat_tail ? 1 : 0;
  const LaunchPlan pl = plan_launch(e, ingress, p, true);
  out4[0] = pl.grouped;
  out4[1] = pl.policy;
  out4[2] = pl.wpw;
  out4[3] = (e->tuned_valid && e->tuned_ingress == ingress && e->tuned_flags == flags) ? 1 : 0;
  return ARCLE_OK;
}

// Times the candidate launch plans of THIS handle — its batch size, this box, the caller's own action arrays where they live — and keeps the
// fastest for later launches with the same ingress form and flags.  The env state is saved first and restored before every candidate and at
// the end, so the call leaves the handle exactly as it found it (only stream order: no host synchronisation is left pending).
extern "C" int arcle_autotune(arcle_env* e, int ingress, int32_t n_batches, const void* sel, const int32_t* op, uint32_t flags, int32_t* report,
                              int32_t report_rows, void* stream) {
  if (!e || !sel || (!op && ingress != arcle::INGRESS_BBOX5)) return ARCLE_ERR_ARG;
  if (ingress < 0 || ingress > arcle::INGRESS_BITS) return fail(e, ARCLE_ERR_ARG, "unknown ingress form");
  if (n_batches <= 0) return fail(e, ARCLE_ERR_ARG, "arcle_autotune: n_batches must be positive");
  if (e->big) return 0;  // (no candidates: the workgroup-per-env launch has one plan)
  if (flags & ~(uint32_t)(ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED | ARCLE_STEP_PACK_OBS))
    return fail(e, ARCLE_ERR_CONFIG, "arcle_autotune: ARCLE_STEP_AUTORESET | _ELIDE_SELECTED | _PACK_OBS only (other flags keep per-env side state it does not save)");
  if (e->d_acct) return fail(e, ARCLE_ERR_CONFIG, "arcle_autotune: not with byte accounting enabled");
  hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return fail(e, ARCLE_ERR_CONFIG, "arcle_autotune: not inside a stream capture");
  }
  DeviceGuard guard(e->device);
  const size_t n = (size_t)e->cfg.n_envs, pbytes = n * (size_t)e->base.PS, pb = payload_bytes(e, ingress);
  // (the timed launches walk the caller's action batches in order, like arcle_step_many: ONE repeated batch is not a workload — the state
  // degenerates under it and the actions never leave the caches; profiles/round5_policy_autotune.txt)
  // (... and the warm-up must reach the steady state of the caches: a plan timed 8 launches after the state copy ranks the non-temporal
  // policies wrongly — 65 536 envs: B 32.2 us measured, 26.3 on a replayed graph; from 48 warm-up launches on the two agree, tools/autotunebench.py)
  int n_warm = 2 * n_batches < 48 ? 48 : (2 * n_batches < 96 ? 2 * n_batches : 96), n_timed = n_batches < 24 ? 24 : (n_batches < 64 ? n_batches : 64);
  if (const char* w = getenv("ARCLE_AUTOTUNE_WARM")) n_warm = atoi(w) > 0 ? atoi(w) : n_warm;
  if (const char* w = getenv("ARCLE_AUTOTUNE_TIMED")) n_timed = atoi(w) > 0 ? atoi(w) : n_timed;
  // scratch: a copy of every plane, the records, the counters; outputs of the timed launches
  int8_t* save_plane[ARCLE_N_PLANES] = {nullptr};
  int8_t* save_rec = nullptr;
  int32_t *save_cnt = nullptr, *t_reward = nullptr;
  uint8_t* t_term = nullptr;
  // (the caller's packed-row buffer and the sticky status word are part of "the handle as it was found": the timed launches write their
  // rows to a scratch buffer, and the status word is saved and put back)
  int8_t *t_pack = nullptr, *const caller_pack = e->pack_out;
  uint32_t* save_status = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ok = hipMalloc((void**)&save_rec, n * ARCLE_REC_BYTES) == hipSuccess && hipMalloc((void**)&save_cnt, n * 8) == hipSuccess &&
            hipMalloc((void**)&t_reward, n * 4) == hipSuccess && hipMalloc((void**)&t_term, n) == hipSuccess &&
            hipMalloc((void**)&save_status, 8) == hipSuccess && hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess;
  if (ok && (flags & ARCLE_STEP_PACK_OBS) && caller_pack) {
    ok = hipMalloc((void**)&t_pack, n * (size_t)arcle_packed_obs_size(e)) == hipSuccess;
    if (ok) e->pack_out = t_pack;
  }
  ok = ok && hipMemcpyAsync(save_status, e->d_status, 8, hipMemcpyDeviceToDevice, st) == hipSuccess;
  for (int i = 0; ok && i < ARCLE_N_PLANES; i++)
    if (e->bufs.plane[i]) ok = hipMalloc((void**)&save_plane[i], pbytes) == hipSuccess;
  auto copy_state = [&](bool save) -> bool {
    bool r = true;
    for (int i = 0; r && i < ARCLE_N_PLANES; i++)
      if (e->bufs.plane[i])
        r = hipMemcpyAsync(save ? save_plane[i] : e->bufs.plane[i], save ? e->bufs.plane[i] : save_plane[i], pbytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    r = r && hipMemcpyAsync(save ? save_rec : e->bufs.rec, save ? e->bufs.rec : save_rec, n * ARCLE_REC_BYTES, hipMemcpyDeviceToDevice, st) == hipSuccess;
    r = r && hipMemcpyAsync(save ? (void*)save_cnt : (void*)e->bufs.cnt, save ? (void*)e->bufs.cnt : (void*)save_cnt, n * 8, hipMemcpyDeviceToDevice, st) == hipSuccess;
    return r;
  };
  int rc = ARCLE_OK;
  LaunchPlan best = {0, WAVES_PER_WG, 0};
  float best_ms = -1.f;
  int rows = 0;
  if (ok) ok = copy_state(true);
  if (ok) {
    e->tuned_valid = 0;
    const int policies[] = {0, 'A', 'B', 'H', 'J'};
    for (int grouped = 0; grouped <= 1 && rc == ARCLE_OK; grouped++)
      for (int pi = 0; pi < (grouped ? 1 : 5) && rc == ARCLE_OK; pi++)
        for (int wpw = 4; wpw <= 8 && rc == ARCLE_OK; wpw += 4) {
          LaunchPlan cand = {policies[pi], wpw, grouped};
          // does the candidate survive planning unchanged?  (a policy / the grouping the launch cannot take is not a candidate)
          StepParams probe = e->base;
          probe.flags = effective_flags(e, flags);
          e->forced = &cand;
          const bool dev_payload = on_device(e, sel) && (ingress == arcle::INGRESS_BBOX5 || on_device(e, op));
          const LaunchPlan got = plan_launch(e, ingress, probe, dev_payload);
          if (got.policy == cand.policy && got.wpw == cand.wpw && got.grouped == cand.grouped) {
            float ms = 0.f;
            bool r = copy_state(false);
            for (int it = 0; r && it < n_warm + n_timed; it++) {
              if (it == n_warm) r = hipEventRecord(ev0, st) == hipSuccess;
              const size_t bi = (size_t)(it % n_batches);
              if (r) rc = launch_step(e, ingress, (const char*)sel + bi * pb, op ? op + bi * n : nullptr, t_reward, (uint8_t*)t_term, flags, stream);
              r = r && rc == ARCLE_OK;
            }
            r = r && hipEventRecord(ev1, st) == hipSuccess && hipEventSynchronize(ev1) == hipSuccess && hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess;
            if (!r && rc == ARCLE_OK) rc = ARCLE_ERR_HIP;
            if (r) {
              if (best_ms < 0.f || ms < best_ms) best_ms = ms, best = cand;
              if (report && rows < report_rows) {
                report[4 * rows + 0] = cand.grouped, report[4 * rows + 1] = cand.policy, report[4 * rows + 2] = cand.wpw;
                report[4 * rows + 3] = (int32_t)(ms * 1e6f / (float)n_timed);  // ns per launch
                rows++;
              }
            }
          }
          e->forced = nullptr;
        }
    if (!copy_state(false) || hipMemcpyAsync(e->d_status, save_status, 8, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      rc = rc == ARCLE_OK ? ARCLE_ERR_HIP : rc;
  } else {
    rc = ARCLE_ERR_HIP;
  }
  e->pack_out = caller_pack;
  if (t_pack) (void)hipFree(t_pack);
  if (save_status) (void)hipFree(save_status);
  for (int i = 0; i < ARCLE_N_PLANES; i++)
    if (save_plane[i]) (void)hipFree(save_plane[i]);
  if (save_rec) (void)hipFree(save_rec);
  if (save_cnt) (void)hipFree(save_cnt);
  if (t_reward) (void)hipFree(t_reward);
  if (t_term) (void)hipFree(t_term);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  if (rc != ARCLE_OK) {
    (void)hipGetLastError();
    if (rc == ARCLE_ERR_HIP) snprintf(e->err, sizeof(e->err), "arcle_autotune: a HIP call failed (out of memory for the state copy?)");
    return rc;
  }
  if (best_ms >= 0.f) {
    e->tuned = best;
    e->tuned_ingress = ingress;
    e->tuned_flags = effective_flags(e, flags);
    e->tuned_valid = 1;
  }
  return rows;
}

extern "C" int arcle_step_mask(arcle_env* e, const int8_t* sel, const int32_t* op, int32_t* reward, uint8_t* term,
                               uint32_t flags, void* stream) {
  return launch_step(e, arcle::INGRESS_MASK, sel, op, reward, term, flags, stream);
}
extern "C" int arcle_step_bbox(arcle_env* e, const int32_t* bbox, const int32_t* op, int32_t* reward, uint8_t* term,
                               uint32_t flags, void* stream) {
  return launch_step(e, arcle::INGRESS_BBOX, bbox, op, reward, term, flags, stream);
}
extern "C" int arcle_step_point(arcle_env* e, const int32_t* xy, const int32_t* op, int32_t* reward, uint8_t* term,
                                uint32_t flags, void* stream) {
  return launch_step(e, arcle::INGRESS_POINT, xy, op, reward, term, flags, stream);
}
extern "C" int arcle_step_bbox5(arcle_env* e, const int32_t* act5, int32_t* reward, uint8_t* term, uint32_t flags, void* stream) {
  return launch_step(e, arcle::INGRESS_BBOX5, act5, nullptr, reward, term, flags, stream);
}
extern "C" int arcle_step_bits(arcle_env* e, const uint8_t* bits, const int32_t* op, int32_t* reward, uint8_t* term,
                               uint32_t flags, void* stream) {
  return launch_step(e, arcle::INGRESS_BITS, bits, op, reward, term, flags, stream);
}

// bytes of one step's selection payload / op array for the whole batch (arcle_step_many strides)
static size_t payload_bytes(const arcle_env* e, int ingress) {
  const size_t n = (size_t)e->cfg.n_envs;
  switch (ingress) {
    case arcle::INGRESS_MASK: return n * (size_t)e->base.P;
    case arcle::INGRESS_BBOX: return n * 16;
    case arcle::INGRESS_POINT: return n * 8;
    case arcle::INGRESS_BBOX5: return n * 20;
    default: return n * (size_t)(e->big ? e->base.PS >> 3 : ARCLE_BITS_STRIDE);
  }
}

extern "C" int arcle_step_many(arcle_env* e, int ingress, int32_t n_steps, const void* sel, const int32_t* op, int32_t* reward,
                               uint8_t* term, uint32_t flags, void* stream) {
  if (!e) return ARCLE_ERR_ARG;
  if (n_steps <= 0) return fail(e, ARCLE_ERR_ARG, "n_steps must be positive");
  if (ingress < 0 || ingress > arcle::INGRESS_BITS) return fail(e, ARCLE_ERR_ARG, "unknown ingress form");
  if (n_steps == 1) return launch_step(e, ingress, sel, op, reward, term, flags, stream);
  const size_t n = (size_t)e->cfg.n_envs, pb = payload_bytes(e, ingress);
  // Host-resident 5-tuple records (a policy on the CPU): step t reads its records from a device staging buffer that the FRONT
  // workgroups of launch t-1 filled from pinned host memory while that launch ran; only step 0 reads across PCIe itself.
  bool prefetch = false;
  // (only where the lean instantiation that carries the copy workgroups applies: the standard 30 x 30 batch with ARCVecEnv's flags)
  const bool pf_kernel = !e->big && width_class(e->base) == arcle::FW_FULL && e->base.H == 30 && e->base.W == 30 && flags == (uint32_t)HOT_FLAGS &&
                         !e->d_acct && launch_wpw(e) == WAVES_PER_WG;
  if (pf_kernel && ingress == arcle::INGRESS_BBOX5 && n_steps > 1 && (n & 3) == 0 && sel) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, sel) == hipSuccess && attr.type == hipMemoryTypeHost) prefetch = true;
    else (void)hipGetLastError();
  }
  if (prefetch && !e->d_stage) {
    // (the staging buffer is allocated by the first such call OUTSIDE a stream capture: allocating would invalidate a capture in
    // progress — a captured call without it falls back to every wave reading its own record across PCIe)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      prefetch = false;
    } else {
      DeviceGuard guard(e->device);
      if (hipMalloc((void**)&e->d_stage, 2 * n * 20) != hipSuccess) {
        (void)hipGetLastError();
        prefetch = false;
      }
    }
  }
  // (device-resident payloads need nothing here: the launches of the standard batch order themselves, see launch_step)
  int rc = ARCLE_OK;
  for (int32_t t = 0; t < n_steps && rc == ARCLE_OK; t++) {
    const void* src = (const char*)sel + (size_t)t * pb;
    if (prefetch) {
      if (t > 0) src = e->d_stage + (size_t)(t & 1) * n * 5;
      e->pf_next = t + 1 < n_steps ? (const int32_t*)((const char*)sel + (size_t)(t + 1) * pb) : nullptr;
      e->pf_stage = e->d_stage + (size_t)((t + 1) & 1) * n * 5;
      e->pf_active = 1;
    }
    rc = launch_step(e, ingress, src, op ? op + (size_t)t * n : nullptr, reward ? reward + (size_t)t * n : nullptr,
                     term ? term + (size_t)t * n : nullptr, flags, stream);
  }
  e->pf_next = nullptr;
  e->pf_stage = nullptr;
  e->pf_active = 0;
  return rc;
}

// Since ABI 5 a launch orders ITSELF from the operations it is about to execute (see arcle_step_kernel): nobody has to know the next step's
// operations, so the round-4 hint has nothing left to do.  Kept, validated and ignored, for callers written against ABI 4.
extern "C" int arcle_hint_next_ops(arcle_env* e, const int32_t* next_op, int32_t stride) {
  if (!e) return ARCLE_ERR_ARG;
  if (next_op && stride <= 0) return fail(e, ARCLE_ERR_ARG, "arcle_hint_next_ops: stride must be positive (1 for op arrays, 5 for BBoxWrapper records)");
  return ARCLE_OK;
}

extern "C" int arcle_set_dispatch_order(arcle_env* e, int enable) {
  if (!e) return ARCLE_ERR_ARG;
  e->order_enabled = enable ? 1 : 0;
  return ARCLE_OK;
}

extern "C" int arcle_mask_bits_stride(const arcle_env* e) {
  if (!e) return ARCLE_ERR_ARG;
  return e->big ? e->base.PS >> 3 : ARCLE_BITS_STRIDE;
}

extern "C" int arcle_pack_mask_bits(arcle_env* e, const int8_t* sel, uint8_t* bits, void* stream) {
  if (!e || !sel || !bits) return ARCLE_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(bits) & 1) return fail(e, ARCLE_ERR_ARG, "bit-packed mask rows must be 2-byte aligned");
  if (e->big) {  // (rows of plane_stride / 8 bytes)
    DeviceGuard guard(e->device);
    arcle_big::BigParams q = big_params(e);
    q.sel = sel;
    q.pack_out = bits;
    return big_done(e, arcle_big::launch_rows(q, 2, stream), "arcle_pack_mask_bits");
  }
  DeviceGuard guard(e->device);
  StepParams p = e->base;
  p.sel = sel;
  hipLaunchKernelGGL(arcle_pack_bits_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, (hipStream_t)stream, p, bits);
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

#ifdef ARCLE_FAST_BUILD
template <int ING>
static int launch_rollout_ing(int fw, dim3 g, dim3 b, hipStream_t st, const StepParams& p) {  // (development builds: the 30 x 30 bbox rollouts only)
  if (ING != arcle::INGRESS_BBOX || fw == arcle::FW_GENERIC || p.H != 30 || p.W != 30) return ARCLE_ERR_CONFIG;
  if (p.flags == (uint32_t)HOT_FLAGS) hipLaunchKernelGGL((arcle_rollout_kernel<arcle::INGRESS_BBOX, arcle::FW_FAST, 30, HOT_FLAGS>), g, b, 0, st, p);
  else if (p.flags == (uint32_t)HOT_PACK_FLAGS) hipLaunchKernelGGL((arcle_rollout_kernel<arcle::INGRESS_BBOX, arcle::FW_FAST, 30, HOT_PACK_FLAGS>), g, b, 0, st, p);
  else hipLaunchKernelGGL((arcle_rollout_kernel<arcle::INGRESS_BBOX, arcle::FW_FAST, 30>), g, b, 0, st, p);
  return ARCLE_OK;
}
#else
template <int ING, int FW>
static void launch_rollout_tbl(dim3 g, dim3 b, hipStream_t st, const StepParams& p) {
  hipLaunchKernelGGL((arcle_rollout_kernel<ING, FW>), g, b, 0, st, p);
}
template <int ING>
static int launch_rollout_ing(int fw, dim3 g, dim3 b, hipStream_t st, const StepParams& p) {
  // (the rollout keeps planes in registers: lane predication does not matter, FW_FULL shares FW_FAST's code)
  if constexpr (ING == arcle::INGRESS_BBOX || ING == arcle::INGRESS_POINT) {  // the front-ends' flag sets: compile-time constants of lean instantiations
    if (fw != arcle::FW_GENERIC && p.H == 30 && p.W == 30 && p.flags == (uint32_t)HOT_FLAGS) {
      hipLaunchKernelGGL((arcle_rollout_kernel<ING, arcle::FW_FAST, 30, HOT_FLAGS>), g, b, 0, st, p);
      return ARCLE_OK;
    }
    if (fw != arcle::FW_GENERIC && p.H == 30 && p.W == 30 && p.flags == (uint32_t)HOT_PACK_FLAGS) {
      hipLaunchKernelGGL((arcle_rollout_kernel<ING, arcle::FW_FAST, 30, HOT_PACK_FLAGS>), g, b, 0, st, p);
      return ARCLE_OK;
    }
  }
  if (fw != arcle::FW_GENERIC && p.H == 30 && p.W == 30) hipLaunchKernelGGL((arcle_rollout_kernel<ING, arcle::FW_FAST, 30>), g, b, 0, st, p);
  else if (fw != arcle::FW_GENERIC) launch_rollout_tbl<ING, arcle::FW_FAST>(g, b, st, p);
  else launch_rollout_tbl<ING, arcle::FW_GENERIC>(g, b, st, p);
  return ARCLE_OK;
}
#endif

static int launch_rollout(arcle_env* e, int ingress, int32_t n_steps, const void* sel, const int32_t* op, int32_t* reward,
                          uint8_t* term, uint32_t flags, void* stream) {
  if (!e || !sel || !op || !reward || !term) return ARCLE_ERR_ARG;
  if (n_steps <= 0) return fail(e, ARCLE_ERR_ARG, "n_steps must be positive");
  if (e->base.n_ops <= 0) return fail(e, ARCLE_ERR_CONFIG, "no op table installed (arcle_set_op_table)");
  if (flags & ~(ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED | ARCLE_STEP_CONTINUE_RULE | ARCLE_STEP_RESET_ON_SUBMIT | ARCLE_STEP_PACK_OBS))
    return fail(e, ARCLE_ERR_ARG, "flag not supported by the rollout kernels");
  if ((flags & (ARCLE_STEP_CONTINUE_RULE | ARCLE_STEP_RESET_ON_SUBMIT)) && ingress != arcle::INGRESS_MASK)
    return fail(e, ARCLE_ERR_CONFIG, "the rollout kernels take ARCLE_STEP_CONTINUE_RULE / _RESET_ON_SUBMIT with mask ingress only");
  if ((flags & ARCLE_STEP_PACK_OBS) && !e->pack_out)
    return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_PACK_OBS without arcle_set_packed_output (rollouts: uint8 [n_steps][n_envs][arcle_packed_obs_size()])");
  if (e->big) {
    // "semantically identical to n_steps calls of arcle_step_*": for the workgroup-per-env kernels that is what a rollout is (the state does
    // not fit a wavefront's registers); with ARCLE_STEP_PACK_OBS every step's rows go to its own slice of the installed buffer
    const size_t n = (size_t)e->cfg.n_envs, pb = payload_bytes(e, ingress);
    int8_t* const pack0 = e->pack_out;
    int rc = ARCLE_OK;
    for (int32_t t = 0; t < n_steps && rc == ARCLE_OK; t++) {
      if (pack0) e->pack_out = pack0 + (size_t)t * n * (size_t)arcle_big::packed_stride(e->base.P);
      rc = launch_step(e, ingress, (const char*)sel + (size_t)t * pb, op + (size_t)t * n, reward + (size_t)t * n, term + (size_t)t * n, flags, stream);
    }
    e->pack_out = pack0;
    return rc;
  }
  DeviceGuard guard(e->device);
  if (e->d_dense_cache) HIP_TRY(e, hipMemsetAsync(e->d_dense_cache, 0, (size_t)e->cfg.n_envs * 8, (hipStream_t)stream));  // (rollouts move grids, keep no pairs)
  StepParams p = e->base;
  p.ingress = ingress;
  p.sel = sel;
  p.op = op;
  p.reward = reward;
  p.term = term;
  p.flags = flags;
  p.acct = nullptr;
  p.rmask = nullptr;
  p.n_steps = n_steps;
  p.pack_out = e->pack_out;
  const dim3 g = grid_for(p.n_envs), b(64 * WAVES_PER_WG);
  hipStream_t st = (hipStream_t)stream;
  const int fw = width_class(p);
  int rc;
  if (ingress == arcle::INGRESS_BBOX) rc = launch_rollout_ing<arcle::INGRESS_BBOX>(fw, g, b, st, p);
  else if (ingress == arcle::INGRESS_POINT) rc = launch_rollout_ing<arcle::INGRESS_POINT>(fw, g, b, st, p);
  else rc = launch_rollout_ing<arcle::INGRESS_MASK>(fw, g, b, st, p);
  if (rc != ARCLE_OK) return fail(e, rc, "this build of libarcle_hip has no rollout kernel for the configuration");
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

extern "C" int arcle_rollout_bbox(arcle_env* e, int32_t n_steps, const int32_t* bbox, const int32_t* op, int32_t* reward,
                                  uint8_t* term, uint32_t flags, void* stream) {
  return launch_rollout(e, arcle::INGRESS_BBOX, n_steps, bbox, op, reward, term, flags, stream);
}
extern "C" int arcle_rollout_point(arcle_env* e, int32_t n_steps, const int32_t* xy, const int32_t* op, int32_t* reward,
                                   uint8_t* term, uint32_t flags, void* stream) {
  return launch_rollout(e, arcle::INGRESS_POINT, n_steps, xy, op, reward, term, flags, stream);
}
extern "C" int arcle_rollout_mask(arcle_env* e, int32_t n_steps, const int8_t* sel, const int32_t* op, int32_t* reward,
                                  uint8_t* term, uint32_t flags, void* stream) {
  return launch_rollout(e, arcle::INGRESS_MASK, n_steps, sel, op, reward, term, flags, stream);
}

extern "C" int arcle_set_sampler(arcle_env* e, const int32_t* pair_off, const int32_t* pair_cnt, int32_t n_problems, uint64_t seed,
                                 int64_t env_base, int32_t* episode, int32_t* cur_task, uint32_t aug_flags) {
  if (!e || !pair_off || !pair_cnt || !episode) return ARCLE_ERR_ARG;
  if (n_problems <= 0) return fail(e, ARCLE_ERR_CONFIG, "the sampler needs at least one problem with a pair");
  if (e->base.n_tasks <= 0) return fail(e, ARCLE_ERR_CONFIG, "no task table installed (arcle_set_task_table)");
  if (aug_flags & ~(ARCLE_AUG_PERMUTE | ARCLE_AUG_ROT90)) return fail(e, ARCLE_ERR_ARG, "unknown augmentation flag");
  e->base.pair_off = pair_off;
  e->base.pair_cnt = pair_cnt;
  e->base.n_problems = n_problems;
  e->base.seed = seed;
  e->base.env_base = env_base;
  e->base.episode = episode;
  e->base.cur_task = cur_task;
  e->base.aug_flags = aug_flags;
  return ARCLE_OK;
}

extern "C" int arcle_reset_sampled(arcle_env* e, const uint8_t* mask, void* stream) {
  if (!e) return ARCLE_ERR_ARG;
  if (e->base.n_problems <= 0) return fail(e, ARCLE_ERR_CONFIG, "no sampler installed (arcle_set_sampler)");
  DeviceGuard guard(e->device);
  if (e->big) {
    arcle_big::BigParams q = big_params(e);
    q.rmask = mask;
    return big_done(e, arcle_big::launch_reset(q, 2, stream), "arcle_reset_sampled");
  }
  StepParams p = e->base;
  p.rmask = mask;
  p.task_idx = nullptr;
  hipLaunchKernelGGL(arcle_reset_table_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, (hipStream_t)stream, p);
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

extern "C" int arcle_reset_from_table_aug(arcle_env* e, const int32_t* task_idx, const uint8_t* mask, const uint8_t* aug_k,
                                          const uint8_t* aug_perm, void* stream) {
  if (!e || !task_idx) return ARCLE_ERR_ARG;
  if (e->base.n_tasks <= 0) return fail(e, ARCLE_ERR_CONFIG, "no task table installed (arcle_set_task_table)");
  if (e->big) {
    DeviceGuard guard(e->device);
    arcle_big::BigParams q = big_params(e);
    q.rmask = mask;
    q.task_idx = task_idx;
    q.aug_k = aug_k;
    q.aug_perm = aug_perm;
    return big_done(e, arcle_big::launch_reset(q, 1, stream), "arcle_reset_from_table_aug");
  }
  DeviceGuard guard(e->device);
  StepParams p = e->base;
  p.rmask = mask;
  p.task_idx = task_idx;
  p.aug_k = aug_k;
  p.aug_perm = aug_perm;
  hipLaunchKernelGGL(arcle_reset_table_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, (hipStream_t)stream, p);
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

extern "C" int arcle_set_dense_output(arcle_env* e, int32_t* dense_out) {
  if (!e) return ARCLE_ERR_ARG;
  if (e->big) {  // (the workgroup-per-env kernels compute the pair from the planes every step: no cache)
    e->base.dense = dense_out;
    return ARCLE_OK;
  }
  if (dense_out && !e->d_dense_cache) {  // the per-env cache of the current grid's pair; (0, 0) = unknown
    DeviceGuard guard(e->device);
    HIP_TRY(e, hipMalloc((void**)&e->d_dense_cache, (size_t)e->cfg.n_envs * 8));
    HIP_TRY(e, hipMemset(e->d_dense_cache, 0, (size_t)e->cfg.n_envs * 8));
    e->base.dense_cache = e->d_dense_cache;
  }
  e->base.dense = dense_out;
  return ARCLE_OK;
}

extern "C" int arcle_invalidate(arcle_env* e, void* stream) {
  if (!e) return ARCLE_ERR_ARG;
  if (e->d_dense_cache) {
    DeviceGuard guard(e->device);
    HIP_TRY(e, hipMemsetAsync(e->d_dense_cache, 0, (size_t)e->cfg.n_envs * 8, (hipStream_t)stream));
  }
  return ARCLE_OK;
}

extern "C" int arcle_set_truncation(arcle_env* e, uint8_t* trunc_out, int32_t step_limit) {
  if (!e) return ARCLE_ERR_ARG;
  if (trunc_out && step_limit <= 0) return fail(e, ARCLE_ERR_ARG, "step_limit must be positive");
  e->base.trunc = trunc_out;
  e->base.step_limit = step_limit;
  return ARCLE_OK;
}

extern "C" int arcle_flat_obs_size(const arcle_env* e, int filtered) {
  if (!e) return ARCLE_ERR_ARG;
  if (filtered && !(e->bufs.plane[ARCLE_PL_SELECTED] && e->bufs.plane[ARCLE_PL_CLIP])) return ARCLE_ERR_CONFIG;
  return arcle::flat_obs_len(e->base, filtered);  // (the same formula for any H x W)
}

static int launch_flatten(arcle_env* e, int8_t* out, int32_t out_stride, int filtered, hipStream_t st) {
  const int len = arcle_flat_obs_size(e, filtered);
  if (len < 0) return fail(e, ARCLE_ERR_CONFIG, "the FilterO2ARC subset needs the O2ARCv2Env state planes");
  if (out_stride < len || (out_stride & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return fail(e, ARCLE_ERR_ARG, "flat observation rows: 16-byte aligned, stride a multiple of 16 >= arcle_flat_obs_size()");
  if (e->big) {
    arcle_big::BigParams q = big_params(e);
    q.flat_out = out;
    q.flat_stride = out_stride;
    q.flat_filter = filtered ? 1 : 0;
    return big_done(e, arcle_big::launch_rows(q, 0, st), "arcle_flatten_obs");
  }
  StepParams p = e->base;
  p.flat_out = out;
  p.flat_stride = out_stride;
  p.flat_filter = filtered ? 1 : 0;
  hipLaunchKernelGGL(arcle_flatten_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, st, p);
  HIP_TRY(e, hipGetLastError());
  // a row reads its planes + the record once and is written once (SURVEY.md 8d accounting of the observation writer)
  if (e->d_acct) e->acct_extra += (uint64_t)p.n_envs * (uint64_t)(2 * len + ARCLE_REC_BYTES);
  return ARCLE_OK;
}

extern "C" int arcle_flatten_obs(arcle_env* e, int8_t* out, int32_t out_stride, int filtered, void* stream) {
  if (!e || !out) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  return launch_flatten(e, out, out_stride, filtered, (hipStream_t)stream);
}

extern "C" int arcle_set_flat_output(arcle_env* e, int8_t* out, int32_t out_stride, int filtered) {
  if (!e) return ARCLE_ERR_ARG;
  if (out) {
    const int len = arcle_flat_obs_size(e, filtered);
    if (len < 0) return fail(e, ARCLE_ERR_CONFIG, "the FilterO2ARC subset needs the O2ARCv2Env state planes");
    if (out_stride < len || (out_stride & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
      return fail(e, ARCLE_ERR_ARG, "flat observation rows: 16-byte aligned, stride a multiple of 16 >= arcle_flat_obs_size()");
  }
  e->flat_out = out;
  e->flat_stride = out_stride;
  e->flat_filtered = filtered ? 1 : 0;
  e->flat_tail = 0;
  return ARCLE_OK;
}

extern "C" int arcle_set_flat_output_ex(arcle_env* e, int8_t* out, int32_t out_stride, int filtered, int tail) {
  if (!e) return ARCLE_ERR_ARG;
  if (out && tail) {
    const int len = arcle_flat_obs_size(e, filtered);
    if (len >= 0 && out_stride < ((len + 15) & ~15) + 16)
      return fail(e, ARCLE_ERR_ARG, "flat rows with a tail: stride >= arcle_flat_obs_size() rounded up to 16, plus 16");
  }
  const int rc = arcle_set_flat_output(e, out, out_stride, filtered);
  if (rc == ARCLE_OK) e->flat_tail = (out && tail) ? 1 : 0;
  return rc;
}

extern "C" int arcle_set_flat_seq(arcle_env* e, int32_t seq) {
  if (!e) return ARCLE_ERR_ARG;
  if (seq < 0 || seq > 255) return fail(e, ARCLE_ERR_ARG, "arcle_set_flat_seq: 0 (off) .. 255");
  e->flat_seq = seq;
  return ARCLE_OK;
}

// ---- state rows at the boundary: ingest (inverse of arcle_flatten_obs), stateless batched transition, plane copies ----------
static int check_rows(arcle_env* e, const void* rows, int32_t stride, int extra) {
  const int len = arcle::flat_obs_len(e->base, 0);
  if (!rows || stride < len + extra) return fail(e, ARCLE_ERR_ARG, "state rows: stride >= arcle_flat_obs_size(env, 0)");
  return ARCLE_OK;
}

extern "C" int arcle_set_state_rows(arcle_env* e, const int8_t* rows, int32_t stride, const uint8_t* mask, void* stream) {
  if (!e) return ARCLE_ERR_ARG;
  if (int rc = check_rows(e, rows, stride, 0)) return rc;
  DeviceGuard guard(e->device);
  if (e->big) {
    arcle_big::BigParams q = big_params(e);
    q.rows_in = rows;
    q.rows_in_stride = stride;
    q.rmask = mask;
    return big_done(e, arcle_big::launch_set_rows(q, stream), "arcle_set_state_rows");
  }
  StepParams p = e->base;
  p.rows_in = rows;
  p.rows_in_stride = stride;
  p.rmask = mask;
  hipLaunchKernelGGL(arcle_set_state_rows_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, (hipStream_t)stream, p);
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

extern "C" int arcle_get_state_rows(arcle_env* e, int8_t* rows, int32_t stride, void* stream) {
  if (!e || !rows) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  return launch_flatten(e, rows, stride, 0, (hipStream_t)stream);
}

template <int ING>
static void launch_transition_ing(int fw, dim3 g, dim3 b, hipStream_t st, const StepParams& p) {
  // (planes live in registers: lane predication does not matter, FW_FULL shares FW_FAST's code — as in the rollout kernels)
  if (fw != arcle::FW_GENERIC) hipLaunchKernelGGL((arcle_transition_rows_kernel<ING, arcle::FW_FAST>), g, b, 0, st, p);
  else hipLaunchKernelGGL((arcle_transition_rows_kernel<ING, arcle::FW_GENERIC>), g, b, 0, st, p);
}

extern "C" int arcle_transition_rows(arcle_env* e, int32_t n_rows, const int8_t* rows_in, int32_t in_stride, int ingress,
                                     const void* sel, const int32_t* op, const int32_t* src_env, int8_t* rows_out,
                                     int32_t out_stride, int tail, int32_t* reward, uint8_t* term, uint32_t flags, void* stream) {
  if (!e || !sel || !op || !reward || !term || !rows_out) return ARCLE_ERR_ARG;
  if (n_rows <= 0) return fail(e, ARCLE_ERR_ARG, "n_rows must be positive");
  if (!src_env && n_rows > e->cfg.n_envs) return fail(e, ARCLE_ERR_ARG, "more rows than envs: pass src_env (which env's answer every row uses)");
  if ((uint64_t)n_rows * ARCLE_MAX_CELLS >= (1ull << 32)) return fail(e, ARCLE_ERR_ARG, "too many rows");
  if (e->base.n_ops <= 0) return fail(e, ARCLE_ERR_CONFIG, "no op table installed (arcle_set_op_table)");
  if (flags & ~(ARCLE_STEP_RESET_ON_SUBMIT | ARCLE_STEP_DENSE | ARCLE_STEP_CONTINUE_RULE))
    return fail(e, ARCLE_ERR_ARG, "arcle_transition_rows takes ARCLE_STEP_RESET_ON_SUBMIT / _DENSE / _CONTINUE_RULE only");
  if ((flags & ARCLE_STEP_DENSE) && !e->base.dense) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_DENSE without arcle_set_dense_output");
  if ((flags & ARCLE_STEP_DENSE) && n_rows > e->cfg.n_envs) return fail(e, ARCLE_ERR_ARG, "ARCLE_STEP_DENSE: the dense output has one pair per env, n_rows <= n_envs");
  if ((flags & ARCLE_STEP_CONTINUE_RULE) && ingress != arcle::INGRESS_MASK) return fail(e, ARCLE_ERR_CONFIG, "ARCLE_STEP_CONTINUE_RULE needs mask ingress");
  if (int rc = check_rows(e, rows_in, in_stride, 0)) return rc;
  const int len = arcle::flat_obs_len(e->base, 0);
  if ((out_stride & 15) || (reinterpret_cast<uintptr_t>(rows_out) & 15) || out_stride < ((len + 15) & ~15) + (tail ? 16 : 0))
    return fail(e, ARCLE_ERR_ARG, "output rows: 16-byte aligned, stride a multiple of 16 >= the row length (+16 with a tail)");
  DeviceGuard guard(e->device);
  if (e->big) {
    // the state does not fit a wavefront, so the stateless transition is three launches over SCRATCH envs (one per row): rows -> scratch
    // planes / records (+ the answer of resident env src_env[r]), one step() of the scratch envs with the fused row writer, i.e.
    // row r of rows_out = FlattenObservation of the stepped state (+ tail).  The resident envs are not touched.  The scratch (8 planes +
    // record + counters per row) is allocated — or grown — here: not inside a stream capture.
    if (ingress != arcle::INGRESS_MASK && ingress != arcle::INGRESS_BBOX && ingress != arcle::INGRESS_POINT)  // (before any allocation or launch)
      return fail(e, ARCLE_ERR_ARG, "arcle_transition_rows takes mask, bbox or point selections");
    const size_t PS = (size_t)e->base.PS, per_row = ARCLE_N_PLANES * PS + ARCLE_REC_BYTES + 8;
    if (n_rows > e->big_scratch_rows) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return fail(e, ARCLE_ERR_CONFIG, "arcle_transition_rows of a big-grid handle allocates its scratch envs on first use: call it once outside the stream capture");
      }
      if (e->big_scratch) {
        HIP_TRY(e, hipDeviceSynchronize());
        (void)hipFree(e->big_scratch);
        e->big_scratch = nullptr;
        e->big_scratch_rows = 0;
      }
      HIP_TRY(e, hipMalloc((void**)&e->big_scratch, per_row * (size_t)n_rows));
      e->big_scratch_rows = n_rows;
    }
    arcle_big::BigParams q = big_params(e);
    const size_t R = (size_t)e->big_scratch_rows;
    for (int i = 0; i < ARCLE_N_PLANES; i++) q.plane[i] = e->bufs.plane[i] ? e->big_scratch + (size_t)i * R * PS : nullptr;
    q.rec = e->big_scratch + ARCLE_N_PLANES * R * PS;
    q.cnt = reinterpret_cast<int32_t*>(e->big_scratch + ARCLE_N_PLANES * R * PS + R * ARCLE_REC_BYTES);
    q.n_envs = n_rows;
    q.n_resident = e->cfg.n_envs;
    q.src_env = src_env;
    q.res_answer = e->bufs.plane[ARCLE_PL_ANSWER];
    q.res_rec = e->bufs.rec;
    q.rows_in = rows_in;
    q.rows_in_stride = in_stride;
    if (int rc = big_done(e, arcle_big::launch_set_rows(q, stream), "arcle_transition_rows (rows in)")) return rc;
    q.ingress = ingress;
    q.sel = sel;
    q.op = op;
    q.reward = reward;
    q.term = term;
    q.flags = flags | ARCLE_STEP_FLAT_OBS;
    q.dense = e->base.dense;
    q.flat_out = rows_out;
    q.flat_stride = out_stride;
    q.flat_filter = 0;
    q.flat_tail = tail ? 1 : 0;
    q.flat_seq = tail ? e->flat_seq : 0;
    return big_done(e, arcle_big::launch_step(q, stream), "arcle_transition_rows");
  }
  StepParams p = e->base;
  p.n_resident = p.n_envs;
  p.n_envs = n_rows;
  p.ingress = ingress;
  p.sel = sel;
  p.op = op;
  p.reward = reward;
  p.term = term;
  p.flags = flags;
  p.acct = nullptr;
  p.rmask = nullptr;
  p.task_idx = src_env;
  p.rows_in = rows_in;
  p.rows_in_stride = in_stride;
  p.flat_out = rows_out;
  p.flat_stride = out_stride;
  p.flat_filter = 0;
  p.flat_tail = tail ? 1 : 0;
  p.flat_seq = tail ? e->flat_seq : 0;
  // in place: a plane the op did not touch stays where it is (the writer's incremental mode); otherwise it is passed through
  if (rows_out == rows_in && out_stride == in_stride) p.flags |= ARCLE_STEP_ROWS_INCREMENTAL;
  const dim3 g = grid_for(n_rows), b(64 * WAVES_PER_WG);
  hipStream_t st = (hipStream_t)stream;
  const int fw = width_class(e->base);
  switch (ingress) {
    case arcle::INGRESS_BBOX: launch_transition_ing<arcle::INGRESS_BBOX>(fw, g, b, st, p); break;
    case arcle::INGRESS_POINT: launch_transition_ing<arcle::INGRESS_POINT>(fw, g, b, st, p); break;
    case arcle::INGRESS_MASK: launch_transition_ing<arcle::INGRESS_MASK>(fw, g, b, st, p); break;
    default: return fail(e, ARCLE_ERR_ARG, "arcle_transition_rows takes mask, bbox or point selections");
  }
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

// plane <-> dense [n_envs][H*W] array (device or pinned host memory): a strided 2-D copy on the stream
extern "C" int arcle_get_plane(arcle_env* e, int plane, int8_t* dst, void* stream) {
  if (!e || !dst || plane < 0 || plane >= ARCLE_N_PLANES) return ARCLE_ERR_ARG;
  if (!e->bufs.plane[plane]) return fail(e, ARCLE_ERR_CONFIG, "this env kind has no such plane");
  DeviceGuard guard(e->device);
  HIP_TRY(e, hipMemcpy2DAsync(dst, (size_t)e->base.P, e->bufs.plane[plane], (size_t)e->base.PS, (size_t)e->base.P, (size_t)e->cfg.n_envs,
                              hipMemcpyDefault, (hipStream_t)stream));
  return ARCLE_OK;
}
extern "C" int arcle_set_plane(arcle_env* e, int plane, const int8_t* src, void* stream) {
  if (!e || !src || plane < 0 || plane >= ARCLE_N_PLANES) return ARCLE_ERR_ARG;
  if (!e->bufs.plane[plane]) return fail(e, ARCLE_ERR_CONFIG, "this env kind has no such plane");
  DeviceGuard guard(e->device);
  HIP_TRY(e, hipMemcpy2DAsync(e->bufs.plane[plane], (size_t)e->base.PS, src, (size_t)e->base.P, (size_t)e->base.P, (size_t)e->cfg.n_envs,
                              hipMemcpyDefault, (hipStream_t)stream));
  return ARCLE_OK;
}

extern "C" int arcle_packed_obs_size(const arcle_env* e) {
  if (!e) return ARCLE_ERR_ARG;
  return (e->base.P + 7 + 15) & ~15;
}

extern "C" int arcle_pack_obs(arcle_env* e, const int32_t* reward, const uint8_t* term, uint8_t* out, void* stream) {
  if (!e || !reward || !term || !out) return ARCLE_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(out) & 15) return fail(e, ARCLE_ERR_ARG, "packed observation rows must be 16-byte aligned");
  DeviceGuard guard(e->device);
  if (e->big) {
    arcle_big::BigParams q = big_params(e);
    q.reward = const_cast<int32_t*>(reward);
    q.term = const_cast<uint8_t*>(term);
    q.pack_out = out;
    return big_done(e, arcle_big::launch_rows(q, 1, stream), "arcle_pack_obs");
  }
  StepParams p = e->base;
  p.reward = const_cast<int32_t*>(reward);
  p.term = const_cast<uint8_t*>(term);
  p.flat_out = reinterpret_cast<int8_t*>(out);
  p.flat_stride = arcle_packed_obs_size(e);
  hipLaunchKernelGGL(arcle_pack_kernel, grid_for(p.n_envs), dim3(64 * WAVES_PER_WG), 0, (hipStream_t)stream, p);
  HIP_TRY(e, hipGetLastError());
  return ARCLE_OK;
}

extern "C" int arcle_set_packed_output(arcle_env* e, uint8_t* out) {
  if (!e) return ARCLE_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(out) & 15) return fail(e, ARCLE_ERR_ARG, "packed observation rows must be 16-byte aligned");
  e->pack_out = reinterpret_cast<int8_t*>(out);
  return ARCLE_OK;
}

__global__ void arcle_status_kernel(uint32_t* status, uint32_t* out, int clear) {
  *out = clear ? atomicExch(status, 0u) : atomicOr(status, 0u);
}

extern "C" int arcle_get_status(arcle_env* e, uint32_t* status, int clear, void* stream) {
  if (!e || !status) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  hipLaunchKernelGGL(arcle_status_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, e->d_status, e->d_status + 1, clear);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(status, e->d_status + 1, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
  return ARCLE_OK;
}

extern "C" int arcle_enable_accounting(arcle_env* e, int on) {
  if (!e) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  if (on && !e->d_acct) {
    HIP_TRY(e, hipMalloc((void**)&e->d_acct, (size_t)e->cfg.n_envs * 64));
    HIP_TRY(e, hipMemset(e->d_acct, 0, (size_t)e->cfg.n_envs * 64));
    e->acct_steps = 0;
  } else if (!on && e->d_acct) {
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipFree(e->d_acct));
    e->d_acct = nullptr;
  }
  return ARCLE_OK;
}

#ifdef ARCLE_TRACE_WAVES  // diagnostic builds only (tools/wavetrace.py, tools/launchtrace.py); absent from the shipped library and from the header
// entry / exit clocks (100 MHz) of every wave of the last two step launches of the LEAN kernels (no accounting): host_out uint64 [2][n_envs][2],
// half (seq & 1) written by launch number seq; *last_seq = the number of the most recent launch
extern "C" int arcle_debug_launch_trace(arcle_env* e, int enable, uint64_t* host_out, int* last_seq) {
  if (!e) return ARCLE_ERR_ARG;
  DeviceGuard guard(e->device);
  const size_t bytes = (size_t)e->cfg.n_envs * 2 * 2 * sizeof(uint64_t);
  if (enable && !e->d_trace) {
    HIP_TRY(e, hipMalloc((void**)&e->d_trace, bytes));
    HIP_TRY(e, hipMemset(e->d_trace, 0, bytes));
    e->trace_seq = 0;
  }
  if (host_out && e->d_trace) {
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(host_out, e->d_trace, bytes, hipMemcpyDeviceToHost));
    if (last_seq) *last_seq = e->trace_seq - 1;
  }
  return ARCLE_OK;
}
extern "C" int arcle_debug_copy_trace(arcle_env* e, uint64_t* host_out) {  // diagnostic builds only
  if (!e || !e->d_acct) return ARCLE_ERR_ARG;
  HIP_TRY(e, hipDeviceSynchronize());
  HIP_TRY(e, hipMemcpy(host_out, e->d_acct, (size_t)e->cfg.n_envs * 64, hipMemcpyDeviceToHost));
  return ARCLE_OK;
}
#endif

extern "C" int arcle_get_accounting_ex(arcle_env* e, uint64_t* bytes, uint64_t* issued, uint64_t* steps, int clear, void* stream) {
  if (!e || !bytes || !issued || !steps) return ARCLE_ERR_ARG;
  if (!e->d_acct) return fail(e, ARCLE_ERR_CONFIG, "accounting is not enabled");
  DeviceGuard guard(e->device);
  const size_t n = (size_t)e->cfg.n_envs;
  uint32_t* h = (uint32_t*)malloc(2 * n * 4);
  if (!h) return ARCLE_ERR_ARG;
  hipError_t err = hipMemcpyAsync(h, e->d_acct, 2 * n * 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (err == hipSuccess && clear) err = hipMemsetAsync(e->d_acct, 0, 2 * n * 4, (hipStream_t)stream);
  if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
  if (err != hipSuccess) {
    free(h);
    snprintf(e->err, sizeof(e->err), "accounting copy failed: %s", hipGetErrorString(err));
    return ARCLE_ERR_HIP;
  }
  uint64_t tot = 0, iss = 0;
  for (size_t i = 0; i < n; i++) tot += h[i], iss += h[n + i];
  free(h);
  *bytes = tot + e->acct_extra;
  *issued = iss + e->acct_extra;
  *steps = e->acct_steps;
  if (clear) e->acct_steps = 0, e->acct_extra = 0;
  return ARCLE_OK;
}

extern "C" int arcle_get_accounting(arcle_env* e, uint64_t* bytes, uint64_t* steps, int clear, void* stream) {
  uint64_t issued = 0;
  return arcle_get_accounting_ex(e, bytes, &issued, steps, clear, stream);
}
