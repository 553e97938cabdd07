// arcle_wave.h — the per-environment body of the ARCLE step kernel, written for ONE CDNA4 wavefront
// (64 lanes) per environment.
//
// Execution model
//   * one wavefront owns one env at a time (a wave may walk several envs, grid-stride); the operation index is
//     therefore wave-uniform and the op dispatch is a scalar branch (no intra-wave divergence on the op);
//   * everything that is per-env and not per-cell — the 16-byte record, the op index, the counters, a bbox / point
//     selection — is fetched with SCALAR loads straight into SGPRs and handled by the scalar ALU; the record stays
//     packed in its four dwords and fields are extracted / inserted on demand;
//   * lane L owns the 16 consecutive cells [16L, 16L+16) of the row-major H x W plane (one aligned dwordx4 per plane
//     per lane — every plane access of the wave is a single coalesced 16 B/lane instruction; the per-env plane
//     stride PS is a multiple of 16 B, 128 B by default);
//   * cell predicates live in 16-bit per-lane masks (bit k <-> cell 16L+k); byte planes live in 4 VGPRs; a mask
//     becomes a byte mask through a 256-entry LDS table (two ds_read_b64 instead of ~16 VALU);
//   * geometric ops (object lift / place, Copy, Paste, Crop) are *uniform flat shifts* of a plane: the plane is
//     staged once in the wave's private LDS tile and read back at a wave-uniform byte offset (dword reads +
//     v_alignbyte); Rotate/Flip are affine index gathers from the LDS tile;
//   * FloodFill runs on a flat bit-board (32 cells per lane) with cross-lane shifts;
//   * planes are written back at the point where they become final (write-through 16 B stores).
//
// This header is compiled by hipcc for gfx950 (arcle_hip.hip).  tests/emu/ compiles the very same header with g++
// against a lock-step 64-thread emulation of the cross-lane primitives (namespace xl) so that the kernel LOGIC can be
// checked against the oracle without a GPU; that emulator is test infrastructure and is never part of the product.
//
// Reference semantics restated here are cited per function (paths relative to /root/reference).
#pragma once
#include <stdint.h>

#include "../../include/arcle_hip.h"

#if !defined(ARCLE_DEV) || !defined(ARCLE_HD)
#error "include through arcle_hip.hip (or the test emulator), which defines ARCLE_DEV / ARCLE_HD and namespace xl"
#endif

namespace arcle {

enum { INGRESS_MASK = 0,    // int8 [N][P] selection masks as given (action['selection'], base.py:134-138)
       INGRESS_BBOX = 1,    // int32 [N][4] BBoxWrapper corners (bbox.py:22-30)
       INGRESS_POINT = 2,   // int32 [N][2] PointWrapper cell (bbox.py:43-49)
       INGRESS_BBOX5 = 3,   // int32 [N][5] the BBoxWrapper action as ONE record (x1, y1, x2, y2, operation): selection form = BBOX
       INGRESS_BITS = 4,    // uint8 [N][128] bit-packed boolean masks (bit f of the row = cell f truthy; base.py:136 accepts bool)
       INGRESS_BBOX5_PF = 5 };  // launcher-internal: BBOX5 read from a device staging buffer, while ARCLE_PF_BLOCKS extra workgroups at the
                                // front of the grid copy the NEXT step's records (pinned host memory) into the other staging buffer
#define ARCLE_PF_BLOCKS 8
#define ARCLE_BITS_STRIDE (ARCLE_MAX_CELLS / 8)
// launcher-internal bit of a compile-time flag set (FL template parameter): the fused flat rows are the FilterO2ARC subset
#define ARCLE_STEPX_FLAT_FILTERED 0x10000
// (0x20000: the round-3/4 table form of ordered dispatch, retired in round 5)
// ... the grid plane is requested speculatively beside the per-env scalar loads — one memory round trip per wave instead of two dependent
// ones.  Pays where a launch is made of memory latency: the streaming regime (state beyond the 256 MiB Infinity Cache, 16+ occupancy rounds
// of waves that each wait on HBM) and batches of at most a wave or two per SIMD (the launch IS one wave's latency chain); at 8192 envs —
// one occupancy round, issue-bound — it loses (profiles/round3_experiments.txt), so it is an instantiation of its own
#define ARCLE_STEPX_STREAM 0x40000
// ... with non-temporal plane stores / a non-temporal speculative load (which pays where depends on the batch size: the launcher picks)
#define ARCLE_STEPX_STORE_NT 0x80000
#define ARCLE_STEPX_EARLY_NT 0x100000
// ... the launch orders ITSELF (round 5): dispatch slots are dealt to the envs in groups of 16 — one slot from each of the 16 strata the hardware
// starts an XCD's workgroups in, 16 contiguous envs — and every wave derives, from the group's 16 op indices alone, which env of the group its
// slot steps: object operations (the longest waves) go to the slots that start first.  No table, no hint, no barrier (see the kernel).
#define ARCLE_STEPX_GROUPED 0x200000
// row strides of the 30 x 30 lean instantiations: 3*900 + 10 and 7*900 + 14, rounded up to 16
#define ARCLE_ROW30_FILTERED_STRIDE 2720
#define ARCLE_ROW30_FULL_STRIDE 6320
ARCLE_HD constexpr bool is_tuple(int ing) { return ing == INGRESS_BBOX || ing == INGRESS_POINT || ing == INGRESS_BBOX5 || ing == INGRESS_BBOX5_PF; }
ARCLE_HD constexpr bool is_cells(int ing) { return ing == INGRESS_MASK || ing == INGRESS_BITS; }
// FW (instantiation parameter): grid-width class of the launch
enum { FW_GENERIC = 0,  // any W
       FW_FAST = 1,     // 16 <= W <= 32: a lane's 16-cell window spans at most two rows (cheap rectangle masks, DPP flood fill)
       FW_FULL = 2 };   // FW_FAST and PS == 1024: all 64 lanes hold cells of the plane row, no lane predication on plane I/O

struct StepParams {
  // ---- step / rollout kernels ----
  int8_t* plane[ARCLE_N_PLANES];
  int8_t* rec;
  int32_t* cnt;
  const int32_t* op;
  const void* sel;  // ingress payload: int8 [N][P] | int32 [N][4] | int32 [N][2] | int32 [N][5] | uint8 [N][128]
  int32_t* reward;
  uint8_t* term;
  int32_t n_envs, H, W, P;
  int32_t PS;  // plane stride in bytes (multiple of 16, >= P)
  int32_t n_ops, max_trial, ingress;
  uint32_t flags;
  uint32_t div_magic;  // floor(65536/W)+1 : (n*div_magic)>>16 == n/W for n < 1040 (checked at create)
  int32_t nseg;        // max row segments a 16-cell lane window can span
  int32_t n_steps;     // rollout kernel only: steps per launch
  int32_t flat_seq;    // != 0 with a row tail: byte 15 of the tail carries this sequence number, stored LAST behind a system-scope release
                       // (arcle_set_flat_seq: the completion signal a host polls in pinned memory instead of synchronising the stream)
  int32_t step_limit;  // ARCLE_STEP_TRUNCATE: truncated = action_steps >= step_limit (TimeLimit, agents/train.py:67)
  uint32_t* status;
  uint32_t* acct;         // optional per-env algorithmic-byte accumulator (ACCT instantiations)
  const uint32_t* d_ops;  // device copy of the op table, ARCLE_MAX_OPS + 1 entries (unused slots 0; slot n_ops is always one)
  uint8_t* trunc;         // optional truncated output (ARCLE_STEP_TRUNCATE)
  int32_t* dense;         // optional int32 [N][2] = (correct cells, total cells) of the dense reward (ARCLE_STEP_DENSE)
  int8_t* flat_out;       // optional flattened observation rows (ARCLE_STEP_FLAT_OBS / flatten kernel)
  int32_t flat_stride;    // bytes between rows of flat_out
  int32_t flat_filter;    // 0 = full state (FlattenObservation), 1 = FilterO2ARC subset (agents/env.py:109-126)
  int8_t* pack_out;       // optional packed per-step rows (ARCLE_STEP_PACK_OBS), stride = packed_stride(P)
  // ---- reset kernels ----
  const uint8_t* rmask;
  const int32_t* task_idx;                 // reset-from-table kernel only
  const int8_t *tbl_in, *tbl_ans;          // task table planes [n_tasks][PS]
  const int8_t *tbl_in_dim, *tbl_ans_dim;  // task table dims   [n_tasks][2]
  int32_t n_tasks;
  uint32_t aug_flags;  // ARCLE_AUG_*: colour permutation / rot90 augmentation at reset (agents/env.py:31-42)
  uint64_t seed;       // RNG seed of the device-side task resampling / augmentation (keyed by global env id)
  int64_t env_base;    // global id of this handle's env 0 (multi-GPU shards)
  int32_t* episode;    // int32 [N] episodes started so far per env (RNG stream position); may be NULL
  int32_t* cur_task;   // int32 [N] task-table index currently loaded per env; may be NULL
  const int32_t *pair_off, *pair_cnt;  // resample: per-task first table entry / number of entries, int32 [n_problems]
  const uint8_t* aug_k;     // explicit augmentation (ARCLE_AUG_EXPLICIT): rot90 count per env, uint8 [N]
  const uint8_t* aug_perm;  // explicit colour permutation per env, uint8 [N][16] (perm[c], c < 10)
  int32_t n_problems;
  int32_t wpw;  // step kernel: waves per workgroup of this launch (set by the launcher)
  int32_t flat_tail;       // 1 = the last 16 bytes of every flat row's stride hold the step outputs (arcle_set_flat_output_ex)
  const int8_t* rows_in;   // state-row kernels: flattened state rows to read (arcle_transition_rows / arcle_set_state_rows)
  int32_t rows_in_stride;
  int32_t n_resident;      // state-row kernels: envs of the handle (src_env range check); n_envs = rows of the launch
  int32_t* dense_cache;    // library-owned int32 [N][2]: the dense pair of the env's CURRENT grid, (0, 0) = unknown (see step_core)
  const int32_t* next_sel; // INGRESS_BBOX5_PF: the NEXT step's records (pinned host memory, int32 [N][5]) ...
  int32_t* stage_out;      // ... and the device staging buffer the front workgroups copy them into while this step runs
  // ---- dispatch order (ARCLE_STEPX_GROUPED instantiations: launches that order themselves) ----
  uint64_t long_mask;      // bit i: op table slot i is an object operation (Move / Rotate / Flip: the longest-running waves)
  uint32_t group_magic;    // != 0: the launch orders itself (ARCLE_STEPX_GROUPED instantiations): floor(2^32 / (n_envs / 128)) + 1
  int32_t spec_grid;       // 1: the launch's lean twin loads the grid plane speculatively (ARCLE_STEPX_STREAM); the accounting instantiation
                           // then counts that load as issued also for the steps that never use it
};

// 16 bytes of a plane = 4 VGPRs; a first-class vector value so that it always lives in registers
#if defined(__clang__)
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
typedef uint32_t U2 __attribute__((ext_vector_type(2)));
#else
typedef uint32_t U4 __attribute__((vector_size(16)));
typedef uint32_t U2 __attribute__((vector_size(8)));
#endif

struct I2 {
  int32_t x, y;
};

struct WaveLDS {
  uint32_t a[256];  // 1024 B staging tile (bytes of one plane)
  uint32_t b[256];  // second tile (object_sel during Rotate/Flip)
};
// LDS of one workgroup: the mask-expansion table + one private tile pair per wave
template <int WAVES>
struct BlockLDS {
  U2 lut[256];  // lut[b] = 8 bytes, byte k = 0xff iff bit k of b
  WaveLDS wave[WAVES];
};

// ------------------------------------------------------------------------------------------------
// small bit helpers (per lane)
// ------------------------------------------------------------------------------------------------
ARCLE_DEV int imin(int a, int b) { return a < b ? a : b; }
ARCLE_DEV int imax(int a, int b) { return a > b ? a : b; }
ARCLE_DEV uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
ARCLE_DEV int i8w(int x) { return (int)(int8_t)(uint8_t)(x & 0xff); }  // wrap to int8
ARCLE_DEV int floordiv2(int a) { return a >> 1; }                       // arithmetic shift == floor(a/2)
ARCLE_DEV uint32_t bits_range(int a, int b) { return (2u << b) - (1u << a); }  // bits a..b, 0<=a<=b<=30

// bit7-per-byte flags: byte != 0
ARCLE_DEV uint32_t nzflags(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }
// bit7-per-byte flags: byte > 0 as int8 (non-zero and sign bit clear)
ARCLE_DEV uint32_t posflags(uint32_t x) { return ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) & ~x & 0x80808080u; }
// sixteen 0x80-per-byte flags (four dwords) -> 16-bit mask, bit k <-> byte k: v_dot4_u32_u8 with the weights 1, 2, 4, 8 (and 16 ..
// 128 for the second dword) sums 128 * 2^k over the flagged bytes — 4 dot products + 2 shifts instead of ~28 shift / or / and steps
ARCLE_DEV uint32_t flags16(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
  uint32_t lo = xl::dot4(t0, 0x08040201u, 0u);
  lo = xl::dot4(t1, 0x80402010u, lo);
  uint32_t hi = xl::dot4(t2, 0x08040201u, 0u);
  hi = xl::dot4(t3, 0x80402010u, hi);
  return (lo >> 7) | (hi << 1);  // lo, hi = 128 * (8-bit value)
}
ARCLE_DEV uint32_t nz16(const U4& v) { return flags16(nzflags(v[0]), nzflags(v[1]), nzflags(v[2]), nzflags(v[3])); }
ARCLE_DEV uint32_t pos16(const U4& v) { return flags16(posflags(v[0]), posflags(v[1]), posflags(v[2]), posflags(v[3])); }
// 0xff for every byte that is > 0 as int8
ARCLE_DEV U4 posbytes(const U4& v) {
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t t = posflags(v[i]);  // 0x80 in qualifying bytes
    r[i] = t | (t - (t >> 7));
  }
  return r;
}
// base + unsigned 32-bit byte offset: scalar loads take the offset in an SGPR, vector accesses as the 32-bit VGPR offset of the
// saddr form — no 64-bit address arithmetic per access
template <typename T>
ARCLE_DEV const T* at(const T* base, uint32_t byte_off) {
  return reinterpret_cast<const T*>(reinterpret_cast<uintptr_t>(base) + byte_off);
}
template <typename T>
ARCLE_DEV T* at(T* base, uint32_t byte_off) {
  return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(base) + byte_off);
}

ARCLE_DEV U4 u4_zero() {
  U4 r;
  r[0] = r[1] = r[2] = r[3] = 0;
  return r;
}
ARCLE_DEV U4 u4_and(const U4& a, const U4& b) {
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = a[i] & b[i];
  return r;
}
ARCLE_DEV U4 u4_and1(const U4& a, uint32_t c) {
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = a[i] & c;
  return r;
}
ARCLE_DEV U4 u4_andn(const U4& a, const U4& m) {  // a & ~m
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = a[i] & ~m[i];
  return r;
}
ARCLE_DEV U4 u4_sel(const U4& m, const U4& a, const U4& b) {  // m ? a : b  (bytewise)
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = (a[i] & m[i]) | (b[i] & ~m[i]);
  return r;
}
ARCLE_DEV U4 u4_sel1(const U4& m, uint32_t a, const U4& b) {  // m ? a(uniform dword) : b
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = (a & m[i]) | (b[i] & ~m[i]);
  return r;
}
ARCLE_DEV uint32_t eq16(const U4& v, uint32_t byte) {  // bytes == byte
  const uint32_t c = (byte & 0xffu) * 0x01010101u;
  return flags16(nzflags(v[0] ^ c), nzflags(v[1] ^ c), nzflags(v[2] ^ c), nzflags(v[3] ^ c)) ^ 0xffffu;
}
ARCLE_DEV uint32_t u4_byte(const U4& v, int k) {  // dynamic byte extract
  uint32_t w = (k & 8) ? ((k & 4) ? v[3] : v[2]) : ((k & 4) ? v[1] : v[0]);
  return (w >> (8 * (k & 3))) & 0xffu;
}

// ------------------------------------------------------------------------------------------------
// the 16-byte scalar record, kept packed in its four dwords (wave-uniform: SGPRs); byte offsets ARCLE_REC_*
// ------------------------------------------------------------------------------------------------
struct Rec {
  uint32_t w[4];
  ARCLE_DEV int sb(int f) const { return (int)((int32_t)(w[f >> 2] << (24 - 8 * (f & 3))) >> 24); }  // signed byte f
  ARCLE_DEV int ub(int f) const { return (int)((w[f >> 2] >> (8 * (f & 3))) & 0xffu); }               // unsigned byte f
  ARCLE_DEV void put(int f, int v) {
    const int sh = 8 * (f & 3);
    w[f >> 2] = (w[f >> 2] & ~(0xffu << sh)) | (((uint32_t)v & 0xffu) << sh);
  }
  ARCLE_DEV void put2(int f, int a, int b) {  // bytes f, f+1 (f even)
    const int sh = 8 * (f & 3);
    w[f >> 2] = (w[f >> 2] & ~(0xffffu << sh)) | ((((uint32_t)a & 0xffu) | (((uint32_t)b & 0xffu) << 8)) << sh);
  }
  // dims are 0..127 (never negative); object_pos and trials_remain are signed
  ARCLE_DEV int in_h() const { return ub(0); }
  ARCLE_DEV int in_w() const { return ub(1); }
  ARCLE_DEV int gh() const { return ub(2); }
  ARCLE_DEV int gw() const { return ub(3); }
  ARCLE_DEV int ch() const { return ub(4); }
  ARCLE_DEV int cw() const { return ub(5); }
  ARCLE_DEV int oh() const { return ub(6); }
  ARCLE_DEV int ow() const { return ub(7); }
  ARCLE_DEV int ox() const { return sb(8); }
  ARCLE_DEV int oy() const { return sb(9); }
  ARCLE_DEV int trials() const { return sb(10); }
  ARCLE_DEV int term() const { return ub(11); }
  ARCLE_DEV int active() const { return ub(12); }
  ARCLE_DEV int parity() const { return ub(13); }
  ARCLE_DEV int ah() const { return ub(14); }
  ARCLE_DEV int aw() const { return ub(15); }
};

// ------------------------------------------------------------------------------------------------
// wave context
// ------------------------------------------------------------------------------------------------
struct Wave {
  const StepParams& p;
  WaveLDS* lds;
  const U2* lut;
  int lane;
  int r0, c0;        // row / col of this lane's first cell
  uint32_t valid16;  // cells of this lane that exist (flat index < P)
  bool live;         // lane holds bytes of the plane row (16*lane < PS); always true for FW_FULL
  // 16 <= W <= 32: the lane's window covers row r0 from column c0 (k1 cells, mask lm) and then row r0+1
  // from column 0; rectangle masks then cost a dozen VALU ops (rect16 fast path)
  int fw;
  int k1;
  uint32_t lm;
  int ingress;  // INGRESS_* (a compile-time constant of the kernel instantiation)
  // current env
  int env;
  uint32_t poff;  // byte offset of this lane's 16 cells inside a plane: env*PS + 16*lane (< 4 GiB, checked at create)
  // rollout mode: the env's planes stay in registers across steps; load/store then never touch HBM
  bool resident;
  mutable U4 cache[ARCLE_N_PLANES];
  mutable uint32_t dirty;  // planes of `cache` that differ from HBM
  // state-row kernels: the planes come out of a flattened state row ON DEMAND (a plane the op never reads is never held in
  // registers: the row writer passes it through at the end, or — in place — leaves it alone); `have` = planes already in `cache`
  const int8_t* row_src;
  int answer_env;
  mutable uint32_t have;
  // accounting instantiations: bytes of global-memory accesses this wave ISSUED (every plane / table / row access, whole 16-byte
  // lanes incl. row padding) — next to the algorithmic figure of SURVEY.md 8d it shows what the implementation really moves
  bool count;
  mutable uint32_t issued;
  mutable uint32_t stored;  // planes written (through `store`) since the wave picked up its env: what changed in this step
  bool store_nt;            // ARCLE_STEPX_STORE_NT instantiations: plane stores are non-temporal instead of write-through
  bool table;               // expand16 reads the workgroup's LDS table (false: computes the byte masks on the vector ALUs)

  // (table_: a compile-time constant at every call site — a null test of `lut_` is not: LDS offset 0 is a valid address, and a kernel
  // that cannot fold the test carries both expansions)
  ARCLE_DEV Wave(const StepParams& p_, WaveLDS* l, const U2* lut_, int lane_, int ingress_, int fw_, bool resident_, bool count_ = false,
                 bool table_ = true)
      : p(p_), lds(l), lut(lut_), lane(lane_) {
    table = table_;
    count = count_;
    issued = 0;
    stored = 0;
    store_nt = false;
    ingress = (ingress_ == INGRESS_BBOX5 || ingress_ == INGRESS_BBOX5_PF) ? INGRESS_BBOX : ingress_;  // (the record forms only differ in where the kernel loads from)
    fw = fw_;
    resident = resident_;
    dirty = 0;
    row_src = nullptr;
    answer_env = 0;
    have = ~0u;
    const uint32_t f0 = 16u * (uint32_t)lane;
    r0 = (int)(xl::mul24(f0, p.div_magic) >> 16);  // f0 < 1024, div_magic < 2^17
    c0 = (int)f0 - (int)xl::mul24((uint32_t)r0, (uint32_t)p.W);
    const int nv = imin(imax(p.P - (int)f0, 0), 16);
    valid16 = (1u << nv) - 1u;
    live = fw == FW_FULL ? true : (int)f0 < p.PS;
    k1 = imin(16, p.W - c0);
    lm = (1u << k1) - 1u;
    env = 0;
    poff = f0;
  }
  ARCLE_DEV void set_env(int e) {
    env = e;
    poff = (uint32_t)e * (uint32_t)p.PS + 16u * (uint32_t)lane;  // (uniform product: scalar ALU)
  }

  // ---- plane I/O: one aligned 16 B access per lane -------------------------------------------
  ARCLE_DEV U4 load_hbm(int pl) const {
    U4 v = u4_zero();
    if (live) v = xl::load16(p.plane[pl], poff);
    if (count) issued += (uint32_t)p.PS;
    return v;
  }
  ARCLE_DEV void store_hbm(int pl, const U4& v) const {
    if (live) {
      if (store_nt) xl::store16_nt(p.plane[pl], poff, v);
      else xl::store16(p.plane[pl], poff, v);
    }
    if (count) issued += (uint32_t)p.PS;
  }
  ARCLE_DEV U4 load_from_row(int pl) const;  // (defined with the row layout, below)
  ARCLE_DEV U4 load(int pl) const {
    if (!resident) return load_hbm(pl);
    if (!(have & (1u << pl))) {
      cache[pl] = load_from_row(pl);
      have |= 1u << pl;
    }
    return cache[pl];
  }
  ARCLE_DEV void store(int pl, const U4& v) const {
    stored |= 1u << pl;
    if (resident) {
      cache[pl] = v;
      have |= 1u << pl;
      dirty |= 1u << pl;
    } else {
      store_hbm(pl, v);
    }
  }

  // ---- 16-bit cell mask -> 16 byte masks (0x00 / 0xff), through the workgroup's LDS table ----------
  // (table == false — the step kernel since round 4: the expansion on the vector ALUs, 5 instructions per four cells, no table to build and
  // no workgroup barrier in front of the op; the rollout kernel, whose steps are pure instruction issue, keeps the table: 2.62 vs 3.77 us
  // per step without it — profiles/round4_experiments.txt §11)
  ARCLE_DEV U4 expand16(uint32_t m) const {
    if (!table) {
      U4 e;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t x = xl::mul24((m >> (4 * i)) & 0xfu, 0x00204081u) & 0x01010101u;
        e[i] = (x << 8) - x;
      }
      return e;
    }
    const U2 lo = lut[m & 0xffu], hi = lut[(m >> 8) & 0xffu];
    U4 r;
    r[0] = lo[0];
    r[1] = lo[1];
    r[2] = hi[0];
    r[3] = hi[1];
    return r;
  }

  // ---- 16-bit mask of this lane's cells inside rows [x1,x2] x cols [y1,y2] (inclusive) ----------
  // Callers pass rectangles inside the H x W plane (x2 < H), so cells beyond P never qualify in the fast form.
  ARCLE_DEV uint32_t rect16(int x1, int x2, int y1, int y2) const {
    if (fw != FW_GENERIC) {
      // column mask of the rectangle (wave-uniform, scalar ALU); empty rectangle -> 0
      const uint32_t cm = (x1 <= x2 && y1 <= y2) ? ((2u << y2) - (1u << y1)) : 0u;
      const uint32_t dx = (uint32_t)(x2 - x1);
      const uint32_t d0 = (uint32_t)(r0 - x1);
      const uint32_t s0 = (d0 <= dx) ? ((cm >> c0) & lm) : 0u;
      const uint32_t s1 = (d0 + 1u <= dx) ? (cm << k1) : 0u;
      return (s0 | s1) & 0xffffu;
    }
    uint32_t m = 0;
    int r = r0, c = c0, k = 0;
    for (int s = 0; s < p.nseg; s++) {
      if (k < 16) {
        int len = imin(p.W - c, 16 - k);
        int lo = imax(y1, c), hi = imin(y2, c + len - 1);
        if (r >= x1 && r <= x2 && lo <= hi) m |= bits_range(k + lo - c, k + hi - c);
        k += len;
        c = 0;
        r++;
      }
    }
    return m & valid16;
  }

  // ---- wave reductions (butterfly over ds_bpermute) -------------------------------------------
  ARCLE_DEV int wave_sum(int v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (int)xl::shfl((uint32_t)v, lane ^ o);
    return (int)xl::uniform((uint32_t)v);
  }
  ARCLE_DEV int wave_min(int v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = imin(v, (int)xl::shfl((uint32_t)v, lane ^ o));
    return (int)xl::uniform((uint32_t)v);
  }
  ARCLE_DEV int wave_max(int v) const {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = imax(v, (int)xl::shfl((uint32_t)v, lane ^ o));
    return (int)xl::uniform((uint32_t)v);
  }
  ARCLE_DEV bool any(bool b) const { return xl::ballot(b) != 0ull; }

  // ---- LDS staging + uniform flat shift ---------------------------------------------------------
  ARCLE_DEV void stage(uint32_t* buf, const U4& v) const {
    xl::lds_fence();  // earlier reads of this tile are complete
    *reinterpret_cast<U4*>(buf + 4 * lane) = v;
    xl::lds_fence();
  }
  // out[f] = staged[f + S] for this lane's 16 cells.  Cells whose source falls outside the tile get
  // unspecified bytes — every caller masks the result with a rectangle that excludes them.
  ARCLE_DEV U4 shifted(const uint32_t* buf, int S) const {
    int base = 4 * lane + (S >> 2);
    uint32_t sh = (uint32_t)S & 3u;
    uint32_t wd[5];
#pragma unroll
    for (int j = 0; j < 5; j++) wd[j] = buf[xl::lds_idx(base + j, 256)];
    U4 r;
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = xl::alignbyte(wd[j + 1], wd[j], sh);
    return r;
  }
  // the same shift for a 16-bit-per-lane cell mask: out bit k = in bit (16L + k + S)
  ARCLE_DEV uint32_t shifted_bits(uint32_t m16, int S) const {
    int q = S >> 4;
    uint32_t rb = (uint32_t)S & 15u;
    int l0 = lane + q, l1 = l0 + 1;
    uint32_t w0 = xl::shfl(m16, l0 & 63), w1 = xl::shfl(m16, l1 & 63);
    if (l0 < 0 || l0 > 63) w0 = 0;
    if (l1 < 0 || l1 > 63) w1 = 0;
    return ((w0 | (w1 << 16)) >> rb) & 0xffffu;
  }
};

// fills the workgroup's mask-expansion table; every thread of the workgroup calls it, followed by xl::wg_barrier()
// (workgroups that use the table have >= 256 threads; the emulator's single wave passes `nthreads` = 64)
// (every thread writes entry tid & 255: in a 512-thread workgroup both halves store the same values — no exec masking, whose
// s_and_saveexec / s_or are scalar instructions on every wave's path)
ARCLE_DEV void lut_init(U2* lut, int tid, int nthreads = 256) {
  for (int b = tid & 255; b < 256; b += nthreads) {
    const uint32_t lo = (uint32_t)b & 0xfu, hi = ((uint32_t)b >> 4) & 0xfu;
    const uint32_t x = (lo * 0x00204081u) & 0x01010101u, y = (hi * 0x00204081u) & 0x01010101u;
    U2 e;
    e[0] = xl::opaque(x << 8) - x;  // x * 0xff without the quarter-rate 32-bit multiply the compiler would pick
    e[1] = xl::opaque(y << 8) - y;
    lut[b] = e;
  }
}

// ------------------------------------------------------------------------------------------------
// selection ingress: action['selection'] as a cell mask (bbox.py:22-30, :43-49 fused on device)
// ------------------------------------------------------------------------------------------------
struct Sel {
  uint32_t nz;   // truthy cells (np.any(sel), ma mask, logical_and)
  uint32_t pos;  // cells with sel > 0
  U4 vals;       // the raw int8 values (for `selected = sel`, object.py:96 / keep_sel :38); mask ingress only
  bool any_nz, any_pos;
  int x0, x1, y0, y1;  // _get_bbox of the truthy cells (object.py:49-58); valid iff any_nz
  bool is_rect;        // built from a bbox / point tuple: every cell of the bbox is 1
  int one_cell;        // mask payload (16 <= W <= 32): flat index of the ONLY truthy cell, -1 if there are none or several
};

// Cell masks of the selection.  For a bbox / point tuple they are a rectangle mask built where an op needs it (most ops of
// the O2ARC table read only the tuple: FloodFill, Copy, Paste, CropGrid, ResizeGrid), for a mask payload ingest_cells made them.
ARCLE_DEV uint32_t sel_nz(const Wave& w, const Sel& s) {
  return (s.is_rect && is_tuple(w.ingress)) ? w.rect16(s.x0, s.x1, s.y0, s.y1) : s.nz;
}
// any cell with sel > 0 ?
ARCLE_DEV bool sel_any_pos(const Wave& w, const Sel& s);
ARCLE_DEV uint32_t sel_pos(const Wave& w, const Sel& s) {
  if (w.ingress == INGRESS_MASK) return pos16(s.vals) & w.valid16;  // (derived where an op needs it: object ops, Copy, Paste)
  if (w.ingress == INGRESS_BITS) return s.nz;                        // boolean masks: truthy == positive
  return w.rect16(s.x0, s.x1, s.y0, s.y1);
}

ARCLE_DEV bool sel_any_pos(const Wave& w, const Sel& s) {
  if (is_tuple(w.ingress)) return s.any_pos;
  if (w.ingress == INGRESS_BITS) return s.any_nz;
  return s.any_nz && w.any(sel_pos(w, s) != 0);
}

// the raw int8 selection values (`selected = sel`, keep_sel object.py:38; mask ingress keeps what it loaded)
ARCLE_DEV U4 sel_values(const Wave& w, const Sel& s) {
  if (w.ingress == INGRESS_MASK) return s.vals;
  return u4_and1(w.expand16(sel_nz(w, s)), 0x01010101u);  // tuples and bit masks: ones where selected
}

// The selection payload of this env.  bbox = 4 ints, point = 2 ints: wave-uniform scalar loads; mask = this lane's
// 16 cells of the contiguous int8 [N][P] array the caller holds (no 16 B alignment guarantee).
ARCLE_DEV U4 load_payload(const Wave& w, int env, size_t step, const void* sel) {  // step: rollout kernels index [step][env]
  const StepParams& p = w.p;
  U4 v = u4_zero();
  const size_t e = step * (size_t)p.n_envs + (size_t)env;
  if (w.ingress == INGRESS_BBOX) {
    v = xl::uload4(reinterpret_cast<const int32_t*>(sel) + 4 * e);
  } else if (w.ingress == INGRESS_POINT) {
    const U2 b = xl::uload2(reinterpret_cast<const int32_t*>(sel) + 2 * e);
    v[0] = b[0];
    v[1] = b[1];
  } else if (w.ingress == INGRESS_BITS) {
    // this lane's 16 cells are 16 consecutive bits of the env's 128-byte row: one 2-byte load, no byte -> bit reduction at all
    v[0] = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(sel) + e * (size_t)ARCLE_BITS_STRIDE + 2 * w.lane);
  } else {
    const int8_t* src = reinterpret_cast<const int8_t*>(sel) + e * (size_t)p.P + 16 * w.lane;
    if ((p.P & 3) == 0 && ((reinterpret_cast<uintptr_t>(sel) & 3) == 0)) {
      // rows of P bytes are only 4-byte aligned: a lane whose 16 bytes lie inside the row issues ONE dword-aligned 16-byte load
      // (unaligned vector access is legal on gfx9+ global memory), the lane holding the row's tail loads dword by dword
      typedef U4 __attribute__((aligned(4))) U4a4;
      if (16 * w.lane + 16 <= p.P) {
        v = *reinterpret_cast<const U4a4*>(src);
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (16 * w.lane + 4 * i < p.P) v[i] = *reinterpret_cast<const uint32_t*>(src + 4 * i);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (16 * w.lane + k < p.P) v[k >> 2] |= (uint32_t)(uint8_t)src[k] << (8 * (k & 3));
    }
  }
  return v;
}

// the same through vector loads (every lane reads the uniform address); used where the value is prefetched far ahead
ARCLE_DEV U4 load_payload_v(const Wave& w, int env, size_t step) {
  const StepParams& p = w.p;
  const size_t e = step * (size_t)p.n_envs + (size_t)env;
  U4 v = u4_zero();
  if (w.ingress == INGRESS_BBOX) {
    v = *reinterpret_cast<const U4*>(reinterpret_cast<const int32_t*>(p.sel) + 4 * e);
  } else if (w.ingress == INGRESS_POINT) {
    const uint32_t* b = reinterpret_cast<const uint32_t*>(p.sel) + 2 * e;
    v[0] = b[0];
    v[1] = b[1];
  } else {
    v = load_payload(w, env, step, p.sel);
  }
  return v;
}

// Ingest: `ingest_scalar` is the wave-uniform part of a bbox / point tuple (returns false for a tuple outside the wrappers'
// action space; the caller raises ARCLE_ST_BAD_SELECTION), `ingest_cells` produces the per-lane cell masks (and everything
// for a mask payload).
struct StepOut;
ARCLE_DEV void raise_status(const StepParams& p, StepOut& out, uint32_t bits);
// (scalar instructions are what the step kernel is short of — profiles/round3_experiments.txt — so the common case is ONE unsigned
// test: a sorted, clipped tuple is non-empty iff both lower corners lie inside the plane; a negative coordinate fails it as a huge
// unsigned value and is told apart, normalised to an empty rectangle and reported inside the rare branch)
ARCLE_DEV void ingest_scalar(const Wave& w, Sel& s, const U4& payload, StepOut& out) {
  const StepParams& p = w.p;
  if (w.ingress == INGRESS_BBOX) {
    // BBoxWrapper.action (bbox.py:22-30): sort the corners, sel[x1:x2+1, y1:y2+1] = 1 (slices clip at H, W;
    // negative coordinates are outside the wrapper's Discrete action space and select nothing here)
    const int bx1 = (int)payload[0], by1 = (int)payload[1], bx2 = (int)payload[2], by2 = (int)payload[3];
    int xa = imin(bx1, bx2), xb = imin(imax(bx1, bx2), p.H - 1);
    int ya = imin(by1, by2), yb = imin(imax(by1, by2), p.W - 1);
    const bool any = p.H == p.W ? umax((uint32_t)xa, (uint32_t)ya) < (uint32_t)p.H : ((uint32_t)xa < (uint32_t)p.H && (uint32_t)ya < (uint32_t)p.W);
    if (!any) {
      if (xl::rare_v(xa | ya) < 0) {
        xa = xb + 1;
        raise_status(p, out, ARCLE_ST_BAD_SELECTION);
      }
    }
    s.x0 = xa; s.x1 = xb; s.y0 = ya; s.y1 = yb;
    s.any_nz = s.any_pos = any;
  } else if (w.ingress == INGRESS_POINT) {
    // PointWrapper.action (bbox.py:43-49)
    const int x = (int)payload[0], y = (int)payload[1];
    const bool ok = (uint32_t)x < (uint32_t)p.H && (uint32_t)y < (uint32_t)p.W;
    s.x0 = x; s.x1 = x; s.y0 = s.y1 = y;
    if (!ok) {
      s.x1 = x - 1;
      raise_status(p, out, ARCLE_ST_BAD_SELECTION);
    }
    s.any_nz = s.any_pos = ok;
  } else {
    return;
  }
  s.is_rect = true;
  s.one_cell = -1;
}
// `want_rect`: the op can take its rectangle shortcuts (object ops, Copy, Crop) — worth testing whether a mask IS its bounding box
// `want_bbox`: the op reads the selection's bounding box at all (Color and FloodFill do not)
ARCLE_DEV void ingest_cells(const Wave& w, Sel& s, const U4& payload, bool want_rect = false, bool want_bbox = true) {
  const StepParams& p = w.p;
  if (is_tuple(w.ingress)) {  // (masks on demand: sel_nz / sel_pos)
    s.nz = s.pos = 0;
    s.vals = u4_zero();
    return;
  }
  const bool bits = w.ingress == INGRESS_BITS;
  const U4 v = payload;
  s.is_rect = false;
  s.one_cell = -1;
  s.nz = (bits ? v[0] : nz16(v)) & w.valid16;
  // (bit masks: the int8 view — ones where selected — is only materialised where something reads it: sel_values, and here for the
  // generic-width FloodFill that sums the values)
  s.vals = bits ? (w.fw == FW_GENERIC ? u4_and1(w.expand16(s.nz), 0x01010101u) : u4_zero()) : v;
  s.pos = 0;  // (masks: sel_pos / sel_any_pos derive it from `vals` on demand)
  s.any_nz = w.any(s.nz != 0);
  s.any_pos = false;
  s.x0 = s.x1 = s.y0 = s.y1 = 0;
  if (s.any_nz && w.fw != FW_GENERIC) {
    // _get_bbox (object.py:49-58) for 16 <= W <= 32, on the scalar unit: rows from the first / last truthy flat index (one ballot
    // + two v_readlane), columns from the OR of every window's column bits (one DPP OR-reduction, no LDS round trips)
    const unsigned long long lanes = xl::ballot(s.nz != 0);
    const int l0 = __builtin_ctzll(lanes), l1 = 63 - __builtin_clzll(lanes);
    const uint32_t m0 = xl::readlane(s.nz, l0), m1 = xl::readlane(s.nz, l1);
    const int fmin = 16 * l0 + __builtin_ctz(m0), fmax = 16 * l1 + 31 - __builtin_clz(m1);
    if (fmin == fmax) s.one_cell = fmin;
    if (!want_bbox) return;
    s.x0 = (int)(((uint32_t)fmin * p.div_magic) >> 16);
    s.x1 = (int)(((uint32_t)fmax * p.div_magic) >> 16);
    const uint32_t cols = xl::wave_or(((s.nz & w.lm) << w.c0) | (s.nz >> w.k1));  // first row segment at column c0, second at column 0
    s.y0 = __builtin_ctz(cols);
    s.y1 = 31 - __builtin_clz(cols);
    if (want_rect) {
      // a mask that is exactly its bounding box filled with ones (what BBoxWrapper / PointWrapper produce on the host,
      // bbox.py:22-30,43-49) takes the rectangle paths of the ops: same results, far fewer instructions
      const uint32_t rm = w.rect16(s.x0, s.x1, s.y0, s.y1);
      const U4 ones = bits ? u4_zero() : u4_and1(w.expand16(rm), 0x01010101u);
      const bool same = s.nz == rm && (bits || (v[0] == ones[0] && v[1] == ones[1] && v[2] == ones[2] && v[3] == ones[3]));
      s.is_rect = !w.any(!same);
    }
  } else if (s.any_nz) {
    // _get_bbox (object.py:49-58): rows via first/last truthy flat index, columns via min/max reduction
    int cmin = 127, cmax = -1, fmin = 4096, fmax = -1;
    int r = w.r0, c = w.c0, k = 0;
    for (int sg = 0; sg < p.nseg; sg++) {
      if (k < 16) {
        int len = imin(p.W - c, 16 - k);
        uint32_t sub = (s.nz >> k) & ((1u << len) - 1u);
        if (sub) {
          int lo = __builtin_ctz(sub), hi = 31 - __builtin_clz(sub);
          cmin = imin(cmin, c + lo);
          cmax = imax(cmax, c + hi);
          fmin = imin(fmin, 16 * w.lane + k + lo);
          fmax = imax(fmax, 16 * w.lane + k + hi);
        }
        k += len;
        c = 0;
        r++;
      }
    }
    fmin = w.wave_min(fmin);
    fmax = w.wave_max(fmax);
    s.x0 = (int)(((uint32_t)fmin * p.div_magic) >> 16);
    s.x1 = (int)(((uint32_t)fmax * p.div_magic) >> 16);
    s.y0 = w.wave_min(cmin);
    s.y1 = w.wave_max(cmax);
  }
}

// ------------------------------------------------------------------------------------------------
// per-step scratch: the grid plane (loaded at most once) and the algorithmic byte count
// ------------------------------------------------------------------------------------------------
struct Scratch {
  U4 grid;
  bool have_grid;    // `grid` holds the current grid plane (loaded, requested early, or just produced by the op)
  bool grid_counted; // ACCT: the byte count already includes the grid read (or the op replaced the plane)
  bool loaded;       // the grid plane was fetched from memory by this step (need_grid)
  uint32_t sel_pending;  // what reset_sel / keep_sel still owe the `selected` plane after the op: 0 nothing (or the op wrote the plane
                         // itself: place), bit 1 = keep_sel's copy of the selection, 1 = reset_sel's zero-fill
  uint32_t bytes;  // algorithmic HBM bytes of this step (SURVEY.md §8d accounting; ACCT instantiations only)
};
#define ARCLE_ACCT(expr)      \
  do {                        \
    if (ACCT) s.bytes += (expr); \
  } while (0)

template <int ACCT>
ARCLE_DEV void need_grid(const Wave& w, Scratch& s) {
  if (!s.have_grid) {
    s.grid = w.load(ARCLE_PL_GRID);
    s.have_grid = true;
    s.loaded = true;
  }
  if (ACCT && !s.grid_counted) {
    s.bytes += w.p.P;  // the op semantically reads the grid
    s.grid_counted = true;
  }
}

// grid[:gh,:gw] == answer with equal dims (base.py:176-177, o2arcenv.py:124-127)
template <int ACCT>
ARCLE_DEV bool grid_equals_answer(const Wave& w, Scratch& s, const Rec& r) {
  if (((r.w[0] >> 16) ^ (r.w[3] >> 16)) != 0u) return false;  // grid_dim != answer_dim
  need_grid<ACCT>(w, s);
  const U4 a = w.load(ARCLE_PL_ANSWER);
  ARCLE_ACCT(w.p.P);
  const U4 m = w.expand16(w.rect16(0, r.gh() - 1, 0, r.gw() - 1));
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) diff |= (s.grid[i] ^ a[i]) & m[i];
  return !w.any(diff != 0);
}

// ------------------------------------------------------------------------------------------------
// object layer (object.py:60-165)
// ------------------------------------------------------------------------------------------------

// _apply_patch + _apply_sel (object.py:113-165).  `tile` holds the object plane staged in LDS with an
// extra flat offset `S0` (object cell f lives at tile[f + S0]); `osel` is the object_sel cell mask in
// the same shifted frame.  Writes grid and selected.
template <int ACCT>
ARCLE_DEV void place(const Wave& w, Scratch& s, const Rec& r, const U4& background, const uint32_t* tile, uint32_t osel,
                     int S0, bool osel_full) {
  const int W = w.p.W;
  s.grid = background;
  U4 selected = u4_zero();
  const int ox = r.ox(), oy = r.oy();
  const int xh = i8w(ox + r.oh()), yw = i8w(oy + r.ow());  // int8 + int8 (object.py:127)
  const int gh = r.gh(), gw = r.gw();
  if (xh > 0 && ox < gh && yw > 0 && oy < gw) {
    const int stx = imax(0, ox), edx = imin(gh, xh), sty = imax(0, oy), edy = imin(gw, yw);
    const uint32_t R = w.rect16(stx, edx - 1, sty, edy - 1);
    const int S = S0 - (ox * W + oy);
    const U4 po = w.shifted(tile, S);
    const U4 rb = w.expand16(R);
    s.grid = u4_sel(u4_and(rb, posbytes(po)), po, background);  // where=(p>0)  object.py:138
    // object.py:165; when object_sel covers the whole object tile (rectangle selection) the placed mask IS R
    selected = u4_and1(osel_full ? rb : w.expand16(w.shifted_bits(osel, S) & R), 0x01010101u);
  }
  w.store(ARCLE_PL_GRID, s.grid);
  w.store(ARCLE_PL_SELECTED, selected);
  s.have_grid = s.grid_counted = true;
  s.sel_pending = 0;
  ARCLE_ACCT(2 * w.p.P);
}

// ------------------------------------------------------------------------------------------------
// affine tile gather for 16 <= W: out cell k of this lane reads tile byte (k < k1 ? B0 : B1) + step*k
// (the lane's window is row r0 from column c0 for k < k1, then row r0+1 from column 0).  step = +1 / -1 are
// contiguous runs (one shifted window per segment, byte-reversed for -1); anything else is a strided gather.
// Cells whose source lies outside the tile get unspecified bytes: callers mask with the destination rectangle.
// ------------------------------------------------------------------------------------------------
ARCLE_DEV U4 window_fwd(const Wave& w, const uint32_t* tile, int first) {  // bytes [first, first+16)
  return w.shifted(tile, first - 16 * w.lane);
}
ARCLE_DEV U4 window_rev(const Wave& w, const uint32_t* tile, int last) {  // bytes last, last-1, ..., last-15
  U4 f = w.shifted(tile, last - 15 - 16 * w.lane), r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = __builtin_bswap32(f[3 - i]);
  return r;
}
ARCLE_DEV U4 gather_affine(const Wave& w, const uint32_t* tile, int B0, int B1, int step) {
  const U4 lmb = w.expand16(w.lm);  // bytes of the first row segment
  if (step == 1) return u4_sel(lmb, window_fwd(w, tile, B0), window_fwd(w, tile, B1));
  if (step == -1) return u4_sel(lmb, window_rev(w, tile, B0), window_rev(w, tile, B1));
  const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
  U4 o = u4_zero();
#pragma unroll
  for (int k = 0; k < 16; k++) {
    int a = ((k < w.k1) ? B0 : B1) + step * k;
    o[k >> 2] |= (uint32_t)t8[xl::lds_idx(a, 1024)] << (8 * (k & 3));
  }
  return o;
}

// Result of _init_objsel (object.py:60-111)
struct Lift {
  bool ok;         // false: inactive and nothing selected -> the op is a no-op
  bool fresh;      // a new selection was lifted
  uint32_t osel;   // object_sel cell mask, in the frame of `tile` (see S0)
  int S0;          // staged object cell f is at tile[f + S0]
  bool osel_full;  // object_sel == the whole h x w object tile (selection was a rectangle)
  U4 object, object_sel, background;
};

// After this call lds->a holds the object bytes (shifted by S0) and L.object/object_sel/background are current.  For
// a fresh selection the tile is the masked GRID (S0 = x0*W + y0), so that Move can place straight from it without a
// second staging.  `write_obj`: store object / object_sel now (Move: final; Rotate/Flip rewrite them themselves).
template <int ACCT>
ARCLE_DEV void init_objsel(const Wave& w, Scratch& s, Rec& r, const Sel& sel, Lift& L, bool write_obj) {
  const int W = w.p.W, P = w.p.P;
  if (sel.any_nz) {  // object.py:67-99
    need_grid<ACCT>(w, s);
    const int h = sel.x1 - sel.x0 + 1, wd = sel.y1 - sel.y0 + 1;
    const uint32_t pos = sel_pos(w, sel);
    const U4 pm = w.expand16(pos);
    // every read of the tile below is masked by a rectangle inside the selection's bbox image, so for a
    // rectangle selection (all cells of the bbox selected) the grid itself can be staged
    w.stage(w.lds->a, sel.is_rect ? s.grid : u4_and(s.grid, pm));
    const int S0 = sel.x0 * W + sel.y0;
    const uint32_t orect = w.rect16(0, h - 1, 0, wd - 1);
    const U4 ob = w.expand16(orect);
    L.object = u4_and(w.shifted(w.lds->a, S0), ob);
    L.object_sel = u4_and1(sel.is_rect ? ob : w.expand16(w.shifted_bits(pos, S0) & orect), 0x01010101u);
    L.background = u4_andn(s.grid, pm);
    r.put2(ARCLE_REC_OBJECT_DIM, h, wd);
    r.put2(ARCLE_REC_OBJECT_POS, sel.x0, sel.y0);
    r.put2(ARCLE_REC_ACTIVE, 1, 0);  // active = 1, rotation_parity = 0
    if (write_obj) {
      w.store(ARCLE_PL_OBJECT, L.object);
      w.store(ARCLE_PL_OBJECT_SEL, L.object_sel);
    }
    w.store(ARCLE_PL_BACKGROUND, L.background);
    ARCLE_ACCT(3 * P);
    L.ok = true;
    L.fresh = true;
    L.osel = pos;  // in the grid frame, consistent with the tile
    L.S0 = S0;
    L.osel_full = sel.is_rect;
    return;
  }
  L.fresh = false;
  L.S0 = 0;
  L.osel_full = false;
  L.osel = 0;
  if (r.active()) {  // object.py:102-107
    L.object = w.load(ARCLE_PL_OBJECT);
    L.object_sel = w.load(ARCLE_PL_OBJECT_SEL);
    L.background = w.load(ARCLE_PL_BACKGROUND);
    ARCLE_ACCT(3 * P);
    w.stage(w.lds->a, L.object);
    L.ok = true;
    L.osel = nz16(L.object_sel);
    return;
  }
  L.ok = false;
}

// dst[:nh,:nw] = T(src[:h,:w]), rest 0 (_pad_assign object.py:43-47) for both object and object_sel.
// src index = ai*i + bj*j + c0 (affine in the destination cell), read from the LDS tiles.  Generic widths.
ARCLE_DEV void tile_transform(const Wave& w, U4& object, U4& object_sel, int nh, int nw, int ai, int bj, int c0) {
  const int W = w.p.W;
  w.stage(w.lds->a, object);
  w.stage(w.lds->b, object_sel);
  const uint8_t* ta = reinterpret_cast<const uint8_t*>(w.lds->a);
  const uint8_t* tb = reinterpret_cast<const uint8_t*>(w.lds->b);
  U4 o = u4_zero(), os = u4_zero();
  int i = w.r0, j = w.c0;
  int src = ai * i + bj * j + c0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (i < nh && j < nw) {
      int sidx = imin(imax(src, 0), 1023);
      o[k >> 2] |= (uint32_t)ta[sidx] << (8 * (k & 3));
      os[k >> 2] |= (uint32_t)tb[sidx] << (8 * (k & 3));
    }
    j++;
    src += bj;
    if (j == W) {
      j = 0;
      i++;
      src += ai - bj * W;
    }
  }
  object = o;
  object_sel = os;
}

// ------------------------------------------------------------------------------------------------
// FloodFill (color.py:79-103, dfs :8-30) on a flat bit-board: lane j < 32 holds cells [32j, 32j+32)
// ------------------------------------------------------------------------------------------------
ARCLE_DEV uint32_t to32(const Wave& w, uint32_t m16) {  // 16-bit/lane -> 32-bit/lane (lanes 0..31)
  uint32_t lo = xl::shfl(m16, (2 * w.lane) & 63), hi = xl::shfl(m16, (2 * w.lane + 1) & 63);
  return w.lane < 32 ? (lo | (hi << 16)) : 0u;
}
ARCLE_DEV uint32_t to16(const Wave& w, uint32_t m32) {
  uint32_t v = xl::shfl(m32, w.lane >> 1);
  return (v >> (16 * (w.lane & 1))) & 0xffffu;
}
// flat shift towards higher indices by `n` bits (0 < n < 2048) of the 1024-bit board
ARCLE_DEV uint32_t board_shl(const Wave& w, uint32_t v, int n) {
  int a = n >> 5, b = n & 31;
  int l0 = w.lane - a, l1 = l0 - 1;
  uint32_t w0 = xl::shfl(v, l0 & 63), w1 = xl::shfl(v, l1 & 63);
  if (l0 < 0 || l0 > 31) w0 = 0;
  if (l1 < 0 || l1 > 31) w1 = 0;
  return b ? ((w0 << b) | (w1 >> (32 - b))) : w0;
}
ARCLE_DEV uint32_t board_shr(const Wave& w, uint32_t v, int n) {
  int a = n >> 5, b = n & 31;
  int l0 = w.lane + a, l1 = l0 + 1;
  uint32_t w0 = xl::shfl(v, l0 & 63), w1 = xl::shfl(v, l1 & 63);
  if (l0 < 0 || l0 > 31) w0 = 0;
  if (l1 < 0 || l1 > 31) w1 = 0;
  return b ? ((w0 >> b) | (w1 << (32 - b))) : w0;
}

// row board <-> per-lane 16-cell windows (16 <= W <= 32): row i = flat cells [W i, W i + W) = bits of up to three windows
ARCLE_DEV uint32_t rows_from16(const Wave& w, uint32_t m16, uint32_t Wb) {
  const uint32_t start = xl::mul24((uint32_t)w.lane, Wb), a = start >> 4, sh = start & 15u;
  uint32_t c0 = xl::shfl(m16, (int)(a & 63u)), c1 = xl::shfl(m16, (int)((a + 1u) & 63u)), c2 = xl::shfl(m16, (int)((a + 2u) & 63u));
  if (a > 63u) c0 = 0;
  if (a + 1u > 63u) c1 = 0;
  if (a + 2u > 63u) c2 = 0;
  const uint32_t lo = c0 | (c1 << 16);
  const uint32_t v = sh ? ((lo >> sh) | (c2 << (32u - sh))) : lo;
  return Wb == 32u ? v : (v & ((1u << Wb) - 1u));
}
ARCLE_DEV uint32_t rows_to16(const Wave& w, uint32_t rows, uint32_t Wb) {  // window = row r0 from column c0 (k1 cells), then row r0 + 1 ...
  if (Wb >= 16u) {  // at most two rows per window
    const uint32_t R0 = xl::shfl(rows, w.r0 & 63), R1 = xl::shfl(rows, (w.r0 + 1) & 63);
    const uint32_t lo = w.r0 > 63 ? 0u : (R0 >> (uint32_t)w.c0);
    const uint32_t hi = (w.r0 + 1 > 63 || w.k1 >= 16) ? 0u : (R1 << (uint32_t)w.k1);
    return (lo | hi) & 0xffffu;
  }
  uint32_t out = 0;  // narrow grids: a 16-cell window spans up to nseg rows
  int k = 0, c = w.c0;
  for (int sg = 0; sg < w.p.nseg; sg++) {
    const int row = w.r0 + sg;
    const uint32_t R = xl::shfl(rows, row & 63);
    if (k < 16 && row <= 63) out |= ((R >> (uint32_t)c) << (uint32_t)k);
    k += (int)Wb - c;
    c = 0;
  }
  return out & 0xffffu;
}

template <int ACCT>
ARCLE_DEV void op_floodfill(const Wave& w, Scratch& s, const Rec& r, const Sel& sel, int color) {
  const StepParams& p = w.p;
  int seed;
  if (sel.is_rect) {  // np.sum(sel) == 1  <=>  1x1 rectangle
    if (!(sel.any_nz && sel.x0 == sel.x1 && sel.y0 == sel.y1)) return;
    seed = sel.x0 * p.W + sel.y0;
  } else if (sel.one_cell >= 0) {
    // one truthy cell: np.sum(sel) is its value — the fill happens iff that value is 1 (color.py:91), seeded there
    seed = sel.one_cell;
    if (w.ingress != INGRESS_BITS) {  // (a boolean mask's one truthy cell IS 1)
      const uint32_t mine = u4_byte(sel.vals, seed & 15);
      if (xl::uniform(xl::shfl(mine, seed >> 4)) != 1u) return;
    }
  } else {
    if (!sel.any_nz) return;  // an all-zero mask sums to 0
    // several truthy cells, none of them negative: the sum is at least 2 (the usual multi-cell selection — no reduction needed)
    if (w.fw != FW_GENERIC && !w.any(sel.nz != sel_pos(w, sel))) return;
    int sum = 0, mx = -128;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      int v = (int)(int8_t)u4_byte(sel.vals, k);
      if ((w.valid16 >> k) & 1u) {
        sum += v;
        mx = imax(mx, v);
      }
    }
    if (w.wave_sum(sum) != 1) return;  // color.py:91
    int gmx = w.wave_max(mx);
    uint32_t em = eq16(sel.vals, (uint32_t)gmx) & w.valid16;  // np.argmax: first maximum (color.py:94)
    int f = em ? 16 * w.lane + __builtin_ctz(em) : 4096;
    seed = w.wave_min(f);
  }
  const int sx = (int)(((uint32_t)seed * p.div_magic) >> 16), sy = seed - sx * p.W;
  const int gh = r.gh(), gw = r.gw();
  if (sx >= gh || sy >= gw) return;  // color.py:96
  need_grid<ACCT>(w, s);
  // colour at the seed: lane seed>>4, byte seed&15
  uint32_t mine = u4_byte(s.grid, seed & 15);
  uint32_t col = xl::uniform(xl::shfl(mine, seed >> 4));
  const uint32_t inside = w.rect16(0, gh - 1, 0, gw - 1);
  const uint32_t M16 = eq16(s.grid, col) & inside;  // fillable cells of this lane's window: same colour as the seed, inside grid_dim
  uint32_t vis;
  if (w.fw != FW_GENERIC || (p.W <= 32 && p.H <= 64)) {
    // W <= 32 and H <= 64 (always so for 16 <= W): ROW BOARD — lane i holds row i as a W-bit word.  A vertical step is two DPP wave shifts; a
    // horizontal fill closes every run of a row in ONE pass with the carry trick: adding the filled bits to the fillable mask
    // ripples a carry up through each run of ones that contains one, (M ^ (M + F)) & M are the cells it passed, and the same on the
    // bit-reversed words fills downwards.  The two alternate until neither adds a cell: the number of passes follows the number of
    // corridor legs / row transitions of the region, not its cell-path length (a 1-wide spiral: ~60 legs against 434 cell steps).
    const uint32_t Wb = (uint32_t)p.W;
    const uint32_t M = rows_from16(w, M16, Wb), rM = xl::bfrev(M);
    // column-wise run masks for vertical steps of 2, 4 and 8 rows: bit j of Pkd says rows i-k+1..i are all fillable in column j
    // (Pku: rows i..i+k-1).  The k-row shifts are DPP row shifts — they stop at the 16-lane row boundary, which only makes those
    // steps conservative there; the 1-row wave shifts cross it.
    const uint32_t P2d = M & xl::lane_prev(M), P2u = M & xl::lane_next(M);
    const uint32_t P4d = P2d & xl::row_prev<2>(P2d), P4u = P2u & xl::row_next<2>(P2u);
    const uint32_t P8d = P4d & xl::row_prev<4>(P4d), P8u = P4u & xl::row_next<4>(P4u);
    uint32_t F = (w.lane == sx) ? (1u << sy) : 0u;
#ifndef ARCLE_FILL_UNROLL
#define ARCLE_FILL_UNROLL 2  // passes per convergence ballot: two save a compare + ballot + branch per pair of passes on the launch's longest waves and
                             // cost at most one idle pass at the end (c5 5.57 -> 5.40 us; 3: 5.46; the C3 mix unchanged — profiles/round5_experiments.txt §14)
#endif
    for (int it = 0; it < 2 * ARCLE_MAX_CELLS; it++) {  // one pass = vertical steps of 1, 2, 4, 8 rows (up to 15 rows of a run), then
      const uint32_t F0 = F;                            // the horizontal fill of every row; one convergence ballot per pass
#pragma unroll
      for (int u = 0; u < ARCLE_FILL_UNROLL; u++) {
        F |= (xl::lane_prev(F) | xl::lane_next(F)) & M;   // rows i-1 / i+1 (0 beyond the wave)
        F |= (xl::row_prev<2>(F) & P2d) | (xl::row_next<2>(F) & P2u);
        F |= (xl::row_prev<4>(F) & P4d) | (xl::row_next<4>(F) & P4u);
        F |= (xl::row_prev<8>(F) & P8d) | (xl::row_next<8>(F) & P8u);
        F |= (xl::lane_prev(F) | xl::lane_next(F)) & M;   // (again: lets a run cross the 16-lane row boundary in the same pass)
        const uint32_t rF = xl::bfrev(F);
        F |= ((M ^ (M + F)) & M) | xl::bfrev((rM ^ (rM + rF)) & rM);
      }
      if (!w.any(F != F0)) break;
    }
    vis = rows_to16(w, F, Wb);
  } else {
    uint32_t Mb = to32(w, M16);
    uint32_t notfirst = to32(w, w.rect16(0, p.H - 1, 1, p.W - 1));
    uint32_t notlast = to32(w, w.rect16(0, p.H - 1, 0, p.W - 2));
    uint32_t F = (w.lane == (seed >> 5)) ? (1u << (seed & 31)) : 0u;
    for (int it = 0; it < ARCLE_MAX_CELLS; it++) {
      uint32_t grow = (board_shl(w, F, 1) & notfirst) | (board_shr(w, F, 1) & notlast) | board_shl(w, F, p.W) |
                      board_shr(w, F, p.W);
      uint32_t Fn = F | (grow & Mb);
      bool changed = w.any(Fn != F);
      F = Fn;
      if (!changed) break;
    }
    vis = to16(w, F);
  }
  s.grid = u4_sel1(w.expand16(vis), ((uint32_t)color & 0xffu) * 0x01010101u, s.grid);
  w.store(ARCLE_PL_GRID, s.grid);
  ARCLE_ACCT(p.P);
}

// dst[:nh,:nw] = src[ai*i + bj*j + c0] (flat index into the staged plane), rest 0: rot90 / transposes of a whole plane
ARCLE_DEV U4 plane_transform(const Wave& w, const U4& v, int nh, int nw, int ai, int bj, int c0) {
  const int W = w.p.W;
  w.stage(w.lds->a, v);
  const uint8_t* ta = reinterpret_cast<const uint8_t*>(w.lds->a);
  U4 o = u4_zero();
  int i = w.r0, j = w.c0;
  int src = ai * i + bj * j + c0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (i < nh && j < nw) o[k >> 2] |= (uint32_t)ta[imin(imax(src, 0), 1023)] << (8 * (k & 3));
    j++;
    src += bj;
    if (j == W) {
      j = 0;
      i++;
      src += ai - bj * W;
    }
  }
  return o;
}

// ------------------------------------------------------------------------------------------------
// task draw + augmentation at reset (device side of Loader.pick / base.py:95-108 and of the research env's
// augmentation, agents/env.py:31-42), keyed by (seed, GLOBAL env id, episode) so that a trajectory does not depend on
// how the batch is sharded over GPUs
// ------------------------------------------------------------------------------------------------
ARCLE_HD uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct TaskDraw {
  int problem, sub;  // index into pair_off / pair_cnt, pair within the problem
  int rot_k;         // np.rot90 count 0..3
  uint64_t perm;     // colour permutation, nibble c = perm[c] (c < 10; nibbles 10..15 stay the identity)
};
#define ARCLE_PERM_IDENTITY 0xFEDCBA9876543210ull
// floor(r / 2^32 * n) for a 32-bit fraction r: the range reduction without a division (n < 2^32)
ARCLE_HD uint32_t mulhi32(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32); }
// Two splitmix64 outputs per draw: z0 -> problem (high word) and pair (low word) by multiply-shift range reduction; z1 -> the quarter
// turns (two low bits) and the colour permutation: its high word is a 32-bit fraction that the Fisher-Yates loop consumes digit by digit
// (j = floor(frac * (i + 1)), frac = the fractional part of frac * (i + 1): one mul-hi and one mul-lo per swap, no division anywhere).
// An auto-reset onto a device-drawn task runs this on the wave that resets the env: the first version (one splitmix64 and one modulo per
// swap, ~1000 scalar instructions) made those waves the tail of the research step's launch (profiles/round4_experiments.txt).
ARCLE_HD TaskDraw draw_task(uint64_t seed, uint64_t gid, uint32_t episode, int n_problems, const int32_t* pair_cnt,
                            uint32_t aug_flags) {
  const uint64_t G = 0x9E3779B97F4A7C15ull;
  TaskDraw d;
  const uint64_t z0 = mix64(seed + gid * G + (uint64_t)episode * 0xD1B54A32D192ED03ull);
  d.problem = (int)mulhi32((uint32_t)(z0 >> 32), (uint32_t)n_problems);
  d.sub = (int)mulhi32((uint32_t)z0, (uint32_t)pair_cnt[d.problem]);
  const uint64_t z1 = mix64(z0 + G);
  d.rot_k = (aug_flags & ARCLE_AUG_ROT90) ? (int)(z1 & 3u) : 0;
  d.perm = ARCLE_PERM_IDENTITY;
  if (aug_flags & ARCLE_AUG_PERMUTE) {  // Fisher-Yates over the ten colours
    uint32_t r = (uint32_t)(z1 >> 32);
#pragma unroll
    for (int i = 9; i > 0; i--) {
      const int j = (int)mulhi32(r, (uint32_t)(i + 1));
      r *= (uint32_t)(i + 1);
      const uint64_t a = (d.perm >> (4 * i)) & 15u, b = (d.perm >> (4 * j)) & 15u;
      d.perm = (d.perm & ~((15ull << (4 * i)) | (15ull << (4 * j)))) | (b << (4 * i)) | (a << (4 * j));
    }
  }
  return d;
}
// eight nibbles -> eight bytes (nibble k of x = byte k of the result)
ARCLE_HD uint64_t spread_nibbles(uint32_t x) {
  uint64_t v = x;
  v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
  v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
  v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
  return v;
}
// byte c < 10 -> perm[c]; other values unchanged.  The sixteen nibbles become a 16-byte table in four (uniform) dwords, and a dword of
// cells is looked up with two v_perm_b32 (table bytes 0-7 / 8-15 by the cell's low three bits) and a select on bit 3 — 9 vector
// instructions per four cells; a cell >= 16 (never in ARC data) keeps its value through the byte-wise path.
ARCLE_DEV U4 permute_colours(const U4& v, uint64_t perm) {
  const uint64_t tlo = spread_nibbles((uint32_t)perm), thi = spread_nibbles((uint32_t)(perm >> 32));
  U4 o;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t x = v[i], sel = x & 0x07070707u;
    const uint32_t lo = xl::perm_bytes((uint32_t)(tlo >> 32), (uint32_t)tlo, sel), hi = xl::perm_bytes((uint32_t)(thi >> 32), (uint32_t)thi, sel);
    uint32_t m = (x >> 3) & 0x01010101u;
    m = (m << 8) - m;  // 0xff where bit 3 of the cell is set
    uint32_t r = (hi & m) | (lo & ~m);
    if (x & 0xf0f0f0f0u) {  // (out-of-palette cells: unchanged)
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const uint32_t c = (x >> (8 * b)) & 0xffu;
        if (c >= 16u) r = (r & ~(0xffu << (8 * b))) | (c << (8 * b));
      }
    }
    o[i] = r;
  }
  return o;
}
// np.rot90(plane[:h,:w], k) zero-padded; updates (h, w)
// (16 <= W <= 32: the affine window gather the Rotate / Flip operations use — no per-cell bounds arithmetic)
ARCLE_DEV U4 plane_transform_fast(const Wave& w, const U4& v, int nh, int nw, int ai, int bj, int c0) {
  w.stage(w.lds->a, v);
  const int B0 = ai * w.r0 + bj * w.c0 + c0, B1 = ai * (w.r0 + 1) - bj * w.k1 + c0;
  return u4_and(gather_affine(w, w.lds->a, B0, B1, bj), w.expand16(w.rect16(0, nh - 1, 0, nw - 1)));
}
ARCLE_DEV U4 rot90_plane(const Wave& w, const U4& v, int& h, int& wd, int k) {
  const int W = w.p.W;
  if (w.fw != FW_GENERIC && k != 0) {
    int nh = h, nw = wd, ai, bj, c0;
    if (k == 1) { ai = -1; bj = W; c0 = wd - 1; nh = wd; nw = h; }
    else if (k == 2) { ai = -W; bj = -1; c0 = (h - 1) * W + wd - 1; }
    else { ai = 1; bj = -W; c0 = (h - 1) * W; nh = wd; nw = h; }
    const U4 o = plane_transform_fast(w, v, nh, nw, ai, bj, c0);
    h = nh;
    wd = nw;
    return o;
  }
  if (k == 1) {
    const U4 o = plane_transform(w, v, wd, h, -1, W, wd - 1);
    const int t = h; h = wd; wd = t;
    return o;
  }
  if (k == 2) return plane_transform(w, v, h, wd, -W, -1, (h - 1) * W + wd - 1);
  if (k == 3) {
    const U4 o = plane_transform(w, v, wd, h, 1, -W, (h - 1) * W);
    const int t = h; h = wd; wd = t;
    return o;
  }
  return v;
}

// Copies table entry `t` (augmented by perm / rot_k) into the env's input + answer planes and the record's dims.
// Returns false (nothing written) when a rot90 does not fit the H x W plane (non-square max_grid_size).
// `soften`: a quarter turn that does not fit is dropped (k &= 2) instead of failing — device-drawn augmentations never fail.
ARCLE_DEV bool load_task(const Wave& w, Rec& r, int t, int rot_k, uint64_t perm, U4& input_out, bool soften = false) {
  const StepParams& p = w.p;
  U4 in = u4_zero(), an = u4_zero();
  if (w.live) {
    in = *reinterpret_cast<const U4*>(p.tbl_in + (size_t)t * p.PS + 16 * w.lane);
    an = *reinterpret_cast<const U4*>(p.tbl_ans + (size_t)t * p.PS + 16 * w.lane);
  }
  // (the two dims of an entry are an aligned int8 pair inside a dword of the [n_tasks][2] arrays: one scalar load each)
  const uint32_t dsh = 8u * ((2u * (uint32_t)t) & 2u);
  const uint32_t din = xl::uload1(at(p.tbl_in_dim, (2u * (uint32_t)t) & ~3u)) >> dsh, dan = xl::uload1(at(p.tbl_ans_dim, (2u * (uint32_t)t) & ~3u)) >> dsh;
  int ih = (int)(din & 0xffu), iw = (int)((din >> 8) & 0xffu);
  int ah = (int)(dan & 0xffu), aw = (int)((dan >> 8) & 0xffu);
  if (perm != ARCLE_PERM_IDENTITY) {  // the reference permutes the un-padded grids: the padding stays 0
    in = u4_and(permute_colours(in, perm), w.expand16(w.rect16(0, ih - 1, 0, iw - 1)));
    an = u4_and(permute_colours(an, perm), w.expand16(w.rect16(0, ah - 1, 0, aw - 1)));
  }
  if (w.count) w.issued += 2u * (uint32_t)p.PS + 4u;
  if ((rot_k & 1) && (iw > p.H || ih > p.W || aw > p.H || ah > p.W)) {
    if (!soften) return false;
    rot_k &= 2;
  }
  if (rot_k) {
    in = rot90_plane(w, in, ih, iw, rot_k);
    an = rot90_plane(w, an, ah, aw, rot_k);
  }
  w.store(ARCLE_PL_INPUT, in);
  w.store(ARCLE_PL_ANSWER, an);
  input_out = in;
  r.w[0] = (r.w[0] & 0xffff0000u) | (uint32_t)ih | ((uint32_t)iw << 8);
  r.w[3] = (r.w[3] & 0x0000ffffu) | ((uint32_t)ah << 16) | ((uint32_t)aw << 24);
  return true;
}

// A new episode for `env` with a task drawn on the device: bumps the env's episode counter, records the table index.  Cannot fail:
// a drawn quarter turn that does not fit a non-square H x W plane is dropped (the draw's k becomes k & 2; arcle_amd/sampling.py
// documents the same rule), so episode / cur_task are committed for a task that really was loaded.
ARCLE_DEV bool load_sampled_task(const Wave& w, Rec& r, int env, U4& input_out) {
  const StepParams& p = w.p;
  const uint32_t ep = xl::uniform((uint32_t)p.episode[env]);
  const TaskDraw d = draw_task(p.seed, (uint64_t)(p.env_base + env), ep, p.n_problems, p.pair_cnt, p.aug_flags);
  const int t = p.pair_off[d.problem] + d.sub;
  const bool ok = load_task(w, r, t, d.rot_k, d.perm, input_out, true);
  xl::lanes_converged();
  if (ok && w.lane == 0) {
    p.episode[env] = (int32_t)(ep + 1u);
    if (p.cur_task) p.cur_task[env] = t;
  }
  return ok;
}

// ------------------------------------------------------------------------------------------------
// init_state (base.py:155-166 + o2arcenv.py:16-34 / arcenv.py:81-89), counters as in reset (base.py:73-79)
// ------------------------------------------------------------------------------------------------
// `input`: the input plane when the caller just wrote it (a fresh task), else it is loaded
// (by value + flag: a pointer to a caller's local would force that local into scratch memory)
ARCLE_DEV void init_state(const Wave& w, Rec& r, I2& cnt, bool have_input = false, U4 input = U4{0u, 0u, 0u, 0u}) {
  const StepParams& p = w.p;
  const U4 in = have_input ? input : w.load(ARCLE_PL_INPUT);
  w.store(ARCLE_PL_GRID, in);
  U4 z = u4_zero();
  if (p.plane[ARCLE_PL_SELECTED]) w.store(ARCLE_PL_SELECTED, z);
  if (p.plane[ARCLE_PL_CLIP]) w.store(ARCLE_PL_CLIP, z);
  if (p.plane[ARCLE_PL_OBJECT]) w.store(ARCLE_PL_OBJECT, z);
  if (p.plane[ARCLE_PL_OBJECT_SEL]) w.store(ARCLE_PL_OBJECT_SEL, z);
  if (p.plane[ARCLE_PL_BACKGROUND]) w.store(ARCLE_PL_BACKGROUND, z);
  const uint32_t idim = r.w[0] & 0xffffu;
  r.w[0] = idim | (idim << 16);                                          // grid_dim = input_dim
  r.w[1] = 0;                                                            // clip_dim, object_dim
  r.w[2] = ((uint32_t)p.max_trial & 0xffu) << 16;                        // object_pos = 0, trials_remain, terminated = 0
  r.w[3] &= 0xffff0000u;                                                 // active, rotation_parity = 0; answer_dim kept
  cnt.x = 0;
  cnt.y = 0;
}

ARCLE_DEV void store_rec(const StepParams& p, int env, int lane, const Rec& r) {
  if (lane == 0) {
    U4 v;
    v[0] = r.w[0];
    v[1] = r.w[1];
    v[2] = r.w[2];
    v[3] = r.w[3];
    *reinterpret_cast<U4*>(p.rec + (size_t)env * ARCLE_REC_BYTES) = v;
  }
}
ARCLE_DEV void store_cnt(const StepParams& p, int env, int lane, const I2& cnt) {
  if (lane == 0) *reinterpret_cast<I2*>(p.cnt + 2 * (size_t)env) = cnt;
}
ARCLE_DEV Rec load_rec(const StepParams& p, int env) {
  const U4 v = xl::uload4(p.rec + (size_t)env * ARCLE_REC_BYTES);
  Rec r;
  r.w[0] = v[0];
  r.w[1] = v[1];
  r.w[2] = v[2];
  r.w[3] = v[3];
  return r;
}
ARCLE_DEV I2 load_cnt(const StepParams& p, int env) {
  const U2 v = xl::uload2(p.cnt + 2 * (size_t)env);
  I2 c;
  c.x = (int32_t)v[0];
  c.y = (int32_t)v[1];
  return c;
}

// ------------------------------------------------------------------------------------------------
// one step() of one env:  O2ARCv2Env.step (o2arcenv.py:130-147) / ARCEnv.step / RawARCEnv.step
// ------------------------------------------------------------------------------------------------
// Descriptor of slot `op`: one scalar load from the device copy of the op table (35 dwords for O2ARCv2Env, resident in
// the scalar cache).  (Round 2 measured a branch-free register decode of the canonical table — ~35 scalar instructions, no
// memory access — at 7.09 us per launch against 6.80 us for the load: instruction issue, not latency, is what a wave of
// this kernel competes for.)
ARCLE_DEV uint32_t decode_op(const StepParams& p, int op) { return xl::uload1(at(p.d_ops, (uint32_t)op * 4u)); }

struct StepOut {
  int reward;      // 0/1
  bool term;       // bool(state['terminated'])
  uint32_t bytes;  // algorithmic HBM bytes of the step (0 for skipped steps)
  uint32_t status; // ARCLE_ST_* bits this env raised in this step (also OR-ed into the handle's sticky status word)
  bool grid_loaded; // the step loaded the grid plane (accounting of the streaming instantiation's speculative load)
  bool have_grid;   // `grid` holds the env's grid plane as the step left it (loaded, requested early, or just produced): the fused
  U4 grid;          // packed-row epilogue takes it from here instead of reading the plane back (dead in every other instantiation)
};
ARCLE_DEV void raise_status(const StepParams& p, StepOut& out, uint32_t bits) {
  xl::atomic_or(p.status, bits);
  out.status |= bits;
}
// ARCLE_STEP_DENSE for a step that did not execute an action (auto-reset, skipped step): the pair (0, 0) — "no dense term";
// the host layer turns it into reward 0 (Gymnasium next-step autoreset: the reset step's reward is 0)
ARCLE_DEV void dense_none(const Wave& w) {
  if (w.lane == 0) {
    w.p.dense[2 * (size_t)w.env] = 0;
    w.p.dense[2 * (size_t)w.env + 1] = 0;
  }
}
// The library keeps the dense pair of every env's CURRENT grid in a cache of its own (StepParams::dense_cache; (0, 0) = unknown):
// a step that leaves grid and grid_dim alone — failed flood fills, Copy, Submit, no-op moves: about a third of the O2ARC mix —
// re-uses it instead of reading the grid and the answer plane again.  Everything that changes a grid OUTSIDE an op invalidates the
// entry: the reset kernels, arcle_set_state_rows, auto-reset steps; host code that edits planes directly calls arcle_invalidate.
ARCLE_DEV void dense_forget(const Wave& w) {
  if (w.p.dense_cache && w.lane == 0) *reinterpret_cast<I2*>(w.p.dense_cache + 2 * (size_t)w.env) = I2{0, 0};
}

// Everything of step() between "record/op/payload are in registers" and "record/counters/outputs go back to
// memory": autoreset, op decode, the operation itself, reward.  Planes are read/written through w.load/w.store, so
// the same code serves the single-step kernel (HBM) and the rollout kernel (register-resident planes).
// FEAT: 1 = the instantiation also carries the rarely used step flags (ARCLE_STEP_FEATURE_FLAGS: device-side task
// re-sampling + augmentation, dense reward, continuation rule, reset_on_submit); the plain instantiations (FEAT = 0) keep
// them out of the hot kernel's code, registers and SGPR spills
template <int ING, int FW, int ACCT, int FEAT, int FL = -1>
ARCLE_DEV StepOut step_core(const Wave& w, Rec& r, I2& cnt0, const U4& payload, const int op, const bool early = false,
                            const U4 early_grid = U4{0u, 0u, 0u, 0u}) {
  const StepParams& p = w.p;
  const int P = p.P, W = p.W, lane = w.lane;
  StepOut out;
  out.reward = 0;
  out.bytes = 0;
  out.status = 0;
  out.grid_loaded = false;
  out.have_grid = false;
  out.grid = u4_zero();
  // FL >= 0: the launch's step flags are this compile-time constant (the launcher picks the instantiation for the common
  // combination), so the flag tests below fold away
  const uint32_t flags = FL >= 0 ? ((uint32_t)FL & 0xffffu) : p.flags;
  if (flags & (ARCLE_STEP_AUTORESET | (FEAT ? ARCLE_STEP_RESAMPLE : 0u))) {
    // next-step autoreset: an env whose episode ended (terminated, or — with ARCLE_STEP_TRUNCATE — out of steps) is
    // re-initialised instead of executing the action; ARCLE_STEP_RESAMPLE first draws a new task on the device
    const bool ended = r.term() != 0 || ((flags & ARCLE_STEP_TRUNCATE) && cnt0.x >= p.step_limit);
    if (ended) {
      bool ok = true;
      U4 in = u4_zero();
      const bool resample = FEAT && (flags & ARCLE_STEP_RESAMPLE);
      if (resample) ok = load_sampled_task(w, r, w.env, in);
      if (ok) init_state(w, r, cnt0, resample, in);
      else raise_status(p, out, ARCLE_ST_AUG_DOMAIN);
      if (FEAT && (flags & ARCLE_STEP_DENSE)) {
        dense_none(w);
        dense_forget(w);
      }
      out.term = 0;
      out.bytes = (uint32_t)((resample ? 9 : 7) * P + 2 * ARCLE_REC_BYTES);
      return out;
    }
  }
  // ARCLE_STEP_DENSE: this env's cached pair, requested now so that the scalar load runs under the op
  I2 dense_prev = I2{0, 0};
  bool dense_known = false;
  if (FEAT && (flags & ARCLE_STEP_DENSE) && p.dense_cache && !w.resident) {
    const U2 c = xl::uload2(p.dense_cache + 2 * (size_t)w.env);
    dense_prev.x = (int32_t)c[0];
    dense_prev.y = (int32_t)c[1];
    dense_known = c[1] != 0u;
  }
  // an index past the table reads slot n_ops, which is always empty (the table copy has one more slot than ARCLE_MAX_OPS)
  const uint32_t slot = (uint32_t)op < (uint32_t)p.n_ops ? (uint32_t)op : (uint32_t)p.n_ops;
  const uint32_t desc = decode_op(p, (int)slot);
  const bool bad_op = ARCLE_OP_KIND(desc) == ARCLE_OP_NONE;
  if (bad_op) {
    // reference: IndexError / TypeError before any mutation
    raise_status(p, out, ARCLE_ST_BAD_OP);
    if (FEAT && (flags & ARCLE_STEP_DENSE)) dense_none(w);
    out.term = r.term() != 0;
    return out;
  }
  const int kind = (int)ARCLE_OP_KIND(desc), arg = (int)ARCLE_OP_ARG(desc);
  const uint32_t oflags = ARCLE_OP_FLAGS(desc);
#if ARCLE_STOP_AT == 3  // (diagnostic builds, tools/gpu_stagepmc.sh: instruction counts of the step's stages)
  xl::sink_s(desc);
  return out;
#endif

  Scratch s;
  s.have_grid = false;
  s.grid_counted = false;
  s.loaded = false;
  if (early) {  // (streaming instantiation: the plane was requested beside the per-env scalars and is in registers by now)
    s.grid = early_grid;
    s.have_grid = true;
  }
  s.bytes = 2 * ARCLE_REC_BYTES + 24;  // record R/W + action in + reward/term out
  int submit_inc = 0;
  bool domain_error = false;
  int eq = -1;  // grid == answer, evaluated at most once (Submit and reward see the same state)

  Sel sel;
  ingest_scalar(w, sel, payload, out);
  ingest_cells(w, sel, payload, kind == ARCLE_OP_MOVE || kind == ARCLE_OP_ROTATE || kind == ARCLE_OP_FLIP || kind == ARCLE_OP_COPY || kind == ARCLE_OP_CROP_GRID,
               kind != ARCLE_OP_COLOR && kind != ARCLE_OP_FLOODFILL);
  if (ING == INGRESS_MASK) ARCLE_ACCT(P);
  if (ING == INGRESS_BITS) ARCLE_ACCT((P + 7) >> 3);
  if (FEAT && is_cells(ING) && (flags & ARCLE_STEP_CONTINUE_RULE) &&
      (kind == ARCLE_OP_MOVE || kind == ARCLE_OP_ROTATE || kind == ARCLE_OP_FLIP)) {
    // the O2ARC trace harness (tests/o2arc_check.py:169-170): an object op whose logged selection equals the env's
    // current `selected` plane continues the active object, i.e. is sent with an empty selection
    const U4 cur = w.load(ARCLE_PL_SELECTED);
    const U4 vm = w.expand16(w.valid16);
    uint32_t diff = 0;
    const U4 given = sel_values(w, sel);
#pragma unroll
    for (int i = 0; i < 4; i++) diff |= (cur[i] ^ given[i]) & vm[i];
    if (!w.any(diff != 0)) {
      sel.nz = sel.pos = 0;
      sel.vals = u4_zero();
      sel.any_nz = sel.any_pos = false;
    }
  }

#if ARCLE_STOP_AT == 4
  xl::sink_s(sel.x0 + sel.x1 + sel.y0 + sel.y1 + (int)sel.any_nz);
  xl::sink_v(sel_nz(w, sel));
  return out;
#endif
  const uint32_t w3_before = r.w[3];  // (only reset_sel below touches the record before an op can turn out to be out of its domain)
  // reset_sel / keep_sel (object.py:10-41) set `selected` BEFORE the wrapped op runs; an object op that places its
  // object overwrites it afterwards.  The plane is written once, after the op, with whichever value is final — and not at
  // all when the op turns out to be out of its domain (the step is skipped).
  // (mask arithmetic instead of a branch: a uniform bool lives in an SGPR pair and costs the scalar unit a compare, a select and
  // an and per use)
  static_assert(ARCLE_OPF_RESET_SEL == 1u && (ARCLE_REC_ACTIVE & 3) == 0, "bit 0 of the flags byte; `active` is byte 0 of its dword");
  const uint32_t rs_mask = (uint32_t)((int32_t)(desc << 15) >> 31) & 0xffu;  // 0xff for an op wrapped by reset_sel (object.py:20-25)
  // with ARCLE_STEP_ELIDE_SELECTED an env that enters the step inactive is known to hold an all-zero `selected`
  // plane already (see include/arcle_hip.h): the zero-fill would rewrite zeros with zeros
  const uint32_t zero_selected = (flags & ARCLE_STEP_ELIDE_SELECTED) ? (r.w[ARCLE_REC_ACTIVE >> 2] & rs_mask) : rs_mask;
  if (ACCT && rs_mask) ARCLE_ACCT(P);  // semantic accounting is unchanged
  r.w[ARCLE_REC_ACTIVE >> 2] &= ~rs_mask;
  if ((oflags & (ARCLE_OPF_KEEP_SEL | ARCLE_OPF_RESET_SEL)) == ARCLE_OPF_KEEP_SEL) ARCLE_ACCT(P);  // object.py:36-40
  static_assert(ARCLE_OPF_KEEP_SEL == 2u, "bit 1");
  s.sel_pending = (oflags & ARCLE_OPF_KEEP_SEL) | (zero_selected ? 1u : 0u);

  switch (kind) {  // transition(): self.operations[op](state, action)   o2arcenv.py:149-151
    case ARCLE_OP_COLOR: {  // color.py:70-74 — whole HxW plane, grid_dim ignored
      if (sel.any_nz) {
        need_grid<ACCT>(w, s);
        s.grid = u4_sel1(w.expand16(sel_nz(w, sel)), ((uint32_t)arg & 0xffu) * 0x01010101u, s.grid);
        w.store(ARCLE_PL_GRID, s.grid);
        ARCLE_ACCT(P);
      }
      break;
    }
    case ARCLE_OP_FLOODFILL:
      op_floodfill<ACCT>(w, s, r, sel, arg);
      break;
    case ARCLE_OP_MOVE: {  // object.py:230-240
      Lift L;
      init_objsel<ACCT>(w, s, r, sel, L, true);
      if (!L.ok) break;
      const int dx = (arg == 0) ? -1 : (arg == 1) ? 1 : 0;
      const int dy = (arg == 2) ? 1 : (arg == 3) ? -1 : 0;
      r.put2(ARCLE_REC_OBJECT_POS, r.ox() + dx, r.oy() + dy);  // :238, int8 wrap
      place<ACCT>(w, s, r, L.background, w.lds->a, L.osel, L.S0, L.osel_full);
      break;
    }
    case ARCLE_OP_ROTATE:
    case ARCLE_OP_FLIP: {  // object.py:177-213 / :265-276
      // pre-compute the geometry so that an out-of-domain transform (the reference raises) skips the step
      int h, wd, x, y, xmin, xmax, ymin, ymax;
      const bool fresh = sel.any_nz;
      if (fresh) {
        xmin = sel.x0; xmax = sel.x1; ymin = sel.y0; ymax = sel.y1;
        h = xmax - xmin + 1; wd = ymax - ymin + 1; x = xmin; y = ymin;
      } else if (r.active()) {
        h = r.oh(); wd = r.ow(); x = r.ox(); y = r.oy();
        xmin = x; xmax = i8w(i8w(x + h) - 1); ymin = y; ymax = i8w(i8w(y + wd) - 1);  // :102-107
      } else {
        break;  // :110-111 total no-op
      }
      int nx = x, ny = y, nh = h, nw = wd, npar = fresh ? 0 : r.parity();
      int ai, bj, c0;
      if (kind == ARCLE_OP_ROTATE) {
        const int k = arg;
        if (k & 1) {
          // exact evaluation of the float centre arithmetic (:187-206) on doubled integers
          int sx2 = fresh ? xmax + xmin : i8w(xmax + xmin);
          int sy2 = fresh ? ymax + ymin : i8w(ymax + ymin);
          if ((h & 1) == (wd & 1)) {
            nx = floordiv2(sx2 - sy2 + 2 * y);
            ny = floordiv2(sy2 - sx2 + 2 * x);
          } else {
            npar = (npar + k) % 2;
            int sig = (k + 2) % 4 - 2, mod = 1 - npar;
            nx = floordiv2(sx2 + imin(sig * (sy2 - 2 * ymin), sig * (sy2 - 2 * ymax)) + 2 * mod);
            ny = floordiv2(sy2 + imin(-sig * (sx2 - 2 * xmin), -sig * (sx2 - 2 * xmax)) + 2 * mod);
          }
          nh = wd;
          nw = h;
          if (wd > p.H || h > p.W || nx < -128 || nx > 127 || ny < -128 || ny > 127) domain_error = true;
        }
        if (k == 1) { ai = -1; bj = W; c0 = wd - 1; }                          // rot90:  new[i,j] = old[j, w-1-i]
        else if (k == 2) { ai = -W; bj = -1; c0 = (h - 1) * W + (wd - 1); }    // rot180: old[h-1-i, w-1-j]
        else { ai = 1; bj = -W; c0 = (h - 1) * W; }                            // rot270: old[h-1-j, i]
      } else {
        // object_dim is NOT updated by Flip, also for D0/D1 (:270-273); the tile written is T(tile)
        if (arg == 0) { ai = W; bj = -1; c0 = wd - 1; }                        // fliplr: old[i, w-1-j]
        else if (arg == 1) { ai = -W; bj = 1; c0 = (h - 1) * W; }              // flipud: old[h-1-i, j]
        else if (arg == 2) { ai = 1; bj = W; c0 = 0; nh = wd; nw = h; }        // D0 transpose: old[j, i]
        else { ai = -1; bj = -W; c0 = (h - 1) * W + (wd - 1); nh = wd; nw = h; }  // D1: old[h-1-j, w-1-i]
        if (arg >= 2 && (wd > p.H || h > p.W)) domain_error = true;
      }
      if (domain_error) break;
      const bool new_geom = kind == ARCLE_OP_ROTATE && (arg & 1);
      if (FW != FW_GENERIC) {
        // ---- lean path (16 <= W <= 32): the transformed tile is gathered straight from the source plane
        //      (the grid for a fresh selection, the stored object when continuing) ------------------------
        const bool rect_sel = fresh && sel.is_rect;
        int S0 = 0;
        U4 src_sel = u4_zero(), background;
        if (fresh) {  // _init_objsel, object.py:67-99, fused with the transform
          need_grid<ACCT>(w, s);
          const U4 pm = w.expand16(sel_pos(w, sel));
          background = u4_andn(s.grid, pm);
          w.store(ARCLE_PL_BACKGROUND, background);
          w.stage(w.lds->a, rect_sel ? s.grid : u4_and(s.grid, pm));
          if (!rect_sel) src_sel = u4_and1(pm, 0x01010101u);
          S0 = xmin * W + ymin;
          r.put2(ARCLE_REC_OBJECT_POS, xmin, ymin);
          r.put2(ARCLE_REC_OBJECT_DIM, h, wd);
          r.put2(ARCLE_REC_ACTIVE, 1, 0);
          ARCLE_ACCT(3 * P);
        } else {  // object.py:102-107
          const U4 so = w.load(ARCLE_PL_OBJECT);
          src_sel = w.load(ARCLE_PL_OBJECT_SEL);
          background = w.load(ARCLE_PL_BACKGROUND);
          ARCLE_ACCT(5 * P);
          w.stage(w.lds->a, so);
        }
        if (!rect_sel) w.stage(w.lds->b, src_sel);
        const int B0 = ai * w.r0 + bj * w.c0 + c0 + S0;
        const int B1 = ai * (w.r0 + 1) - bj * w.k1 + c0 + S0;
        const uint32_t orect = w.rect16(0, nh - 1, 0, nw - 1);
        const U4 ob = w.expand16(orect);
        const U4 object = u4_and(gather_affine(w, w.lds->a, B0, B1, bj), ob);
        const U4 object_sel = rect_sel ? u4_and1(ob, 0x01010101u) : u4_and(gather_affine(w, w.lds->b, B0, B1, bj), ob);
        w.store(ARCLE_PL_OBJECT, object);
        w.store(ARCLE_PL_OBJECT_SEL, object_sel);
        if (new_geom) {
          r.put2(ARCLE_REC_OBJECT_POS, nx, ny);
          r.put2(ARCLE_REC_OBJECT_DIM, nh, nw);
          r.put(ARCLE_REC_PARITY, npar);
        }
        w.stage(w.lds->a, object);
        // Flip D0/D1 leave object_dim = (h,w) while the tile is (w,h) (:270-273): only when the two agree is the
        // placed selection exactly the destination rectangle
        place<ACCT>(w, s, r, background, w.lds->a, rect_sel ? orect : nz16(object_sel), 0,
                    rect_sel && nh == r.oh() && nw == r.ow());
        break;
      }
      Lift L;
      init_objsel<ACCT>(w, s, r, sel, L, false);
      tile_transform(w, L.object, L.object_sel, nh, nw, ai, bj, c0);
      w.store(ARCLE_PL_OBJECT, L.object);
      w.store(ARCLE_PL_OBJECT_SEL, L.object_sel);
      if (new_geom) {
        r.put2(ARCLE_REC_OBJECT_POS, nx, ny);
        r.put2(ARCLE_REC_OBJECT_DIM, nh, nw);
        r.put(ARCLE_REC_PARITY, npar);
      }
      if (!L.fresh) ARCLE_ACCT(2 * P);  // object, object_sel written back (already counted when fresh)
      w.stage(w.lds->a, L.object);
      place<ACCT>(w, s, r, L.background, w.lds->a, nz16(L.object_sel), 0, false);
      break;
    }
    case ARCLE_OP_COPY: {  // object.py:291-312
      if (!sel_any_pos(w, sel)) break;
      const int ss_h = arg ? r.gh() : r.in_h(), ss_w = arg ? r.gw() : r.in_w();
      if (sel.x1 > ss_h || sel.y1 > ss_w) break;  // :301 (sic: > not >=)
      U4 src;
      if (arg) {
        need_grid<ACCT>(w, s);
        src = s.grid;
      } else {
        src = w.load(ARCLE_PL_INPUT);
        ARCLE_ACCT(P);
      }
      const int h = sel.x1 - sel.x0 + 1, wd = sel.y1 - sel.y0 + 1;
      // where=logical_and(src, sel): for a rectangle selection the destination rectangle already excludes the rest
      w.stage(w.lds->a, sel.is_rect ? src : u4_and(src, w.expand16(sel_nz(w, sel))));
      const U4 clip = u4_and(w.shifted(w.lds->a, sel.x0 * W + sel.y0), w.expand16(w.rect16(0, h - 1, 0, wd - 1)));
      w.store(ARCLE_PL_CLIP, clip);
      r.put2(ARCLE_REC_CLIP_DIM, h, wd);
      ARCLE_ACCT(P);
      break;
    }
    case ARCLE_OP_PASTE: {  // object.py:317-348
      if (!sel_any_pos(w, sel)) break;
      const int h = r.ch(), wd = r.cw();
      if (h == 0 || wd == 0) break;  // :334
      const int ex = imin(sel.x0 + h, p.H), ey = imin(sel.y0 + wd, p.W);  // :340-341 clipped to HxW, not grid_dim
      need_grid<ACCT>(w, s);
      const U4 clip = w.load(ARCLE_PL_CLIP);
      ARCLE_ACCT(P);
      w.stage(w.lds->a, clip);
      const U4 pc = w.shifted(w.lds->a, -(sel.x0 * W + sel.y0));
      uint32_t R = w.rect16(sel.x0, ex - 1, sel.y0, ey - 1);
      if (!arg) R &= pos16(pc);  // paste_blank=False: where=(patch>0)
      s.grid = u4_sel(w.expand16(R), pc, s.grid);
      w.store(ARCLE_PL_GRID, s.grid);
      ARCLE_ACCT(P);
      break;
    }
    case ARCLE_OP_COPY_FROM_INPUT: {  // critical.py:28-29
      s.grid = w.load(ARCLE_PL_INPUT);
      s.have_grid = s.grid_counted = true;
      w.store(ARCLE_PL_GRID, s.grid);
      r.w[0] = (r.w[0] & 0xffffu) | (r.w[0] << 16);  // grid_dim = input_dim
      ARCLE_ACCT(2 * P);
      break;
    }
    case ARCLE_OP_RESET_GRID: {  // critical.py:17
      s.grid = u4_zero();
      s.have_grid = s.grid_counted = true;
      w.store(ARCLE_PL_GRID, s.grid);
      ARCLE_ACCT(P);
      break;
    }
    case ARCLE_OP_RESIZE_GRID: {  // critical.py:39-46
      if (!sel.any_nz) break;
      s.grid = u4_zero();
      s.have_grid = s.grid_counted = true;
      w.store(ARCLE_PL_GRID, s.grid);
      r.put2(ARCLE_REC_GRID_DIM, sel.x1 - sel.x0 + 1, sel.y1 - sel.y0 + 1);
      ARCLE_ACCT(P);
      break;
    }
    case ARCLE_OP_CROP_GRID: {  // critical.py:56-66
      if (!sel.any_nz) break;
      need_grid<ACCT>(w, s);
      const int h = sel.x1 - sel.x0 + 1, wd = sel.y1 - sel.y0 + 1;
      w.stage(w.lds->a, sel.is_rect ? s.grid : u4_and(s.grid, w.expand16(sel_nz(w, sel))));
      s.grid = u4_and(w.shifted(w.lds->a, sel.x0 * W + sel.y0), w.expand16(w.rect16(0, h - 1, 0, wd - 1)));
      w.store(ARCLE_PL_GRID, s.grid);
      r.put2(ARCLE_REC_GRID_DIM, h, wd);
      ARCLE_ACCT(P);
      break;
    }
    case ARCLE_OP_RESIZE_TO_ANSWER: {  // arcenv.py:31-35
      need_grid<ACCT>(w, s);
      const int ah = r.ah(), aw = r.aw();
      r.put2(ARCLE_REC_GRID_DIM, ah, aw);
      s.grid = u4_and(s.grid, w.expand16(w.rect16(0, ah - 1, 0, aw - 1)));
      w.store(ARCLE_PL_GRID, s.grid);
      ARCLE_ACCT(P);
      break;
    }
    case ARCLE_OP_SUBMIT: {  // base.py:172-183 (reset_on_submit=False)
      int trials = r.trials();
      if (trials != 0) {
        trials = i8w(trials - 1);  // :174 int8 wrap
        r.put(ARCLE_REC_TRIALS, trials);
        submit_inc = 1;
        if (FEAT && (flags & ARCLE_STEP_RESET_ON_SUBMIT)) {
          // base.py:179-180: init_state() rebinds current_state inside submit — the decrement, the `terminated` of a
          // correct answer and the trials-exhausted check below all land on the discarded dict;
          // what the caller sees is the re-initialised state, and reward() is evaluated on it
          I2 keep = cnt0;
          init_state(w, r, keep);
          s.have_grid = s.grid_counted = false;
          break;
        }
        eq = grid_equals_answer<ACCT>(w, s, r) ? 1 : 0;
        if (eq) r.put(ARCLE_REC_TERMINATED, 1);
      }
      if (trials == 0) r.put(ARCLE_REC_TERMINATED, 1);
      break;
    }
    default:
      break;
  }

#if ARCLE_STOP_AT == 5
  return out;
#endif
  if (domain_error) {  // the reference raised inside the op: the step did not happen (nothing was written yet)
    raise_status(p, out, ARCLE_ST_ROTATE_DOMAIN);
    if (FEAT && (flags & ARCLE_STEP_DENSE)) dense_none(w);
    r.w[3] = w3_before;
    out.term = r.term() != 0;
    return out;
  }
  if (s.sel_pending) {
    if (s.sel_pending & ARCLE_OPF_KEEP_SEL) w.store(ARCLE_PL_SELECTED, sel_values(w, sel));
    else w.store(ARCLE_PL_SELECTED, u4_zero());
  }

  // reward(): only the LAST op of the table can be rewarded (o2arcenv.py:121-128)
  int reward = 0;
  if (op == p.n_ops - 1) {
    if (eq < 0) eq = grid_equals_answer<ACCT>(w, s, r) ? 1 : 0;
    reward = eq;
  }
  if (FEAT && (flags & ARCLE_STEP_DENSE)) {
    // the research env's dense reward (agents/env.py:44-58) as an exact integer pair (correct cells, total cells);
    // the host forms  sparse*100 - 1 + correct/total
    I2 pair = dense_prev;
    ARCLE_ACCT(16);  // cache entry in, pair out
    if (!dense_known || (w.stored & (1u << ARCLE_PL_GRID))) {  // the grid (or grid_dim: every op that changes it stores the plane) moved
      need_grid<ACCT>(w, s);
      const U4 a = w.load(ARCLE_PL_ANSWER);
      ARCLE_ACCT(P);
      const int gh = r.gh(), gw = r.gw(), ah = r.ah(), aw = r.aw();
      const int mh = imin(gh, ah), mw = imin(gw, aw);
      // cells of this lane's window that match inside the common rectangle
      const uint32_t same = (flags16(nzflags(s.grid[0] ^ a[0]), nzflags(s.grid[1] ^ a[1]), nzflags(s.grid[2] ^ a[2]), nzflags(s.grid[3] ^ a[3])) ^ 0xffffu) &
                            w.rect16(0, mh - 1, 0, mw - 1);
      pair.x = (int)xl::wave_add((uint32_t)__builtin_popcount(same));
      int total = mh * mw;
      if ((gh <= ah) == (gw <= aw)) total += ah * aw > gh * gw ? ah * aw - gh * gw : gh * gw - ah * aw;
      else total += (gh > ah ? gh - ah : ah - gh) * mw + (gw > aw ? gw - aw : aw - gw) * mh;
      pair.y = total;
      if (p.dense_cache && !w.resident && lane == 0) *reinterpret_cast<I2*>(p.dense_cache + 2 * (size_t)w.env) = pair;
    }
    if (lane == 0) *reinterpret_cast<I2*>(p.dense + 2 * (size_t)w.env) = pair;
  }
  cnt0.x += 1;  // o2arcenv.py:142
  cnt0.y += submit_inc;
  out.reward = reward;
  out.term = r.term() != 0;
  out.bytes = s.bytes;
  out.grid_loaded = s.loaded;
  out.have_grid = s.have_grid;
  out.grid = s.grid;
  return out;
}

ARCLE_DEV void flat_row(const Wave& w, const Rec& r, bool only_stored = false);  // (the observation writers, below)
struct StepOut;
ARCLE_DEV void flat_tail(const Wave& w, const StepOut& out, const I2& cnt, bool truncated);
ARCLE_HD int flat_obs_len(const StepParams& p, int filtered);
ARCLE_DEV void pack_row(const Wave& w, const Rec& r, uint32_t reward, uint32_t term, int8_t* out, int stride, bool have_grid = false,
                        U4 grid = U4{0u, 0u, 0u, 0u});
ARCLE_HD int packed_stride(int P);

// The per-env inputs of one step: record, op index, counters and the selection payload — four independent loads
// (scalar loads for everything but a full mask), issued together so that they share ONE latency window.
struct StepInputs {
  U4 rec;
  U2 cnt;
  uint32_t op;
  U4 payload;
};
// (the four array bases arrive as separate kernel arguments: the first dwords of the argument segment are preloaded into SGPRs
// when the wave is created, so these loads do not wait for a fetch of the argument block)
template <int ING>
ARCLE_DEV StepInputs load_inputs(const Wave& w, int env, const int8_t* rec, const int32_t* cnt, const int32_t* op, const void* sel) {
  StepInputs in;
  // 32-bit unsigned byte offsets: the scalar loads take them as an SGPR offset (no 64-bit address arithmetic per array)
  const uint32_t e = (uint32_t)env;
  in.rec = xl::uload4(at(rec, e * (uint32_t)ARCLE_REC_BYTES));
  if (ING != INGRESS_BBOX5 && ING != INGRESS_BBOX5_PF) in.op = xl::uload1(at(op, e * 4u));
  in.cnt = xl::uload2(at(cnt, e * 8u));
  if (ING == INGRESS_BBOX5 || ING == INGRESS_BBOX5_PF) {  // one 20-byte record per env: the four corners, then the operation (rows are only dword aligned)
    in.payload = xl::uload4(at(sel, e * 20u));
    in.op = xl::uload1(at(sel, e * 20u + 16u));
  } else if (ING == INGRESS_BBOX) {
    in.payload = xl::uload4(at(sel, e * 16u));
  } else if (ING == INGRESS_POINT) {
    const U2 b = xl::uload2(at(sel, e * 8u));
    in.payload = u4_zero();
    in.payload[0] = b[0];
    in.payload[1] = b[1];
  } else {
    in.payload = load_payload(w, env, 0, sel);
  }
  return in;
}
template <int ING>
ARCLE_DEV StepInputs load_inputs(const Wave& w, int env) {
  return load_inputs<ING>(w, env, w.p.rec, w.p.cnt, w.p.op, w.p.sel);
}

// One wave = one env of the launch.  (A grid-stride variant — a wave walking several envs with the next env's scalars
// prefetched — measured no faster on this access pattern, tools/membench.hip "E=2/4/8 seq", and its loop-invariant
// code motion costs SGPRs on the single-env path.)
template <int ING, int FW, int ACCT, int FEAT, int FL = -1>
ARCLE_DEV void wave_step(Wave& w, int env, StepInputs& in, uint64_t t_entry = 0, uint64_t t_lut = 0, const bool early = false,
                         const U4 early_grid = U4{0u, 0u, 0u, 0u}) {
  const StepParams& p = w.p;
  const int lane = w.lane;
  // (FL >= 0: the kernel wrote the compile-time flag set into its copy of the parameters, so p.flags folds as well)
  const uint32_t flags = FL >= 0 ? ((uint32_t)FL & 0xffffu) : p.flags;
  if (is_tuple(ING)) xl::arrived(in.rec, in.cnt, in.op, in.payload);  // one wait for all four scalar loads
  else xl::arrived3(in.rec, in.cnt, in.op);
#ifdef ARCLE_TRACE_WAVES  // diagnostic build: per-wave shader clocks into the acct buffer (as uint64[N][8])
  const uint64_t t_in = xl::clock();
#endif
#ifndef ARCLE_NO_TOV
  if (is_tuple(ING)) {  // the tuple's arithmetic (sort, clip, rectangle masks, shift distances) runs on the vector ALUs: xl::tov
#pragma unroll
    for (int i = 0; i < 4; i++) in.payload[i] = xl::tov(in.payload[i]);
  }
#endif
  Rec r;
  r.w[0] = in.rec[0];
  r.w[1] = in.rec[1];
  r.w[2] = in.rec[2];
  r.w[3] = in.rec[3];
  I2 cnt0;
  cnt0.x = (int32_t)in.cnt[0];
  cnt0.y = (int32_t)in.cnt[1];
  w.set_env(env);
  // (diagnostic builds, profiles/round3_experiments.txt: n dependent scalar adds / s_nop / vector adds in every wave — which pipe is short)
#ifdef ARCLE_EXP_S
  { uint32_t d = in.op; asm volatile(".rept %c1\n s_add_u32 %0, %0, 1\n .endr" : "+s"(d) : "n"(ARCLE_EXP_S) : "scc"); xl::sink_s(d); }
#endif
#ifdef ARCLE_EXP_N
  asm volatile(".rept %c0\n s_nop 0\n .endr" :: "n"(ARCLE_EXP_N));
#endif
#ifdef ARCLE_EXP_V
  { uint32_t d = (uint32_t)lane; asm volatile(".rept %c1\n v_add_u32 %0, %0, 1\n .endr" : "+v"(d) : "n"(ARCLE_EXP_V)); asm volatile("" :: "v"(d)); }
#endif
#if ARCLE_STOP_AT == 2
  xl::sink_s(r.w[0] + r.w[1] + r.w[2] + r.w[3] + (uint32_t)cnt0.x + (uint32_t)cnt0.y + in.op + in.payload[0] + in.payload[3]);
  return;
#endif
  StepOut out = step_core<ING, FW, ACCT, FEAT, FL>(w, r, cnt0, in.payload, (int)in.op, early, early_grid);
#ifdef ARCLE_TRACE_WAVES
  const uint64_t t_core = xl::clock();
#endif
  // ---- epilogue: record, counters and the step outputs ------------------------------------------------
  const bool truncated = (flags & ARCLE_STEP_TRUNCATE) && cnt0.x >= p.step_limit;
  xl::lanes_converged();  // (emulator: every lane has read the record / counters before lane 0 rewrites them)
  if (lane == 0) {
    // (the 16 B record is written back unconditionally: comparing it with what was loaded costs 13 scalar instructions per wave,
    // 6.72 vs 6.85 us per launch)
    U4 rv;
    rv[0] = r.w[0];
    rv[1] = r.w[1];
    rv[2] = r.w[2];
    rv[3] = r.w[3];
    const uint32_t e = (uint32_t)env;
    xl::store_at(p.rec, e * (uint32_t)ARCLE_REC_BYTES, rv);
    xl::store_at(p.cnt, e * 8u, cnt0);
    xl::store_at(p.reward, e * 4u, (int32_t)out.reward);
    xl::store_at(p.term, e, (uint8_t)out.term);
    if (flags & ARCLE_STEP_TRUNCATE) xl::store_at(p.trunc, e, (uint8_t)truncated);
  }
  if (ACCT) {  // what the wave moved besides planes: record + counters in and out, action in, outputs out
    const uint32_t act = ING == INGRESS_MASK ? (uint32_t)p.P : ING == INGRESS_BITS ? 2u * 64u : ING == INGRESS_POINT ? 12u : 20u;  // (records: 20)
    w.issued += 2u * ARCLE_REC_BYTES + 16u + act + 5u;
    if (p.spec_grid && !out.grid_loaded) w.issued += (uint32_t)p.PS;  // (the lean twin's speculative grid load of a step that never used it)
    if (flags & ARCLE_STEP_TRUNCATE) { w.issued += 1u; out.bytes += 1u; }
    if (FEAT && (flags & ARCLE_STEP_DENSE)) w.issued += 8u;
  }
  if (FEAT && (flags & ARCLE_STEP_FLAT_OBS)) {
    // fused observation writer: the flattened row of the state this step just produced (FlattenObservation, optionally after
    // FilterO2ARC), written by the same wave — no second launch, no re-read of
    // the record.  The planes are read back through the wave's own L1 path (program order, see xl::own_stores_visible).
    xl::own_stores_visible();
    const uint32_t issued_before = w.issued;
    flat_row(w, r, (flags & ARCLE_STEP_ROWS_INCREMENTAL) != 0);
    if (p.flat_tail) flat_tail(w, out, cnt0, truncated);
    if (ACCT) {
      out.bytes += 2u * (uint32_t)flat_obs_len(p, p.flat_filter) + ARCLE_REC_BYTES;  // planes + record read once, row written once
      // issued: the plane re-reads were counted by load_hbm; the row bytes written = what was read (+ scalars and padding)
      w.issued += (flags & ARCLE_STEP_ROWS_INCREMENTAL) ? (w.issued - issued_before) / (uint32_t)p.PS * (uint32_t)p.P + 16u : (uint32_t)p.flat_stride;
    }
  }
  // fused packed row for the multi-GPU gather (grid | grid_dim | reward | terminated): in the feature instantiations, and in
  // the lean ones whose compile-time flags ask for it
  if ((FEAT || FL >= 0) && (flags & ARCLE_STEP_PACK_OBS)) {
    xl::own_stores_visible();
    pack_row(w, r, (uint32_t)out.reward, (uint32_t)out.term, p.pack_out, packed_stride(p.P), out.have_grid, out.grid);
    if (ACCT) {
      out.bytes += (uint32_t)(p.P + packed_stride(p.P));
      w.issued += (uint32_t)packed_stride(p.P);
    }
  }
  if (lane == 0) {
#ifdef ARCLE_TRACE_WAVES
    if (ACCT) {
      uint64_t* tr = reinterpret_cast<uint64_t*>(p.acct) + 8 * (size_t)env;
      tr[0] = t_entry;
      tr[1] = t_lut;
      tr[2] = t_in;
      tr[3] = t_core;
      tr[4] = xl::clock();
    }
#else
    if (ACCT && p.acct) {  // [0, N): algorithmic bytes, [N, 2N): bytes of the accesses actually issued
      p.acct[env] += out.bytes;
      p.acct[(size_t)p.n_envs + env] += w.issued;
    }
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// n_steps consecutive step()s of one env in ONE launch (a rollout / trace replay: the caller already holds the
// whole action sequence, e.g. tests/o2arc_check.py:139-199 of the reference or a scripted policy).  The env's
// planes and record live in registers for the whole rollout: HBM sees the planes once in and (if changed) once
// out, plus the action in and 5 B of reward/terminated out per step.
//   sel: int32 [n_steps][n_envs][4|2] | int8 [n_steps][n_envs][P]   op: int32 [n_steps][n_envs]
//   reward: int32 [n_steps][n_envs]     term: uint8 [n_steps][n_envs]
// ------------------------------------------------------------------------------------------------
template <int ING, int FW, int FL = -1>
ARCLE_DEV void wave_rollout(const StepParams& p, WaveLDS* lds, const U2* lut, int env, int lane) {
  Wave w(p, lds, lut, lane, ING, FW, false);
  w.set_env(env);
#pragma unroll
  for (int pl = 0; pl < ARCLE_N_PLANES; pl++) w.cache[pl] = p.plane[pl] ? w.load_hbm(pl) : u4_zero();
  w.resident = true;
  Rec r = load_rec(p, env);
  I2 cnt = load_cnt(p, env);
  const size_t N = (size_t)p.n_envs;
  // The next action is fetched while the current one executes (through VGPRs: vector loads of a uniform address +
  // readfirstlane at use, so that the in-order vmcnt lets the op body run under the load).
  U4 next_payload = load_payload_v(w, env, 0);
  uint32_t next_op = (uint32_t)p.op[env];
  for (int t = 0; t < p.n_steps; t++) {
    U4 payload = next_payload;
    if (is_tuple(ING)) {
#pragma unroll
      for (int i = 0; i < 4; i++) payload[i] = xl::uniform(payload[i]);
    }
    const int op = (int)xl::uniform(next_op);
    if (t + 1 < p.n_steps) {
      next_payload = load_payload_v(w, env, (size_t)t + 1);
      next_op = (uint32_t)p.op[((size_t)t + 1) * N + env];
    }
    // (the feature flags a rollout accepts — continuation rule, reset_on_submit — belong to mask-ingress trace replay;
    //  FL >= 0: the flag set is a compile-time constant of this instantiation, as in the step kernel)
    const StepOut out = step_core<ING, FW, 0, is_cells(ING) ? 1 : 0, FL>(w, r, cnt, payload, op);
    if (lane == 0) {
      p.reward[(size_t)t * N + env] = out.reward;
      p.term[(size_t)t * N + env] = (uint8_t)out.term;
    }
    // ARCLE_STEP_PACK_OBS (round 5): the packed observation row of EVERY step — grid | grid_dim | reward | terminated, what a learner gathers —
    // out of the registers the state lives in: pack_out is [n_steps][n_envs][packed stride]; the Gym contract "an observation after every
    // step" without leaving the chip between the steps
    if ((FL >= 0 ? (uint32_t)FL : p.flags) & ARCLE_STEP_PACK_OBS) {
      const int stride = packed_stride(p.P);
      pack_row(w, r, (uint32_t)out.reward, (uint32_t)out.term, p.pack_out + (size_t)t * N * (size_t)stride, stride, true, w.load(ARCLE_PL_GRID));
    }
  }
#pragma unroll
  for (int pl = 0; pl < ARCLE_N_PLANES; pl++)
    if (w.dirty & (1u << pl)) w.store_hbm(pl, w.cache[pl]);
  xl::lanes_converged();
  store_rec(p, env, lane, r);
  store_cnt(p, env, lane, cnt);
}

// ------------------------------------------------------------------------------------------------
// reset kernels
// ------------------------------------------------------------------------------------------------
ARCLE_DEV void wave_reset(const StepParams& p, WaveLDS* lds, const U2* lut, int env, int lane) {
  if (p.rmask && !xl::uniform((uint32_t)p.rmask[env])) return;
  Wave w(p, lds, lut, lane, INGRESS_BBOX, FW_GENERIC, false);
  w.set_env(env);
  Rec r = load_rec(p, env);
  I2 cnt;
  init_state(w, r, cnt);
  dense_forget(w);
  xl::lanes_converged();  // (emulator) every lane has read the record before lane 0 rewrites it
  store_rec(p, env, lane, r);
  store_cnt(p, env, lane, cnt);
}

// reset() with a caller-chosen task (base.py:95-108) or — task_idx == NULL — a task drawn on the device: the (input,
// answer) pair comes from the device task table, optionally augmented (agents/env.py:31-42)
ARCLE_DEV void wave_reset_table(const StepParams& p, WaveLDS* lds, const U2* lut, int env, int lane) {
  if (p.rmask && !xl::uniform((uint32_t)p.rmask[env])) return;
  Wave w(p, lds, lut, lane, INGRESS_BBOX, FW_GENERIC, false);
  w.set_env(env);
  Rec r;
  r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
  U4 in = u4_zero();
  bool ok;
  if (p.task_idx) {
    const int t = (int)xl::uniform((uint32_t)p.task_idx[env]);
    if (t < 0 || t >= p.n_tasks) {
      xl::atomic_or(p.status, ARCLE_ST_BAD_TASK);
      return;
    }
    int k = 0;
    uint64_t perm = ARCLE_PERM_IDENTITY;
    if (p.aug_k) k = (int)xl::uniform((uint32_t)p.aug_k[env]) & 3;
    if (p.aug_perm) {
      perm = ARCLE_PERM_IDENTITY & ~0xFFFFFFFFFFull;  // (nibbles 10..15 stay the identity: the lookup table has sixteen entries)
      for (int c = 0; c < 10; c++) perm |= (uint64_t)(xl::uniform((uint32_t)p.aug_perm[16 * (size_t)env + c]) & 15u) << (4 * c);
    }
    ok = load_task(w, r, t, k, perm, in);
    if (ok && p.cur_task && lane == 0) p.cur_task[env] = t;
  } else {
    ok = load_sampled_task(w, r, env, in);
  }
  if (!ok) {
    xl::atomic_or(p.status, ARCLE_ST_AUG_DOMAIN);
    return;
  }
  I2 cnt;
  init_state(w, r, cnt, true, in);
  dense_forget(w);
  store_rec(p, env, lane, r);
  store_cnt(p, env, lane, cnt);
}

// ------------------------------------------------------------------------------------------------
// flattened observation row of one env — what the reference's policies consume: gymnasium FlattenObservation of the
// state dict (keys sorted, nested object_states in place; agents/models/GPTPolicy.py:17-35 `unflatten_vec`), or of its
// FilterO2ARC subset (agents/env.py:109-126; agents/train.py:61-68 stacks FlattenObservation on it):
//   full     clip clip_dim grid grid_dim input input_dim | active background object object_dim object_pos object_sel
//            rotation_parity | selected terminated trials_remain                       = 7*H*W + 14 bytes (6314 at 30x30)
//   filtered active clip clip_dim grid grid_dim object object_dim object_pos trials_remain = 3*H*W + 10 bytes (2710)
// The segments start at arbitrary byte offsets of the row (row base 16-byte aligned, stride a multiple of 16 >= the logical
// length).  A plane segment is written as whole aligned 16-byte chunks — the plane is staged in the wave's 1 KiB LDS tile and each
// lane reads its chunk at the segment's misalignment S with the same uniform flat shift the object ops use — plus at most 15 head
// and 15 tail bytes stored singly; scalars are byte stores.  No per-row LDS buffer: the writer runs at the step kernel's occupancy.
// ------------------------------------------------------------------------------------------------
struct FlatRow {
  int8_t* row;       // global memory, this env's row
  int off;           // bytes written so far
  bool only_stored;  // ARCLE_STEP_ROWS_INCREMENTAL: rewrite only the segments of planes this step stored
};
// whole 16-byte chunk `chunk` of the row.  A plain (write-back) store: the write-through form the plane stores use is slower here
// (transition rows 35.5 -> 39.7 us, fused FilterO2ARC rows 10.8 -> 13.9 us: the boundary bytes of a row go to lines its chunk stores
// also touch, and a write-through store drops the line from the L2 — profiles/round3_experiments.txt)
ARCLE_DEV void row_store16(const Wave&, const FlatRow& fr, int chunk, const U4& v) {
  *reinterpret_cast<U4*>(fr.row + 16 * (size_t)chunk) = v;
}
ARCLE_DEV void flat_plane(const Wave& w, FlatRow& fr, int pl) {
  if (!w.p.plane[pl]) return;
  if (fr.only_stored && !(w.stored & (1u << pl))) {  // incremental rows: the segment already holds this plane (unchanged this step)
    fr.off += w.p.P;
    return;
  }
  const int P = w.p.P, off = fr.off, lane = w.lane;
  if (w.resident && w.row_src && !(w.stored & (1u << pl)) && (reinterpret_cast<uintptr_t>(w.row_src) & 15) == 0) {
    // state-row kernels, a plane the op did not change, input row 16-byte aligned: input and output rows share one layout, so the
    // segment is passed through chunk for chunk (aligned 16-byte loads and stores, no staging, no shift) plus its edge bytes
    const int8_t* in = w.row_src;
    const int c0 = (off + 15) >> 4, n_full = ((off + P) >> 4) - c0;
    if (lane < n_full) {
      const U4 c = *reinterpret_cast<const U4*>(in + 16 * (size_t)(c0 + lane));
      row_store16(w, fr, c0 + lane, c);
    }
    const int head = imin(16 * c0 - off, P);
    if (lane < head) fr.row[off + lane] = in[off + lane];
    const int tail = n_full >= 0 ? ((off + P) & 15) : 0;
    if (lane < tail) fr.row[16 * (c0 + n_full) + lane] = in[16 * (c0 + n_full) + lane];
    fr.off += P;
    return;
  }
  w.stage(w.lds->a, w.load(pl));  // (the state-row kernels keep the planes in registers: `load` serves them from there)
  const int c0 = (off + 15) >> 4, S = 16 * c0 - off;  // first whole chunk of the row inside the segment; its plane byte offset
  const int n_full = ((off + P) >> 4) - c0;           // whole chunks (<= 64); negative: the segment ends inside its first chunk
  const U4 o = w.shifted(w.lds->a, S);
  if (lane < n_full) row_store16(w, fr, c0 + lane, o);
  const uint8_t* t8 = reinterpret_cast<const uint8_t*>(w.lds->a);
  const int head = imin(S, P);                        // row bytes [off, 16 c0) = plane bytes [0, S)
  if (lane < head) fr.row[off + lane] = (int8_t)t8[xl::lds_idx(lane, 1024)];
  const int tail = n_full >= 0 ? ((off + P) & 15) : 0;  // row bytes [16 (c0 + n_full), off + P) = plane bytes [16 n_full + S, P)
  if (lane < tail) fr.row[16 * (c0 + n_full) + lane] = (int8_t)t8[xl::lds_idx(16 * n_full + S + lane, 1024)];
  fr.off += P;
}
ARCLE_DEV void flat_scalar(const Wave& w, FlatRow& fr, const Rec& r, int field, int n) {
  // (n <= 2 bytes of one record dword — the 2-byte fields start at even offsets: a constant register index and a per-lane shift,
  // not a per-lane index into the record, which would put the record into scratch memory)
  const uint32_t word = r.w[field >> 2];
  if (w.lane < n) fr.row[fr.off + w.lane] = (int8_t)((word >> (8 * ((field & 3) + (w.lane < n ? w.lane : 0)))) & 0xffu);
  fr.off += n;
}
ARCLE_HD int flat_obs_len(const StepParams& p, int filtered) {
  const bool o2 = p.plane[ARCLE_PL_SELECTED] != nullptr, clip = p.plane[ARCLE_PL_CLIP] != nullptr;
  if (filtered) return 3 * p.P + 10;
  return 2 * p.P + 6 + (clip ? p.P + 2 : 0) + (o2 ? 4 * p.P + 6 : 0);
}
// the row of w's env from the planes in memory and the record `r` (the step kernel calls it with the record it just produced)
ARCLE_DEV void flat_row(const Wave& w, const Rec& r, bool only_stored) {
  const StepParams& p = w.p;
  const int lane = w.lane, env = w.env;
  const bool o2 = p.plane[ARCLE_PL_SELECTED] != nullptr, clip = p.plane[ARCLE_PL_CLIP] != nullptr;
  FlatRow fr;
  fr.row = p.flat_out + (size_t)env * p.flat_stride;
  fr.off = 0;
  fr.only_stored = only_stored;
  if (p.flat_filter) {
    flat_scalar(w, fr, r, ARCLE_REC_ACTIVE, 1);
    flat_plane(w, fr, ARCLE_PL_CLIP);
    flat_scalar(w, fr, r, ARCLE_REC_CLIP_DIM, 2);
    flat_plane(w, fr, ARCLE_PL_GRID);
    flat_scalar(w, fr, r, ARCLE_REC_GRID_DIM, 2);
    flat_plane(w, fr, ARCLE_PL_OBJECT);
    flat_scalar(w, fr, r, ARCLE_REC_OBJECT_DIM, 2);
    flat_scalar(w, fr, r, ARCLE_REC_OBJECT_POS, 2);
    flat_scalar(w, fr, r, ARCLE_REC_TRIALS, 1);
  } else {
    flat_plane(w, fr, ARCLE_PL_CLIP);
    if (clip) flat_scalar(w, fr, r, ARCLE_REC_CLIP_DIM, 2);
    flat_plane(w, fr, ARCLE_PL_GRID);
    flat_scalar(w, fr, r, ARCLE_REC_GRID_DIM, 2);
    flat_plane(w, fr, ARCLE_PL_INPUT);
    flat_scalar(w, fr, r, ARCLE_REC_INPUT_DIM, 2);
    if (o2) {
      flat_scalar(w, fr, r, ARCLE_REC_ACTIVE, 1);
      flat_plane(w, fr, ARCLE_PL_BACKGROUND);
      flat_plane(w, fr, ARCLE_PL_OBJECT);
      flat_scalar(w, fr, r, ARCLE_REC_OBJECT_DIM, 2);
      flat_scalar(w, fr, r, ARCLE_REC_OBJECT_POS, 2);
      flat_plane(w, fr, ARCLE_PL_OBJECT_SEL);
      flat_scalar(w, fr, r, ARCLE_REC_PARITY, 1);
      flat_plane(w, fr, ARCLE_PL_SELECTED);
    }
    flat_scalar(w, fr, r, ARCLE_REC_TERMINATED, 1);
    flat_scalar(w, fr, r, ARCLE_REC_TRIALS, 1);
  }
  // row padding up to the stride (or up to the step-output tail)
  if (lane < 16 && fr.off + lane < p.flat_stride - (p.flat_tail ? 16 : 0)) fr.row[fr.off + lane] = 0;
}

// Optional tail of a flat row (arcle_set_flat_output_ex, tail = 1): the last 16 bytes of the row's stride carry the step outputs, so
// that ONE copy of the row brings everything a caller of step() needs —
//   int32 reward | int32 action_steps | int32 submit_count | uint8 terminated | uint8 truncated | uint8 status (ARCLE_ST_* raised by
//   THIS env in THIS step) | 0
ARCLE_DEV void flat_tail(const Wave& w, const StepOut& out, const I2& cnt, bool truncated) {
  const StepParams& p = w.p;
  if (w.lane == 0) {
    U4 t;
    t[0] = (uint32_t)out.reward;
    t[1] = (uint32_t)cnt.x;
    t[2] = (uint32_t)cnt.y;
    t[3] = (uint32_t)out.term | ((uint32_t)truncated << 8) | ((out.status & 0xffu) << 16);
    int8_t* const dst = p.flat_out + (size_t)w.env * p.flat_stride + (p.flat_stride - 16);
    if (p.flat_seq) {
      // completion signal for a host that polls the row's tail in pinned memory (the single-env classes): every store of this wave —
      // the whole row, the first three tail words — is made visible at system scope BEFORE the last word, which carries the caller's
      // sequence number in its top byte
      reinterpret_cast<uint32_t*>(dst)[0] = t[0];
      reinterpret_cast<uint32_t*>(dst)[1] = t[1];
      reinterpret_cast<uint32_t*>(dst)[2] = t[2];
      xl::release_store_system(reinterpret_cast<uint32_t*>(dst) + 3, t[3] | ((uint32_t)p.flat_seq << 24));
    } else {
      *reinterpret_cast<U4*>(dst) = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// state rows IN: the inverse of flat_row (full layout) — a flattened state row becomes planes + record again.  This is what makes
// transition(state, action) (o2arcenv.py:149-151; README.md:55 `env.transition(deepcopy(state), action)`) a stateless, batched
// device operation, and what a checkpoint restores from.
// ------------------------------------------------------------------------------------------------
// this lane's 16 cells of the plane segment that starts at byte `off` of the row (any alignment); bytes >= P are zero
ARCLE_DEV U4 row_plane(const Wave& w, const int8_t* row, int off) {
  const int P = w.p.P, f0 = 16 * w.lane;
  U4 v = u4_zero();
  if (f0 + 16 <= P) {
    v = xl::load16u(row + off + f0);
  } else if (f0 < P) {  // the lane holding the segment's tail: byte by byte, never past the segment (the row may end right behind it)
#pragma unroll
    for (int k = 0; k < 16; k++)
      if (f0 + k < P) v[k >> 2] |= (uint32_t)(uint8_t)row[off + f0 + k] << (8 * (k & 3));
  }
  if (w.count) w.issued += (uint32_t)P;
  return v;
}
ARCLE_DEV uint32_t row_byte(const int8_t* row, int off) { return xl::uniform((uint32_t)(uint8_t)row[off]); }
// byte offset of plane `pl`'s segment in a full state row (the layout flat_row writes); pl is a constant at every call site
ARCLE_DEV int row_offset(const StepParams& p, int pl) {
  const int P = p.P;
  const int c = p.plane[ARCLE_PL_CLIP] ? P + 2 : 0;  // clip, clip_dim in front of grid
  switch (pl) {
    case ARCLE_PL_CLIP: return 0;
    case ARCLE_PL_GRID: return c;
    case ARCLE_PL_INPUT: return c + P + 2;
    case ARCLE_PL_BACKGROUND: return c + 2 * P + 4 + 1;
    case ARCLE_PL_OBJECT: return c + 3 * P + 5;
    case ARCLE_PL_OBJECT_SEL: return c + 4 * P + 9;
    default: return c + 5 * P + 10;  // ARCLE_PL_SELECTED
  }
}
ARCLE_DEV U4 Wave::load_from_row(int pl) const {
  if (pl == ARCLE_PL_ANSWER) return xl::load16(p.plane[ARCLE_PL_ANSWER], (uint32_t)answer_env * (uint32_t)p.PS + 16u * (uint32_t)lane);
  if (!p.plane[pl]) return u4_zero();
  return row_plane(*this, row_src, row_offset(p, pl));
}
// `sink(plane id, bytes)` receives every plane of the row (planes = false: the scalars only); the record's state fields are filled
// in (answer_dim untouched)
template <typename Sink>
ARCLE_DEV void read_state_row(const Wave& w, const int8_t* row, Rec& r, Sink&& sink, bool planes = true) {
  const StepParams& p = w.p;
  const int P = p.P;
  const bool o2 = p.plane[ARCLE_PL_SELECTED] != nullptr, clip = p.plane[ARCLE_PL_CLIP] != nullptr;
  int off = 0;
  auto plane = [&](int pl) __attribute__((always_inline)) {
    if (planes) sink(pl, row_plane(w, row, off));
    off += P;
  };
  auto scalar = [&](int field, int n) __attribute__((always_inline)) {
    r.put(field, (int)row_byte(row, off));
    if (n == 2) r.put(field + 1, (int)row_byte(row, off + 1));
    off += n;
  };
  if (clip) {
    plane(ARCLE_PL_CLIP);
    scalar(ARCLE_REC_CLIP_DIM, 2);
  }
  plane(ARCLE_PL_GRID);
  scalar(ARCLE_REC_GRID_DIM, 2);
  plane(ARCLE_PL_INPUT);
  scalar(ARCLE_REC_INPUT_DIM, 2);
  if (o2) {
    scalar(ARCLE_REC_ACTIVE, 1);
    plane(ARCLE_PL_BACKGROUND);
    plane(ARCLE_PL_OBJECT);
    scalar(ARCLE_REC_OBJECT_DIM, 2);
    scalar(ARCLE_REC_OBJECT_POS, 2);
    plane(ARCLE_PL_OBJECT_SEL);
    scalar(ARCLE_REC_PARITY, 1);
    plane(ARCLE_PL_SELECTED);
  }
  scalar(ARCLE_REC_TERMINATED, 1);
  scalar(ARCLE_REC_TRIALS, 1);
}

// arcle_set_state_rows: row `env` of p.rows_in -> the resident state of env `env` (planes + record; the task's answer / answer_dim
// and the counters are not part of a state row and stay)
ARCLE_DEV void wave_set_state_row(const StepParams& p, WaveLDS* lds, const U2* lut, int env, int lane) {
  if (p.rmask && !xl::uniform((uint32_t)p.rmask[env])) return;
  Wave w(p, lds, lut, lane, INGRESS_BBOX, FW_GENERIC, false);
  w.set_env(env);
  Rec r = load_rec(p, env);
  read_state_row(w, p.rows_in + (size_t)env * p.rows_in_stride, r, [&](int pl, const U4& v) { w.store_hbm(pl, v); });
  dense_forget(w);
  xl::lanes_converged();
  store_rec(p, env, lane, r);
}

// arcle_transition_rows: ONE operation applied to the state held in row `row` of p.rows_in, result written as row `row` of
// p.flat_out — transition(state, action) of the reference for a whole batch of (state, action) pairs, none of which touches the
// handle's resident envs.  The task-side data a transition can read (answer, answer_dim: Submit and the reward) come from resident
// env p.task_idx[row] (NULL: env `row`).  reward / terminated as step(); the row's optional tail (p.flat_tail) carries them as
// well, with action_steps = 1 and submit_count = 1 iff the op was a Submit that counted (base.py:174-175).
template <int ING, int FW>
ARCLE_DEV void wave_transition_row(const StepParams& p, WaveLDS* lds, const U2* lut, int row, int lane) {
  Wave w(p, lds, lut, lane, ING, FW, false);
  int src = row;
  if (p.task_idx) src = (int)xl::uniform((uint32_t)p.task_idx[row]);
  StepOut out;
  out.reward = 0;
  out.term = false;
  out.bytes = 0;
  out.status = 0;
  out.grid_loaded = false;
  out.have_grid = false;
  I2 cnt;
  cnt.x = cnt.y = 0;
  const int8_t* rin = p.rows_in + (size_t)row * p.rows_in_stride;
  if (src < 0 || src >= p.n_resident) {  // no such env to take the answer from: the row is passed through untouched
    raise_status(p, out, ARCLE_ST_BAD_TASK);
    src = 0;
  }
  w.set_env(src);
  Rec r = load_rec(p, src);  // (answer_dim; every state field is overwritten from the row)
  read_state_row(w, rin, r, [&](int, const U4&) {}, false);  // the scalars now; planes on demand (Wave::load)
  w.resident = true;
  w.row_src = rin;
  w.answer_env = src;
  w.have = 0;
  w.env = row;  // outputs (dense pair, flat row) are indexed by the row
  const U4 pay = load_payload(w, row, 0, p.sel);
  const int op = (int)xl::uniform((uint32_t)p.op[row]);
  if (!out.status) out = step_core<ING, FW, 0, 1>(w, r, cnt, pay, op);
  xl::lanes_converged();
  if (lane == 0) {
    p.reward[row] = out.reward;
    p.term[row] = (uint8_t)out.term;
  }
  // planes the op did not touch are passed through from the input row — or, in place (the launcher sets the incremental flag when
  // rows_out is rows_in), left where they are
  flat_row(w, r, (p.flags & ARCLE_STEP_ROWS_INCREMENTAL) != 0);
  if (p.flat_tail) flat_tail(w, out, cnt, false);
}

ARCLE_DEV void wave_flatten(const StepParams& p, WaveLDS* lds, const U2* lut, int env, int lane) {
  Wave w(p, lds, lut, lane, INGRESS_BBOX, FW_GENERIC, false);
  w.set_env(env);
  const Rec r = load_rec(p, env);
  flat_row(w, r);
}

// ------------------------------------------------------------------------------------------------
// packed minimal observation of one env, what a central learner gathers per step:
//   row = grid (H*W bytes) | grid_dim (2) | reward int32 LE (4) | terminated (1) | zero padding to a multiple of 16
// one aligned 16 B store per lane; p.flat_out / p.flat_stride name the destination, p.reward / p.term the step outputs
// ------------------------------------------------------------------------------------------------
ARCLE_HD int packed_stride(int P) { return (P + 7 + 15) & ~15; }
// `reward` / `term`: the step outputs of this env (the fused epilogue passes what it just computed)
// `have_grid` / `grid`: the plane as the step left it in registers (the fused epilogue: no read-back, no extra round trip at the end of the wave)
ARCLE_DEV void pack_row(const Wave& w, const Rec& r, uint32_t reward, uint32_t term, int8_t* out, int stride, bool have_grid, U4 grid) {
  const int P = w.p.P, lane = w.lane;
  if (16 * lane >= stride) return;
  U4 v = have_grid ? grid : w.load(ARCLE_PL_GRID);  // (bytes >= P of the plane row are zero padding)
  if (16 * lane + 16 > P) {          // this lane's window holds the metadata bytes
    // the 7 metadata bytes as one little-endian word: grid_dim (2), reward int32 (4), terminated (1)
    const uint64_t meta = (uint64_t)(uint32_t)r.gh() | ((uint64_t)(uint32_t)r.gw() << 8) | ((uint64_t)reward << 16) | ((uint64_t)(term & 0xffu) << 48);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int b = 16 * lane + k - P;
      if (b >= 0) {
        const uint32_t byte = b < 7 ? (uint32_t)(meta >> (8 * b)) & 0xffu : 0u;
        v[k >> 2] = (v[k >> 2] & ~(0xffu << (8 * (k & 3)))) | (byte << (8 * (k & 3)));
      }
    }
  }
  *reinterpret_cast<U4*>(out + (size_t)w.env * stride + 16 * lane) = v;
}
ARCLE_DEV void wave_pack_obs(const StepParams& p, WaveLDS* lds, const U2* lut, int env, int lane) {
  Wave w(p, lds, lut, lane, INGRESS_BBOX, FW_GENERIC, false);
  w.set_env(env);
  const Rec r = load_rec(p, env);
  pack_row(w, r, (uint32_t)p.reward[env], (uint32_t)p.term[env], p.flat_out, p.flat_stride);
}

}  // namespace arcle
