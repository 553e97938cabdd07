// arcle_big.h — the step / reset / row kernels' bodies for grids BEYOND one wavefront: H * W > ARCLE_MAX_CELLS (1024 cells),
// H, W <= 127 (grid dims are int8 in the reference's state dict, base.py:162-166).  The reference takes any max_grid_size
// (/root/reference/arcle/envs/base.py:37-49); ARC itself never exceeds 30 x 30, so this is the completeness path, not the headline:
// the 30 x 30 batch keeps its one-wavefront-per-env kernels (arcle_wave.h).
//
// Execution model — ONE WORKGROUP per env (64 … 512 threads: two 16-byte chunks of a plane per thread in the compile-time-specialised LEAN
// launches — one wavefront per env at 40 x 40 —, one chunk per thread in the generic kernel):
//   * a plane is PS = H*W rounded up to 128 bytes; thread t owns the 16-byte chunks t, t + NT, t + 2 NT ... of it: every global plane
//     access is one aligned 16 B load / store per thread, consecutive threads on consecutive chunks (fully coalesced);
//   * the selection and up to three planes are staged in the workgroup's LDS (16 guard bytes + 4 x PS bytes + 320, dynamic: 7 KB at 40 x 40,
//     64 KB at 127 x 127); the geometric ops work on whole chunks: object lift / place, Copy, Paste and Crop are FLAT SHIFTS of a plane (16
//     consecutive tile bytes through five consecutive dword reads behind one clamp and funnel shifts, the op's rectangle as a byte mask of
//     two column runs built by packed subtractions), Rotate / Flip gather cell by cell with an add + clamp + byte read and take the
//     rectangle as a mask (W < 16 keeps per-cell forms);
//   * reductions of a mask selection (any / sum / arg-max / bounding box) are formed per chunk on whole words and combined by two ballots and
//     six DPP reductions per wavefront, one LDS slot per wavefront and ONE workgroup barrier (tuple selections need none: arithmetic on the
//     tuple); the few LDS atomics left (dense pair, byte accounting; the CPU emulation of this step) are compiled with
//     -amdgpu-atomic-optimizer-strategy=DPP (arcle_amd/_lib.py): the compiler's default turns a same-address atomic into a scalar loop over
//     the lanes;
//   * FloodFill runs on 128-bit row boards: the eligibility boards are built on whole chunks (bytes equal to the seed's colour -> a 16-bit
//     map OR-ed into the one or two rows a chunk touches), then one thread per row: a pass pulls the fill from the rows above and below and
//     spreads it along the row's eligible runs with the carry trick (E + F ripples through a run of ones), until no row changes; the
//     region is coloured chunk by chunk from 16 board bits;
//   * per-env scalars (the 16-byte record, counters, op descriptor) are loaded by every thread — the same address: the compiler proves it
//     and issues scalar loads — and kept in scalar registers; thread 0 writes them back.
// What a launch costs is the number of instructions its wavefronts issue (one per ~4 cycles and SIMD, of any kind; one scalar instruction
// per cycle and CU) — and every wavefront of a workgroup runs the env's whole scalar program: hence few wavefronts per env, compile-time
// flag sets and the instruction diet of the whole-chunk forms (profiles/round6_experiments.txt §1-2).
// All control flow around barriers is workgroup-uniform (it depends only on the env's record, the op and the reduced selection).
//
// The same header is compiled by hipcc for gfx950 (arcle_big.hip) and by g++ for tests/emu/big_emu.cpp, which runs the body on host
// threads with a pthread barrier as the workgroup barrier — test infrastructure, never part of the product.
//
// Reference semantics restated here are cited per function (paths relative to /root/reference); they are the ones arcle_wave.h cites.
#pragma once
#include <stdint.h>

#include "arcle_big_params.h"

#if !defined(ARCLE_BIG_DEV) || !defined(ARCLE_BIG_ROWS)
#error "include through arcle_big.hip (or the test emulator), which defines ARCLE_BIG_DEV, ARCLE_BIG_ROWS and namespace bx"
#endif

namespace arcle_big {

struct alignas(16) V16 {
  uint32_t w[4];
};
union Chunk {
  V16 v;
  uint32_t w[4];
  int8_t b[16];
  uint8_t u[16];
};
struct Red {  // reduction block in LDS (64 bytes)
  int32_t any_nz, any_pos, sum;
  uint32_t amax;  // (value + 128) << 16 | (0xffff - cell): max value, first occurrence (np.argmax)
  int32_t x0, x1, y0, y1;
  int32_t neq;      // grid != answer somewhere / selection != selected somewhere
  int32_t flag[3];  // FloodFill: some row changed in pass k (slot k % 3)
  int32_t pad[4];
};
struct B128 {
  uint64_t lo, hi;
};

ARCLE_BIG_DEV int i8w(int x) { return (int)(int8_t)(uint8_t)(x & 0xff); }  // wrap to int8 (NumPy int8 arithmetic)
ARCLE_BIG_DEV int imin(int a, int b) { return a < b ? a : b; }
ARCLE_BIG_DEV int imax(int a, int b) { return a > b ? a : b; }
ARCLE_BIG_DEV int floordiv2(int a) { return a >> 1; }  // floor toward -inf (arithmetic shift)

ARCLE_BIG_DEV Chunk ldg(const int8_t* base, int c) {
  Chunk o;
  o.v = *reinterpret_cast<const V16*>(base + 16 * (size_t)c);
  return o;
}
// 16 bytes at any alignment (global memory takes unaligned vector loads; the compiler is told nothing about the alignment)
ARCLE_BIG_DEV Chunk ldu(const int8_t* p) {
  Chunk o;
  __builtin_memcpy(&o, p, 16);
  return o;
}
ARCLE_BIG_DEV void stg(int8_t* base, int c, const Chunk& v) { *reinterpret_cast<V16*>(base + 16 * (size_t)c) = v.v; }
ARCLE_BIG_DEV Chunk zero_chunk() {
  Chunk o;
  o.w[0] = o.w[1] = o.w[2] = o.w[3] = 0u;
  return o;
}

// the chunk c of a plane built cell by cell: cell(f, i, j) for the cells f = i * W + j < P of the chunk, zero behind P (row padding)
// f / W for a cell index 0 <= f < 127 * 128 of a plane: (f * wm) >> 21 with wm = 2^21 / W + 1 (BigParams::w_magic; exact: f * (wm - 2^21 / W)
// <= f < 2^21 / W for every W <= 127) — one 24-bit multiply and a shift instead of the ~25-instruction division by a run-time value
ARCLE_BIG_DEV int div_w(int f, uint32_t wm) { return (int)(bx::mul24((uint32_t)f, wm) >> 21); }

template <class F>
ARCLE_BIG_DEV Chunk build_chunk(int c, int W, uint32_t wm, int P, F&& cell) {
  Chunk o;
  int f = 16 * c;
  int i = div_w(f, wm), j = f - i * W;
#pragma unroll
  for (int k = 0; k < 16; k++, f++) {
    o.b[k] = f < P ? (int8_t)cell(f, i, j) : (int8_t)0;
    if (++j == W) {
      j = 0;
      ++i;
    }
  }
  return o;
}

// ---- task draw (the same function of (seed, global env id, episode) as arcle_wave.h draw_task, without augmentation) -------------------
ARCLE_BIG_DEV uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
ARCLE_BIG_DEV uint32_t mulhi32(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32); }
#define ARCLE_BIG_PERM_IDENTITY 0xFEDCBA9876543210ull
struct Draw {
  int entry;      // task-table index
  int rot_k;      // np.rot90 count 0..3
  uint64_t perm;  // colour permutation, nibble c = perm[c]
};
// (z0 -> problem and pair by multiply-shift range reduction; z1 -> the quarter turns and, its high word a 32-bit fraction consumed digit by
// digit, the Fisher-Yates swaps over the ten colours: the function arcle_wave.h draw_task and arcle_amd/sampling.py compute)
ARCLE_BIG_DEV Draw draw_task(const BigParams& p, int env, uint32_t episode) {
  const uint64_t G = 0x9E3779B97F4A7C15ull;
  const uint64_t z0 = mix64(p.seed + (uint64_t)(p.env_base + env) * G + (uint64_t)episode * 0xD1B54A32D192ED03ull);
  const int problem = (int)mulhi32((uint32_t)(z0 >> 32), (uint32_t)p.n_problems);
  const int sub = (int)mulhi32((uint32_t)z0, (uint32_t)p.pair_cnt[problem]);
  Draw d;
  d.entry = p.pair_off[problem] + sub;
  const uint64_t z1 = mix64(z0 + G);
  d.rot_k = (p.aug_flags & ARCLE_AUG_ROT90) ? (int)(z1 & 3u) : 0;
  d.perm = ARCLE_BIG_PERM_IDENTITY;
  if (p.aug_flags & ARCLE_AUG_PERMUTE) {
    uint32_t r = (uint32_t)(z1 >> 32);
    for (int i = 9; i > 0; i--) {
      const int j = (int)mulhi32(r, (uint32_t)(i + 1));
      r *= (uint32_t)(i + 1);
      const uint64_t a = (d.perm >> (4 * i)) & 15u, b = (d.perm >> (4 * j)) & 15u;
      d.perm = (d.perm & ~((15ull << (4 * i)) | (15ull << (4 * j)))) | (b << (4 * i)) | (a << (4 * j));
    }
  }
  return d;
}

// ---- 128-bit row boards ------------------------------------------------------------------------------------------------------------
// the fill spread along the runs of E (eligible cells) that hold a seed, towards the higher bits: E + S ripples a carry through every
// run from its lowest seed upwards; (E ^ (E + S)) & E are the cells the carry crossed
ARCLE_BIG_DEV B128 spread_up(B128 E, B128 S) {
  B128 t;
  t.lo = E.lo + S.lo;
  t.hi = E.hi + S.hi + (t.lo < E.lo ? 1ull : 0ull);
  B128 r;
  r.lo = ((E.lo ^ t.lo) & E.lo) | S.lo;
  r.hi = ((E.hi ^ t.hi) & E.hi) | S.hi;
  return r;
}
ARCLE_BIG_DEV B128 rev128(B128 a) {
  B128 r;
  r.lo = bx::brev64(a.hi);
  r.hi = bx::brev64(a.lo);
  return r;
}
ARCLE_BIG_DEV B128 spread(B128 E, B128 S) {
  const B128 u = spread_up(E, S);
  const B128 d = rev128(spread_up(rev128(E), rev128(S)));
  B128 r;
  r.lo = u.lo | d.lo;
  r.hi = u.hi | d.hi;
  return r;
}

struct Layout;
// bytes [lo, hi) of a chunk = 0xff (0 <= lo, hi <= 16; empty when hi <= lo): per byte k, k >= lo and k < hi as the top bits of two packed
// subtractions from (k | 0x80) — no borrow crosses a byte for subtrahends <= 128 — widened to whole bytes
ARCLE_BIG_DEV Chunk range_mask16(int lo, int hi) {
  const uint32_t l2 = (uint32_t)lo | ((uint32_t)lo << 8), h2 = (uint32_t)hi | ((uint32_t)hi << 8);
  const uint32_t l = l2 | (l2 << 16), h = h2 | (h2 << 16);  // the bound in every byte (two shift-or each)
  Chunk o;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t t = 0x83828180u + 0x04040404u * (uint32_t)q;
    const uint32_t m = (((t - l) & ~(t - h)) & 0x80808080u) >> 7;
    o.w[q] = (m << 8) - m;
  }
  return o;
}
// Flag bits the LEAN instantiations of the step kernel may see at run time (every other bit is known to be clear when the launcher picks
// them, so the code behind it folds away): the flag sets ARCVecEnv steps with when no row epilogue, dense reward or trace rule is on.
enum : uint32_t { LEAN_FLAGS = ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED | ARCLE_STEP_RESAMPLE | ARCLE_STEP_TRUNCATE | ARCLE_STEP_RESET_ON_SUBMIT };
// The per-workgroup context.  CPT_ > 0: the launch has at most CPT_ 16-byte chunks of a plane per thread (PS / 16 <= CPT_ x workgroup size)
// — every "my chunks" loop is then CPT_ guarded bodies in a row, no induction variable, no back edge; CPT_ = 0: a run-time loop.  LEAN_:
// compile-time knowledge of the launch — flags within LEAN_FLAGS, W >= 16, no byte accounting, no transition_rows scratch envs.
// CtxT<0, false> is the generic form: every run-time parameter honoured (the reset / row kernels, the emulator, tuning launches).
// Why several chunks per thread at all: the step is one scalar program per env that EVERY wavefront of the workgroup executes (~280 scalar
// instructions a wave) beside its share of the vector work, and a launch of these kernels takes about four cycles per instruction a SIMD
// issues, of any kind (profiles/round6_experiments.txt §2) — a plane spread over half as many wavefronts runs a third fewer instructions
// per env.
template <int CPT_, bool LEAN_>
struct CtxT {
  static constexpr int CPT = CPT_;
  static constexpr bool LEAN = LEAN_;
  const BigParams& p;
  int env, tid, NT, H, W, P, PS, nch;
  uint32_t wm;  // BigParams::w_magic (div_w)
  int8_t *S, *A, *B, *C;
  Red* red;
  uint64_t *Eb, *Fb;
  int8_t* sc;   // 32 bytes of LDS: the scalar block of a row (record, reward, terminated) — indexed per byte, so not in registers
  Layout* lay;  // the row layout being written / read, in LDS for the same reason (dynamic indexing of a local array means scratch memory)
  size_t po;
  bool sel01 = false;   // the selection tile S holds only 0 / 1 (tuple ingress: set by step_env_t; a compile-time fact of the instantiation)
  mutable uint32_t io;  // 16-byte global-memory accesses this thread issued (plane chunks, table chunks, mask chunks, row units): the
                        // byte accounting of arcle_enable_accounting — a register increment per access, summed per env when asked for
  ARCLE_BIG_DEV CtxT(const BigParams& p_, int env_, int8_t* lds)
      : p(p_), env(env_), tid(bx::tid()), NT(bx::nt()), H(p_.H), W(p_.W), P(p_.P), PS(p_.PS), nch(p_.PS >> 4), wm(p_.w_magic), io(0) {
    lds += LDS_GUARD;  // (shifted16 may read up to 16 bytes in front of a tile)
    S = lds;
    A = lds + PS;
    B = lds + 2 * PS;
    C = lds + 3 * PS;
    red = reinterpret_cast<Red*>(lds + 4 * PS);
    sc = lds + 4 * PS + 64;
    if (boards_in_tiles(PS, H)) {  // the row boards of the flood fill in the tiles no fill uses
      Eb = reinterpret_cast<uint64_t*>(B);
      Fb = reinterpret_cast<uint64_t*>(C);
    } else {
      Eb = reinterpret_cast<uint64_t*>(lds + 4 * PS + 64 + 256);
      Fb = Eb + 256;
    }
    lay = reinterpret_cast<Layout*>(sc + 32);
    po = (size_t)env * (size_t)PS;
  }
  ARCLE_BIG_DEV void count(uint32_t k = 1) const {
    if (!LEAN_) io += k;
  }
  ARCLE_BIG_DEV bool wide() const { return LEAN_ || W >= 16; }  // whole-chunk (SWAR) forms of the geometric ops
  // 0xff in every byte of a selection word that is > 0 / != 0 (pos_bytes / nz_bytes below; three instructions when the bytes are 0 / 1)
  ARCLE_BIG_DEV uint32_t sel_pos(uint32_t v) const;
  ARCLE_BIG_DEV uint32_t sel_nz(uint32_t v) const;
  ARCLE_BIG_DEV int8_t* g(int pl) const { return p.plane[pl] + po; }
  ARCLE_BIG_DEV bool has(int pl) const { return p.plane[pl] != nullptr; }
  // chunk c of a state plane of this env: load / store (counted)
  ARCLE_BIG_DEV Chunk gl(int pl, int c) const {
    count();
    return ldg(p.plane[pl] + po, c);
  }
  ARCLE_BIG_DEV void gs(int pl, int c, const Chunk& v) const {
    count();
    stg(p.plane[pl] + po, c, v);
  }
  // global plane -> LDS tile / fill
  ARCLE_BIG_DEV void stage(int8_t* dst, const int8_t* src) const;
  ARCLE_BIG_DEV void stage_g(int8_t* dst, int pl) const { stage(dst, p.plane[pl] + po); }
  ARCLE_BIG_DEV void fill(int8_t* dst, const Chunk& v) const;
};
typedef CtxT<0, false> Ctx;
// "for every chunk c of a plane that this thread owns" (c = tid, tid + NT, ...; with X::CPT > 0 at most that many: the loop unrolls fully.
// Kept ROLLED — a loop of uniform trip count with the thread's guard inside — the two-chunk kernels spill fewer scalar registers but run
// ~10 % more instructions per env and lose 3 %: profiles/round6_experiments.txt §2)
// (CPT >= 2: a launch that NEEDS its second chunk has fewer threads than the plane has chunks, so every thread's FIRST chunk exists — no
// guard around it, three scalar instructions fewer per loop; the launchers pick CPT = ceil(chunks / threads) exactly)
#define BIG_EACH_CHUNK(x, c)                                                                                                              \
  _Pragma("unroll") for (int c = (x).tid, n_##c = 0; (((x).CPT >= 2 && n_##c == 0) || c < (x).nch) && ((x).CPT == 0 || n_##c < (x).CPT); \
                         c += (x).NT, n_##c++)
template <int CPT_, bool LEAN_>
ARCLE_BIG_DEV void CtxT<CPT_, LEAN_>::stage(int8_t* dst, const int8_t* src) const {
  // (all loads first, then the LDS writes: the chunks' round trips overlap)
  if (CPT_ > 0) {
    Chunk v[CPT_ > 0 ? CPT_ : 1];
#pragma unroll
    for (int k = 0; k < CPT_; k++) {
      const int c = tid + k * NT;
      if ((CPT_ >= 2 && k == 0) || c < nch) {
        count();
        v[k] = ldg(src, c);
      }
    }
#pragma unroll
    for (int k = 0; k < CPT_; k++) {
      const int c = tid + k * NT;
      if ((CPT_ >= 2 && k == 0) || c < nch) stg(dst, c, v[k]);
    }
    return;
  }
  for (int c = tid; c < nch; c += NT) {
    count();
    stg(dst, c, ldg(src, c));
  }
}
template <int CPT_, bool LEAN_>
ARCLE_BIG_DEV void CtxT<CPT_, LEAN_>::fill(int8_t* dst, const Chunk& v) const {
  BIG_EACH_CHUNK(*this, c) stg(dst, c, v);
}

// init_state (base.py:155-166 + o2arcenv.py:16-34 / arcenv.py:81-89): grid := input, the other state planes := 0, the record's state
// fields; `src` = the input plane to copy (the env's own, or a task-table entry that is also written to PL_INPUT)
template <class X>
ARCLE_BIG_DEV void init_planes(const X& x, const int8_t* src, bool write_input) {
  const Chunk z = zero_chunk();
  BIG_EACH_CHUNK(x, c) {
    const Chunk in = ldg(src, c);
    x.count();
    if (write_input) x.gs(ARCLE_PL_INPUT, c, in);
    x.gs(ARCLE_PL_GRID, c, in);
    if (x.has(ARCLE_PL_SELECTED)) x.gs(ARCLE_PL_SELECTED, c, z);
    if (x.has(ARCLE_PL_CLIP)) x.gs(ARCLE_PL_CLIP, c, z);
    if (x.has(ARCLE_PL_OBJECT)) x.gs(ARCLE_PL_OBJECT, c, z);
    if (x.has(ARCLE_PL_OBJECT_SEL)) x.gs(ARCLE_PL_OBJECT_SEL, c, z);
    if (x.has(ARCLE_PL_BACKGROUND)) x.gs(ARCLE_PL_BACKGROUND, c, z);
  }
}
// The 16-byte record of an env as the step kernel holds it: FOUR WORDS (scalar registers: the record arrives through a scalar load), a field
// read is one bit-field extract and a write one insert into its word — kept apart as sixteen byte values the compiler re-assembles all four
// words from their bytes before the store, ~30 scalar instructions at the end of every step.  Indexable like the byte array it replaces.
struct Rec16 {
  uint32_t w[4];
  struct Ref {
    uint32_t* p;
    int sh;
    ARCLE_BIG_DEV operator int8_t() const { return (int8_t)(uint8_t)(*p >> sh); }
    ARCLE_BIG_DEV Ref& operator=(int8_t v) {
      *p = (*p & ~(0xffu << sh)) | ((uint32_t)(uint8_t)v << sh);
      return *this;
    }
    ARCLE_BIG_DEV Ref& operator=(const Ref& o) { return *this = (int8_t)o; }
  };
  ARCLE_BIG_DEV Ref operator[](int i) { return Ref{&w[i >> 2], 8 * (i & 3)}; }
  ARCLE_BIG_DEV int8_t operator[](int i) const { return (int8_t)(uint8_t)(w[i >> 2] >> (8 * (i & 3))); }
};
template <class R>
ARCLE_BIG_DEV void init_rec(R&& r, int max_trial) {
  r[ARCLE_REC_GRID_DIM] = r[ARCLE_REC_INPUT_DIM];
  r[ARCLE_REC_GRID_DIM + 1] = r[ARCLE_REC_INPUT_DIM + 1];
  r[ARCLE_REC_CLIP_DIM] = r[ARCLE_REC_CLIP_DIM + 1] = 0;
  r[ARCLE_REC_OBJECT_DIM] = r[ARCLE_REC_OBJECT_DIM + 1] = 0;
  r[ARCLE_REC_OBJECT_POS] = r[ARCLE_REC_OBJECT_POS + 1] = 0;
  r[ARCLE_REC_TRIALS] = (int8_t)i8w(max_trial);
  r[ARCLE_REC_TERMINATED] = 0;
  r[ARCLE_REC_ACTIVE] = 0;
  r[ARCLE_REC_PARITY] = 0;
}

// Copies task-table entry t — optionally augmented: colours permuted (the un-padded grids only: the padding stays 0), np.rot90(., k) —
// into the env's input and answer planes and the record's dims, then runs init_state's plane part.  Returns false (nothing written) when
// a quarter turn does not fit the H x W plane; `soften`: such a turn is dropped (k &= 2) instead — device-drawn augmentations never fail
// (the rule of arcle_wave.h load_task).  Workgroup-uniform; contains barriers when it augments.
template <class X, class R>
ARCLE_BIG_DEV bool load_task(const X& x, R&& r, int t, int rot_k, uint64_t perm, bool soften) {
  const BigParams& p = x.p;
  int ih = p.tbl_in_dim[2 * (size_t)t], iw = p.tbl_in_dim[2 * (size_t)t + 1];
  int ah = p.tbl_ans_dim[2 * (size_t)t], aw = p.tbl_ans_dim[2 * (size_t)t + 1];
  if ((rot_k & 1) && (iw > x.H || ih > x.W || aw > x.H || ah > x.W)) {
    if (!soften) return false;
    rot_k &= 2;
  }
  const int8_t* const tin = p.tbl_in + (size_t)t * x.PS;
  const int8_t* const tan = p.tbl_ans + (size_t)t * x.PS;
  if (rot_k == 0 && perm == ARCLE_BIG_PERM_IDENTITY) {
    BIG_EACH_CHUNK(x, c) {
      x.count();
      x.gs(ARCLE_PL_ANSWER, c, ldg(tan, c));
    }
    init_planes(x, tin, true);
  } else {
    x.stage(x.A, tin);
    x.stage(x.B, tan);
    bx::sync();
    const int W = x.W;
    for (int which = 0; which < 2; which++) {
      const int8_t* const src = which ? x.B : x.A;
      const int h = which ? ah : ih, w = which ? aw : iw;
      int nh = h, nw = w, ai = W, bj = 1, c0 = 0;  // identity
      if (rot_k == 1) { ai = -1; bj = W; c0 = w - 1; nh = w; nw = h; }                // np.rot90(x, 1)[i, j] = x[j, w-1-i]
      else if (rot_k == 2) { ai = -W; bj = -1; c0 = (h - 1) * W + (w - 1); }          // x[h-1-i, w-1-j]
      else if (rot_k == 3) { ai = 1; bj = -W; c0 = (h - 1) * W; nh = w; nw = h; }      // x[h-1-j, i]
      const int dpl = which ? ARCLE_PL_ANSWER : ARCLE_PL_INPUT;
      BIG_EACH_CHUNK(x, c)
        x.gs(dpl, c, build_chunk(c, W, x.wm, x.P, [&](int, int i, int j) {
              const bool in = i < nh && j < nw;
              const int v = (uint8_t)src[in ? c0 + ai * i + bj * j : 0];
              const int pv = v < 16 ? (int)((perm >> (4 * v)) & 15u) : v;  // (cells beyond the palette keep their value)
              return in ? (int8_t)pv : (int8_t)0;
            }));
      if (which) { ah = nh; aw = nw; } else { ih = nh; iw = nw; }
    }
    init_planes(x, x.g(ARCLE_PL_INPUT), false);  // (every thread reads back the chunks it has just written itself)
  }
  r[ARCLE_REC_INPUT_DIM] = (int8_t)ih;
  r[ARCLE_REC_INPUT_DIM + 1] = (int8_t)iw;
  r[ARCLE_REC_ANSWER_DIM] = (int8_t)ah;
  r[ARCLE_REC_ANSWER_DIM + 1] = (int8_t)aw;
  return true;
}

ARCLE_BIG_DEV Chunk rect_mask16(int c, int W, uint32_t wm, int r0, int r1, int c0, int c1);  // (defined with the whole-chunk forms below)
ARCLE_BIG_DEV uint32_t nz_bytes(uint32_t v);

// answer.shape == grid_dim and grid[:h,:w] == answer (base.py:177, o2arcenv.py:124-127); workgroup-uniform result.
// (two barriers; the caller has made the grid plane in global memory final and visible — a barrier since its last store)
template <class X, class R>
ARCLE_BIG_DEV bool grid_equals_answer(const X& x, const R& r) {
  const int gh = r[ARCLE_REC_GRID_DIM], gw = r[ARCLE_REC_GRID_DIM + 1];
  if (gh != r[ARCLE_REC_ANSWER_DIM] || gw != r[ARCLE_REC_ANSWER_DIM + 1]) return false;
  if (x.tid == 0) x.red->neq = 0;
  bx::sync();
  bool differs = false;
  const int lastc = imin(x.nch, (gh * x.W + 15) >> 4);  // cells of the rows >= gh are never compared
  BIG_EACH_CHUNK(x, c) {
    if (c >= lastc) break;
    const Chunk a = x.gl(ARCLE_PL_GRID, c), b = x.gl(ARCLE_PL_ANSWER, c);
    if ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3])) {
      if (x.wide()) {  // (whole words: the differing bytes under the gh x gw rectangle's byte mask — no loop over the cells)
        const Chunk in = rect_mask16(c, x.W, x.wm, 0, gh, 0, gw);
        differs |= (((a.w[0] ^ b.w[0]) & in.w[0]) | ((a.w[1] ^ b.w[1]) & in.w[1]) | ((a.w[2] ^ b.w[2]) & in.w[2]) | ((a.w[3] ^ b.w[3]) & in.w[3])) != 0u;
      } else {
        int f = 16 * c;
        int i = div_w(f, x.wm), j = f - i * x.W;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          if (i < gh && j < gw && a.b[k] != b.b[k]) differs = true;
          if (++j == x.W) {
            j = 0;
            ++i;
          }
        }
      }
    }
  }
  if (differs) x.red->neq = 1;
  bx::sync();
  return x.red->neq == 0;
}

// ---- flattened rows --------------------------------------------------------------------------------------------------------------------
// A logical row is a list of segments: a state plane (P bytes) or a few bytes of the scalar block sc[] (the 16-byte record followed by
// reward int32 LE at 16, terminated at 20).  Layouts: gymnasium FlattenObservation of the state dict (keys sorted, object_states in
// place) or of its FilterO2ARC subset (agents/env.py:109-126) — the orders arcle_wave.h flat_row writes — and the packed gather row
// grid | grid_dim | reward | terminated.
struct Seg {
  int16_t plane;  // -1: scalars
  int16_t soff;   // offset in sc[]
  int32_t start, len;
};
struct Layout {
  Seg s[18];
  int n, len;
  ARCLE_BIG_DEV void add_plane(int pl, int P) {
    s[n].plane = (int16_t)pl;
    s[n].soff = 0;
    s[n].start = len;
    s[n].len = P;
    len += P;
    n++;
  }
  ARCLE_BIG_DEV void add_sc(int off, int k) {
    s[n].plane = -1;
    s[n].soff = (int16_t)off;
    s[n].start = len;
    s[n].len = k;
    len += k;
    n++;
  }
};
ARCLE_BIG_DEV void flat_layout(const BigParams& p, int filtered, Layout& L) {  // (one thread builds it in LDS; a barrier follows)
  L.n = 0;
  L.len = 0;
  const bool o2 = p.plane[ARCLE_PL_SELECTED] != nullptr, clip = p.plane[ARCLE_PL_CLIP] != nullptr;
  const int P = p.P;
  if (filtered) {
    L.add_sc(ARCLE_REC_ACTIVE, 1);
    L.add_plane(ARCLE_PL_CLIP, P);
    L.add_sc(ARCLE_REC_CLIP_DIM, 2);
    L.add_plane(ARCLE_PL_GRID, P);
    L.add_sc(ARCLE_REC_GRID_DIM, 2);
    L.add_plane(ARCLE_PL_OBJECT, P);
    L.add_sc(ARCLE_REC_OBJECT_DIM, 2);
    L.add_sc(ARCLE_REC_OBJECT_POS, 2);
    L.add_sc(ARCLE_REC_TRIALS, 1);
    return;
  }
  if (clip) {
    L.add_plane(ARCLE_PL_CLIP, P);
    L.add_sc(ARCLE_REC_CLIP_DIM, 2);
  }
  L.add_plane(ARCLE_PL_GRID, P);
  L.add_sc(ARCLE_REC_GRID_DIM, 2);
  L.add_plane(ARCLE_PL_INPUT, P);
  L.add_sc(ARCLE_REC_INPUT_DIM, 2);
  if (o2) {
    L.add_sc(ARCLE_REC_ACTIVE, 1);
    L.add_plane(ARCLE_PL_BACKGROUND, P);
    L.add_plane(ARCLE_PL_OBJECT, P);
    L.add_sc(ARCLE_REC_OBJECT_DIM, 2);
    L.add_sc(ARCLE_REC_OBJECT_POS, 2);
    L.add_plane(ARCLE_PL_OBJECT_SEL, P);
    L.add_sc(ARCLE_REC_PARITY, 1);
    L.add_plane(ARCLE_PL_SELECTED, P);
  }
  L.add_sc(ARCLE_REC_TERMINATED, 1);
  L.add_sc(ARCLE_REC_TRIALS, 1);
}
ARCLE_BIG_DEV void packed_layout(const BigParams& p, Layout& L) {
  L.n = 0;
  L.len = 0;
  L.add_plane(ARCLE_PL_GRID, p.P);
  L.add_sc(ARCLE_REC_GRID_DIM, 2);
  L.add_sc(16, 4);
  L.add_sc(20, 1);
}

// writes bytes [0, limit) of the row at `dst` (16-byte aligned, limit a multiple of 16): the layout's bytes, zeros behind them.  A plane
// segment lands at an arbitrary byte offset of the row, so each thread assembles whole 16-byte units of the row (one aligned store —
// rows may live in pinned host memory) from the bytes of the segments that cross it.
template <class X>
ARCLE_BIG_DEV void write_row(const X& x, const Layout& L, const int8_t* sc, int8_t* dst, int limit) {
  const int nu = limit >> 4;
  int seg = 0;  // the segment holding this thread's current unit: units only move forward, so the search resumes where it stopped
  auto unit = [&](int u) {
    const int b0 = 16 * u;
    Chunk v = zero_chunk();
    if (b0 < L.len) {
      while (seg + 1 < L.n && L.s[seg + 1].start <= b0) seg++;
      int si = seg;
      const Seg s0 = L.s[si];
      if (s0.plane >= 0 && b0 + 16 <= s0.start + s0.len) {
        // the unit lies inside one plane segment: 16 consecutive bytes of the plane
        v = ldu(x.p.plane[s0.plane] + x.po + (b0 - s0.start));
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const int b = b0 + k;
          while (si + 1 < L.n && L.s[si + 1].start <= b) si++;
          if (b < L.len) {
            const Seg& s = L.s[si];
            const int o = b - s.start;
            v.b[k] = s.plane < 0 ? sc[s.soff + o] : x.p.plane[s.plane][x.po + o];
          }
        }
      }
    }
    x.count(2);  // (the unit stored + the 16 source bytes read)
    return v;
  };
  // four units per round: their loads are issued together, then the stores (a row in pinned host memory is written across PCIe — a loop
  // of load -> store -> load would pay a round trip per unit)
  for (int u = x.tid; u < nu; u += 4 * x.NT) {
    const int u1 = u + x.NT, u2 = u + 2 * x.NT, u3 = u + 3 * x.NT;
    const Chunk v0 = unit(u);
    const Chunk v1 = u1 < nu ? unit(u1) : zero_chunk();
    const Chunk v2 = u2 < nu ? unit(u2) : zero_chunk();
    const Chunk v3 = u3 < nu ? unit(u3) : zero_chunk();
    stg(dst, u, v0);
    if (u1 < nu) stg(dst, u1, v1);
    if (u2 < nu) stg(dst, u2, v2);
    if (u3 < nu) stg(dst, u3, v3);
  }
}

// the FLAT_OBS / PACK_OBS epilogue of a step (and the stand-alone flatten / pack kernels): rows of the env's CURRENT state
template <class X, class R>
ARCLE_BIG_DEV void emit_rows(const X& x, const R& r, uint32_t flags, int reward, int term, int cnt0, int cnt1, bool truncated,
                             uint32_t st) {
  const BigParams& p = x.p;
  // (entered behind a barrier: nobody is still reading the LDS tail area)
  int8_t* const sc = x.sc;
  if (x.tid == 0) {
#pragma unroll
    for (int k = 0; k < 16; k++) sc[k] = r[k];
    sc[16] = (int8_t)(reward & 0xff);
    sc[17] = (int8_t)((reward >> 8) & 0xff);
    sc[18] = (int8_t)((reward >> 16) & 0xff);
    sc[19] = (int8_t)((reward >> 24) & 0xff);
    sc[20] = (int8_t)term;
    sc[21] = sc[22] = sc[23] = 0;
  }
  if ((flags & ARCLE_STEP_PACK_OBS) && p.pack_out) {
    if (x.tid == 0) packed_layout(p, *x.lay);
    bx::sync();
    const int stride = packed_stride(p.P);
    write_row(x, *x.lay, sc, reinterpret_cast<int8_t*>(p.pack_out) + (size_t)x.env * stride, stride);
    bx::sync();
  }
  if ((flags & ARCLE_STEP_FLAT_OBS) && p.flat_out) {
    if (x.tid == 0) flat_layout(p, p.flat_filter, *x.lay);
    bx::sync();
    int8_t* const row = p.flat_out + (size_t)x.env * p.flat_stride;
    write_row(x, *x.lay, sc, row, p.flat_stride - (p.flat_tail ? 16 : 0));
    if (p.flat_tail) {
      // int32 reward | int32 action_steps | int32 submit_count | uint8 terminated | uint8 truncated | uint8 status | seq
      // (arcle_set_flat_output_ex / arcle_set_flat_seq: the last word behind a system-scope release, after every row store of the
      // workgroup — hence the barrier)
      bx::sync_release();
      if (x.tid == 0) {
        uint32_t* const t = reinterpret_cast<uint32_t*>(row + (p.flat_stride - 16));
        const uint32_t last = (uint32_t)term | ((uint32_t)truncated << 8) | ((st & 0xffu) << 16);
        t[0] = (uint32_t)reward;
        t[1] = (uint32_t)cnt0;
        t[2] = (uint32_t)cnt1;
        if (p.flat_seq) bx::release_store_system(t + 3, last | ((uint32_t)p.flat_seq << 24));
        else t[3] = last;
      }
    }
  }
}

// ---- whole-chunk (SWAR) forms of the gathers, for W >= 16 (a chunk then spans at most two plane rows) ------------------------------------
// The object lift / place, Copy, Paste and Crop are FLAT SHIFTS of a plane: destination cell f reads source cell f + delta for one delta per
// op (rows and columns move together in the row-major layout), valid wherever the destination lies inside the op's rectangle.  So a
// chunk's 16 source bytes are 16 CONSECUTIVE bytes of the LDS tile — five aligned dword reads and four funnel shifts — and the rectangle
// becomes a byte mask built from (at most) two column runs: ~130 vector instructions per chunk instead of ~25 per cell
// (profiles/round5_experiments.txt §19-20).  W < 16 keeps the per-cell form.
ARCLE_BIG_DEV Chunk shifted16(const int8_t* tile, int off, int PS) {  // bytes [off, off + 16) of the tile; bytes outside [0, PS) are unspecified
  const uint32_t* const t32 = reinterpret_cast<const uint32_t*>(tile);
  // the five words d .. d + 4, d clamped to [-4, PS / 4]: a window with any byte inside the tile is read where it lies (its words outside the
  // tile come from the 16 guard bytes in front of the first tile, a neighbouring tile or the block behind the last one — never from outside
  // the workgroup's LDS), a window wholly outside is unspecified anyway.  ONE clamp and five consecutive words (paired reads).
  const int d = imin(imax(off >> 2, -4), PS >> 2);
  const uint32_t sh = 8u * (uint32_t)(off & 3);
  uint32_t w[5];
#pragma unroll
  for (int q = 0; q < 5; q++) w[q] = t32[d + q];
  Chunk o;
#pragma unroll
  for (int q = 0; q < 4; q++) o.w[q] = bx::alignbit(w[q + 1], w[q], sh);  // (v_alignbit_b32: sh = 0 / 8 / 16 / 24)
  return o;
}
// byte mask of the cells of chunk c inside rows [r0, r1) x columns [c0, c1) (W >= 16; the rectangle lies inside the plane)
ARCLE_BIG_DEV Chunk rect_mask16(int c, int W, uint32_t wm, int r0, int r1, int c0, int c1) {
  const int f0 = 16 * c, i0 = div_w(f0, wm), j0 = f0 - i0 * W, n0 = imin(16, W - j0);
  Chunk m = zero_chunk();
  if (i0 >= r0 && i0 < r1) m = range_mask16(imin(imax(c0 - j0, 0), n0), imin(imax(c1 - j0, 0), n0));
  if (n0 < 16 && i0 + 1 >= r0 && i0 + 1 < r1) {
    const Chunk m2 = range_mask16(n0 + imin(imax(c0, 0), 16 - n0), n0 + imin(imax(c1, 0), 16 - n0));
#pragma unroll
    for (int q = 0; q < 4; q++) m.w[q] |= m2.w[q];
  }
  return m;
}
ARCLE_BIG_DEV uint32_t nz_bytes(uint32_t v) {  // 0xff in every byte of v that is non-zero
  const uint32_t t = ((((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u) >> 7;
  return (t << 8) - t;
}
ARCLE_BIG_DEV uint32_t pos_bytes(uint32_t v) {  // 0xff in every byte of v that is > 0 as an int8
  const uint32_t t = ((((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) & ~v) & 0x80808080u) >> 7;
  return (t << 8) - t;
}
template <int CPT_, bool LEAN_>
ARCLE_BIG_DEV uint32_t CtxT<CPT_, LEAN_>::sel_pos(uint32_t v) const {
  // (bit 0 of every byte first: a shifted window's bytes from outside the tile are anything, and must not borrow into their neighbours)
  const uint32_t b = v & 0x01010101u;
  return sel01 ? (b << 8) - b : pos_bytes(v);
}
template <int CPT_, bool LEAN_>
ARCLE_BIG_DEV uint32_t CtxT<CPT_, LEAN_>::sel_nz(uint32_t v) const {
  const uint32_t b = v & 0x01010101u;
  return sel01 ? (b << 8) - b : nz_bytes(v);
}

// Rotate / Flip (W >= 16): chunk c of dst[:nh, :nw] = src[c0 + ai * i + bj * j] — a true gather (a column of the source becomes a row), but the
// source index of consecutive cells advances by bj, restarts once where the chunk crosses into its second plane row, and the [:nh, :nw]
// rectangle is applied as a byte mask afterwards: an add, a clamp (v_med3) and a byte read per cell instead of the compare chain
ARCLE_BIG_DEV Chunk gather_affine16(const int8_t* tile, int c, int W, uint32_t wm, int P, int nh, int nw, int c0, int ai, int bj) {
  const int f0 = 16 * c, i0 = div_w(f0, wm), j0 = f0 - i0 * W, n0 = imin(16, W - j0);
  const int ta = c0 + ai * i0 + bj * j0, tb = c0 + ai * (i0 + 1) - bj * n0;  // cell k: ta + bj * k in the first row, tb + bj * k in the second
  Chunk o;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int t = (k < n0 ? ta : tb) + bj * k;
    o.b[k] = tile[imin(imax(t, 0), P - 1)];
  }
  const Chunk in = rect_mask16(c, W, wm, 0, nh, 0, nw);
#pragma unroll
  for (int q = 0; q < 4; q++) o.w[q] &= in.w[q];
  return o;
}

// Copy / CropGrid (W >= 16): chunk c of the h x w tile whose cell (i, j) is plane A's cell (x0 + i, y0 + j) where the selection S is non-zero
template <class X>
ARCLE_BIG_DEV Chunk cut_out16(const X& x, int c, int x0, int y0, int h, int w) {
  const int delta = x0 * x.W + y0;
  const Chunk sv = shifted16(x.S, 16 * c + delta, x.PS), av = shifted16(x.A, 16 * c + delta, x.PS);
  const Chunk in = rect_mask16(c, x.W, x.wm, 0, h, 0, w);
  Chunk o;
#pragma unroll
  for (int q = 0; q < 4; q++) o.w[q] = av.w[q] & in.w[q] & x.sel_nz(sv.w[q]);
  return o;
}

// ---- FloodFill (color.py:88-100, dfs :8-30): 4-connected region of (sx, sy) among the cells of the gh x gw grid that hold its colour.
// The grid is staged in A.  Iterative propagation is equivalent to the reference's DFS (the visited set does not depend on the order).
template <class X>
ARCLE_BIG_DEV void flood_fill(const X& x, int gh, int gw, int sx, int sy, int colour) {
  const int W = x.W, H = x.H;
  const int col = x.A[sx * W + sy];
  B128* const E = reinterpret_cast<B128*>(x.Eb);
  B128* const F = reinterpret_cast<B128*>(x.Fb);
  if (x.wide()) {
    // The eligibility boards E from whole chunks (W >= 16): a chunk's cells that hold the seed's colour inside the gh x gw grid as a 16-bit
    // map (bytes equal <=> XOR is zero; the grid's rectangle as a byte mask; four multiply-gathers) OR-ed into the one or two rows it
    // touches — ~60 vector instructions a chunk instead of ~6 per cell of a row in ONE thread (at 64 x 64 that loop was 400 instructions
    // long on a wavefront most of whose lanes idled).
    int32_t* const E32 = reinterpret_cast<int32_t*>(E);
    for (int i = x.tid; i < H; i += x.NT) E[i].lo = E[i].hi = 0;
    bx::sync();
    const uint32_t colw = ((uint32_t)col & 0xffu) * 0x01010101u;
    BIG_EACH_CHUNK(x, c) {
      const Chunk v = ldg(x.A, c), in = rect_mask16(c, W, x.wm, 0, gh, 0, gw);
      uint32_t m16 = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t eq = ~nz_bytes(v.w[q] ^ colw) & in.w[q] & 0x01010101u;
        m16 |= ((bx::mul32(eq, 0x01020408u) >> 24) & 15u) << (4 * q);
      }
      if (m16) {
        const int f0 = 16 * c, i0 = div_w(f0, x.wm), j0 = f0 - i0 * W, n0 = imin(16, W - j0);
        const uint32_t part0 = m16 & ((1u << n0) - 1u), part1 = m16 >> n0;
        if (part0) {  // columns j0 .. j0 + n0 - 1 of row i0: at most two words of the row's board
          const int wi = j0 >> 5, sh = j0 & 31;
          bx::lds_or(E32 + 4 * i0 + wi, (int)(part0 << sh));
          if (sh > 16 && (part0 >> (32 - sh))) bx::lds_or(E32 + 4 * i0 + wi + 1, (int)(part0 >> (32 - sh)));
        }
        if (part1) bx::lds_or(E32 + 4 * (i0 + 1), (int)part1);  // columns 0 .. of row i0 + 1 (a row of the grid: part1 is empty behind the last)
      }
    }
    bx::sync();
    for (int i = x.tid; i < H; i += x.NT) {
      B128 f;
      f.lo = f.hi = 0;
      if (i == sx) {
        if (sy < 64) f.lo = 1ull << sy;
        else f.hi = 1ull << (sy - 64);
        f = spread(E[i], f);
      }
      F[i] = f;
    }
  } else
  for (int i = x.tid; i < H; i += x.NT) {
    B128 e;
    e.lo = e.hi = 0;
    if (i < gh) {
      const int8_t* row = x.A + i * W;
      for (int j = 0; j < gw; j++)
        if (row[j] == col) {
          if (j < 64) e.lo |= 1ull << j;
          else e.hi |= 1ull << (j - 64);
        }
    }
    B128 f;
    f.lo = f.hi = 0;
    if (i == sx) {
      if (sy < 64) f.lo = 1ull << sy;
      else f.hi = 1ull << (sy - 64);
      f = spread(e, f);
    }
    E[i] = e;
    F[i] = f;
  }
  if (x.tid == 0) x.red->flag[0] = x.red->flag[1] = x.red->flag[2] = 0;
  bx::sync();
  // Chaotic relaxation: inside a pass a thread re-reads its neighbours' boards LIVE (volatile LDS reads, no barrier) FILL_INNER times
  // and publishes its own row as soon as it grows — rows of one wavefront advance in lock-step, so the fill climbs FILL_INNER rows of a
  // vertical corridor per pass instead of one (one barrier per pass instead of two per row).  Every intermediate board is a subset of
  // the region (the update is monotone), so whatever the interleaving the fixpoint is the reference's region; a pass in which NO thread
  // changed anything evaluated every row against boards that were constant throughout: the fixpoint.  Flag slots rotate over three
  // passes: slot (k + 1) % 3 is cleared during pass k — last read at the end of pass k - 2, behind a barrier every thread has passed.
  volatile uint64_t* const Fv = reinterpret_cast<volatile uint64_t*>(F);
  // (a thread's rows are tid, tid + NT, ...: ARCLE_BIG_ROWS of them at most — 1 on the GPU, where a workgroup has at least 128 threads — in
  // loops of constant trip count, so that the boards stay in registers)
  B128 mine[ARCLE_BIG_ROWS], elig[ARCLE_BIG_ROWS];
#pragma unroll
  for (int k = 0; k < ARCLE_BIG_ROWS; k++) {
    const int i = x.tid + k * x.NT;
    if (i < H) {
      mine[k] = F[i];
      elig[k] = E[i];
    }
  }
  for (int pass = 0;; pass++) {
    bool changed = false;
    for (int it = 0; it < FILL_INNER; it++) {
#pragma unroll
      for (int k = 0; k < ARCLE_BIG_ROWS; k++) {
        const int i = x.tid + k * x.NT;
        if (i >= H) continue;
        const B128 cur = mine[k], e = elig[k];
        B128 s = cur;
        if (i > 0) {
          s.lo |= Fv[2 * (i - 1)];
          s.hi |= Fv[2 * (i - 1) + 1];
        }
        if (i + 1 < H) {
          s.lo |= Fv[2 * (i + 1)];
          s.hi |= Fv[2 * (i + 1) + 1];
        }
        s.lo &= e.lo;
        s.hi &= e.hi;
        if (s.lo != cur.lo || s.hi != cur.hi) {
          const B128 n = spread(e, s);
          Fv[2 * i] = n.lo;
          Fv[2 * i + 1] = n.hi;
          mine[k] = n;
          changed = true;
        }
      }
    }
    if (changed) x.red->flag[pass % 3] = 1;
    if (x.tid == 0) x.red->flag[(pass + 1) % 3] = 0;
    bx::sync();
    if (!x.red->flag[pass % 3]) break;
  }
  // the region takes the colour: chunks of the staged grid, rewritten where the board has a bit
  if (x.wide()) {  // (whole chunks: the chunk's 16 board bits out of its one or two rows, widened to bytes by a multiply per nibble)
    const uint32_t* const F32 = reinterpret_cast<const uint32_t*>(F);
    const uint32_t cw = ((uint32_t)colour & 0xffu) * 0x01010101u;
    BIG_EACH_CHUNK(x, c) {
      const int f0 = 16 * c, i0 = div_w(f0, x.wm), j0 = f0 - i0 * W, n0 = imin(16, W - j0);
      uint32_t m16 = 0;
      if (i0 < H) {
        const int wi = j0 >> 5;
        const uint32_t lo = F32[4 * i0 + wi], hi = wi < 3 ? F32[4 * i0 + wi + 1] : 0u;
        m16 = bx::alignbit(hi, lo, (uint32_t)(j0 & 31)) & ((1u << n0) - 1u);
      }
      if (n0 < 16 && i0 + 1 < H) m16 |= (F32[4 * (i0 + 1)] & ((1u << (16 - n0)) - 1u)) << n0;
      if (!m16) continue;
      Chunk o = ldg(x.A, c);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t b = bx::mul24((m16 >> (4 * q)) & 15u, 0x00204081u) & 0x01010101u, m = (b << 8) - b;
        o.w[q] = (o.w[q] & ~m) | (cw & m);
      }
      x.gs(ARCLE_PL_GRID, c, o);
    }
    return;
  }
  BIG_EACH_CHUNK(x, c) {
    bool any = false;
    const Chunk o = build_chunk(c, W, x.wm, x.P, [&](int f, int i, int j) {
      const B128 fr = F[i];
      const bool in = j < 64 ? ((fr.lo >> j) & 1ull) != 0 : ((fr.hi >> (j - 64)) & 1ull) != 0;
      any |= in;
      return in ? (int8_t)colour : x.A[f];
    });
    if (any) x.gs(ARCLE_PL_GRID, c, o);
  }
}

// _apply_patch (object.py:113-138) + _apply_sel (object.py:140-165): grid := background, selected := 0, then the object tile `O`
// (LDS, tile origin at cell 0) is drawn at object_pos wherever it is > 0 and `Q` becomes the selection there; clipped to grid_dim.
// `cut`: nullptr when `bg` is the background itself; else `bg` is the grid and the background is where(cut > 0, 0, grid) (object.py:87-88).
// `lift_delta` >= 0 (W >= 16, a fresh selection that is only MOVED): there are no object tiles — the object is the grid `bg` under the
// selection `cut`, read at the lift's flat shift on top of the placement's: one gather pass instead of lift + barrier + place.
// CUT / LIFT: whether `cut` is given / the lift is fused — compile-time facts of every call site but one, so that the tests leave the
// per-word loops (a uniform branch costs two scalar instructions every time it is reached).
template <bool CUT, bool LIFT, class X, class R>
ARCLE_BIG_DEV void place_t(const X& x, const R& r, const int8_t* bg, const int8_t* cut, const int8_t* O, const int8_t* Q, int lift_delta) {
  const int W = x.W;
  const int px = r[ARCLE_REC_OBJECT_POS], py = r[ARCLE_REC_OBJECT_POS + 1];
  const int h = r[ARCLE_REC_OBJECT_DIM], w = r[ARCLE_REC_OBJECT_DIM + 1];
  const int gh = r[ARCLE_REC_GRID_DIM], gw = r[ARCLE_REC_GRID_DIM + 1];
  const int xh = i8w(px + h), yw = i8w(py + w);  // int8 + int8
  const bool draw = xh > 0 && px < gh && yw > 0 && py < gw;
  const int stx = imax(px, 0), edx = imin(gh, xh), sty = imax(py, 0), edy = imin(gw, yw);
  if (x.wide()) {  // whole chunks: the object tile read at the flat shift -(px * W + py), the destination rectangle as a byte mask
    const int d2 = px * W + py;
    BIG_EACH_CHUNK(x, c) {
      const Chunk bgc = ldg(bg, c);
      Chunk pv, qv;
      if (LIFT) {
        const Chunk sv = shifted16(cut, 16 * c - d2 + lift_delta, x.PS);
        pv = shifted16(bg, 16 * c - d2 + lift_delta, x.PS);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t m = x.sel_pos(sv.w[q]);  // object.py:78 sel > 0
          pv.w[q] &= m;                           // :81
          qv.w[q] = 0x01010101u & m;              // :84
        }
      } else {
        pv = shifted16(O, 16 * c - d2, x.PS);
        qv = shifted16(Q, 16 * c - d2, x.PS);
      }
      const Chunk in = draw ? rect_mask16(c, W, x.wm, stx, edx, sty, edy) : zero_chunk();
      Chunk grid, sel;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        uint32_t b = bgc.w[q];
        if (CUT) b &= ~x.sel_pos(reinterpret_cast<const uint32_t*>(cut)[4 * c + q]);  // background = where(sel > 0, 0, grid)
        const uint32_t m = in.w[q] & pos_bytes(pv.w[q]);                                // :138 where=(p > 0)
        grid.w[q] = (pv.w[q] & m) | (b & ~m);
        sel.w[q] = qv.w[q] & in.w[q];                                                   // :165
      }
      x.gs(ARCLE_PL_GRID, c, grid);
      x.gs(ARCLE_PL_SELECTED, c, sel);
    }
    return;
  }
  BIG_EACH_CHUNK(x, c) {
    Chunk sel = zero_chunk();
    const Chunk bgc = ldg(bg, c);
    const Chunk cutc = CUT ? ldg(cut, c) : zero_chunk();
    const Chunk grid = build_chunk(c, W, x.wm, x.P, [&](int f, int i, int j) {
      const int k = f & 15;
      const bool in = draw && i >= stx && i < edx && j >= sty && j < edy;
      const int t = in ? (i - px) * W + (j - py) : 0;
      const int8_t pv = O[t], qv = Q[t];  // (unconditional reads at a clamped index, see the lift)
      int8_t gv = cutc.b[k] > 0 ? (int8_t)0 : bgc.b[k];
      if (in && pv > 0) gv = pv;              // :138 where=(p > 0)
      sel.b[k] = in ? qv : (int8_t)0;         // :165
      return gv;
    });
    x.gs(ARCLE_PL_GRID, c, grid);
    x.gs(ARCLE_PL_SELECTED, c, sel);
  }
}

template <class X, class R>
ARCLE_BIG_DEV void place(const X& x, const R& r, const int8_t* bg, const int8_t* cut, const int8_t* O, const int8_t* Q) {
  if (cut) place_t<true, false>(x, r, bg, cut, O, Q, -1);
  else place_t<false, false>(x, r, bg, nullptr, O, Q, -1);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// one step() of one env: O2ARCv2Env.step (o2arcenv.py:130-147) / ARCEnv.step (arcenv.py:155-172) / RawARCEnv.step (arcenv.py:60-76)
// ------------------------------------------------------------------------------------------------------------------------------------
// X = CtxT<CPT, LEAN> (what the launch guarantees, see there); ING: ING_T_ANY = the ingress form is p.ingress whatever it is, ING_T_MASKS =
// one of the mask forms (int8 / bit-packed), ING_T_TUPLES = one of the tuple forms (bbox / point / bbox5) — the other family's code is not
// compiled —, ING_T_EXACT + f = exactly the form f (enum arcle_ingress): the form's tests and the other forms' loads fold away too.
enum { ING_T_ANY = -1, ING_T_MASKS = 0, ING_T_TUPLES = 1, ING_T_EXACT = 16 };
// diagnostic builds (-DARCLE_BIG_STOP_AT=k, tools/gpu_r6k.sh): the workgroup leaves after stage k of the step — 1 the env's scalars are in,
// 2 the selection is built and the op's plane staged, 3 the geometry is known, 4 the op has run — so that PMC passes count the instructions
// per stage (results are meaningless; never defined in the product build)
// (-DARCLE_BIG_EXP_S=n / -DARCLE_BIG_EXP_V=n: n dependent scalar / vector adds injected into every wave — what one more instruction of
// either kind costs a launch, profiles/round6_experiments.txt §2h)
#if defined(ARCLE_BIG_EXP_S) || defined(ARCLE_BIG_EXP_V)
#define BIG_INJECT()                                                                    \
  do {                                                                                  \
    int es_ = opi, ev_ = tid;                                                           \
    _Pragma("unroll") for (int k_ = 0; k_ < BIG_EXP_S_N; k_++) asm volatile("s_add_i32 %0, %0, 1" : "+s"(es_)); \
    _Pragma("unroll") for (int k_ = 0; k_ < BIG_EXP_V_N; k_++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(ev_)); \
    if (es_ == 0x7fffffff || ev_ == 0x7fffffff) st |= 1u << 31;                         \
  } while (0)
#ifdef ARCLE_BIG_EXP_S
enum { BIG_EXP_S_N = ARCLE_BIG_EXP_S };
#else
enum { BIG_EXP_S_N = 0 };
#endif
#ifdef ARCLE_BIG_EXP_V
enum { BIG_EXP_V_N = ARCLE_BIG_EXP_V };
#else
enum { BIG_EXP_V_N = 0 };
#endif
#else
#define BIG_INJECT() \
  do {               \
  } while (0)
#endif
#ifdef ARCLE_BIG_STOP_AT
#define BIG_STOP(k)                 \
  do {                              \
    if (ARCLE_BIG_STOP_AT == (k)) { \
      if (tid == 0) p.reward[env] = (int)st + cnt0 + opi; \
      return;                       \
    }                               \
  } while (0)
#else
#define BIG_STOP(k) \
  do {              \
  } while (0)
#endif
// FL >= 0: the launch's flag set is exactly FL (the launcher checked): every flag test folds — FL = AUTORESET | ELIDE_SELECTED is what
// ARCVecEnv(autoreset=True) and the benchmark step with.
template <class X, int ING, int FL = -1>
ARCLE_BIG_DEV void step_env_t(const BigParams& p, const int env, int8_t* lds) {
  X x(p, env, lds);
  const int ingress = ING >= ING_T_EXACT ? ING - ING_T_EXACT : p.ingress;
  constexpr bool TUPLES = ING == ING_T_TUPLES || ING == ING_T_EXACT + ING_BBOX || ING == ING_T_EXACT + ING_POINT || ING == ING_T_EXACT + ING_BBOX5;
  constexpr bool MASKS = ING == ING_T_MASKS || ING == ING_T_EXACT + ING_MASK || ING == ING_T_EXACT + ING_BITS;
  x.sel01 = TUPLES;  // (rectangles and points are written to S as 0 / 1)
  const int tid = x.tid, H = x.H, W = x.W, P = x.P, nch = x.nch;
  Rec16 r;
  {
    const Chunk rc = ldg(p.rec, env);
#pragma unroll
    for (int k = 0; k < 4; k++) r.w[k] = rc.w[k];
  }
  int cnt0 = p.cnt[2 * (size_t)env], cnt1 = p.cnt[2 * (size_t)env + 1];
  const uint32_t flags = FL >= 0 ? (uint32_t)FL : X::LEAN ? (p.flags & (uint32_t)LEAN_FLAGS) : p.flags;
  const bool mask_ingress = ING == ING_T_ANY ? (ingress == ING_MASK || ingress == ING_BITS) : MASKS;
  const bool scratch_rows = !X::LEAN && p.res_rec != nullptr;  // arcle_transition_rows: the envs are scratch envs, one per row
  const bool accounting = !X::LEAN && p.acct != nullptr;
  int reward = 0, submit_inc = 0;
  uint32_t st = 0;
  bool counted = false;  // the step happened: action_steps += 1
  int opi = 0;
  // ---- the action's scalars ----
  int pay[5] = {0, 0, 0, 0, 0};
  if (mask_ingress) {
  } else if (ingress == ING_BBOX) {
    for (int k = 0; k < 4; k++) pay[k] = reinterpret_cast<const int32_t*>(p.sel)[4 * (size_t)env + k];
  } else if (ingress == ING_POINT) {
    for (int k = 0; k < 2; k++) pay[k] = reinterpret_cast<const int32_t*>(p.sel)[2 * (size_t)env + k];
  } else if (ingress == ING_BBOX5) {
    for (int k = 0; k < 5; k++) pay[k] = reinterpret_cast<const int32_t*>(p.sel)[5 * (size_t)env + k];
  }
  // (workgroup-uniform by construction; telling the compiler so turns the dependent op-table read into a scalar load)
  opi = bx::uniform(!mask_ingress && ingress == ING_BBOX5 ? pay[4] : p.op[env]);
  if (!TUPLES && tid == 0) {  // (tuple selections reduce by arithmetic; grid == answer, the fill and the dense pair clear their own slots)
    Red* q = x.red;
    q->any_nz = q->any_pos = q->sum = 0;
    q->amax = 0u;
    q->x0 = q->y0 = 1 << 20;
    q->x1 = q->y1 = -1;
    q->neq = 0;
  }
  bx::sync();  // every thread holds the record / counters; the reduction block is clear
  BIG_INJECT();
  BIG_STOP(1);

  do {
    if (scratch_rows) {  // arcle_transition_rows: a row whose src_env names no resident env is passed through untouched
      const int src = p.src_env ? p.src_env[env] : env;
      if (src < 0 || src >= p.n_resident) {
        st |= ARCLE_ST_BAD_TASK;
        break;
      }
    }
    if (flags & (ARCLE_STEP_AUTORESET | ARCLE_STEP_RESAMPLE)) {
      // next-step autoreset (see ARCLE_STEP_AUTORESET): an env whose episode ended is re-initialised instead of executing the action
      const bool ended = r[ARCLE_REC_TERMINATED] != 0 || ((flags & ARCLE_STEP_TRUNCATE) && cnt0 >= p.step_limit);
      if (ended) {
        if (flags & ARCLE_STEP_RESAMPLE) {  // ... on a new task drawn on the device, augmented as the sampler says
          const uint32_t ep = (uint32_t)p.episode[env];
          const Draw d = draw_task(p, env, ep);
          load_task(x, r, d.entry, d.rot_k, d.perm, true);
          bx::sync();  // (every thread has read episode[env])
          if (tid == 0) {
            p.episode[env] = (int32_t)(ep + 1u);
            if (p.cur_task) p.cur_task[env] = d.entry;
          }
        } else {
          init_planes(x, x.g(ARCLE_PL_INPUT), false);
        }
        init_rec(r, p.max_trial);
        cnt0 = cnt1 = 0;
        break;
      }
    }
    // an index past the table reads slot n_ops, which is always empty
    const uint32_t slot = (uint32_t)opi < (uint32_t)p.n_ops ? (uint32_t)opi : (uint32_t)p.n_ops;
    const uint32_t desc = bx::sload32(p.d_ops, slot);  // (never written by a kernel: a scalar load, whatever the step stored before it)
    const int kind = (int)ARCLE_OP_KIND(desc), arg = (int)ARCLE_OP_ARG(desc);
    const uint32_t oflags = ARCLE_OP_FLAGS(desc);
    if (kind == ARCLE_OP_NONE) {  // reference: IndexError / TypeError before any mutation
      st |= ARCLE_ST_BAD_OP;
      break;
    }

    // ---- the plane the op will gather from, requested into A NOW: its round trip overlaps the selection's (one dependent memory phase
    // fewer per step).  For an object op under mask ingress whether the selection is fresh is only known after the reduction: the grid
    // is the guess (a continuing object restages its background below); tuple selections know at once. ----
    int staged = -1;  // the plane A holds (arcle_plane) once the ingest barrier has passed
    bool staged_obj = false;  // ... and B / C hold the stored object / object_sel
    {
      const bool tuple = !mask_ingress;
      bool tuple_any = false;
      if (tuple && ingress == ING_POINT) tuple_any = (uint32_t)pay[0] < (uint32_t)H && (uint32_t)pay[1] < (uint32_t)W;
      else if (tuple) tuple_any = (uint32_t)imin(pay[0], pay[2]) < (uint32_t)H && (uint32_t)imin(pay[1], pay[3]) < (uint32_t)W;
      const bool will_be_active = (oflags & ARCLE_OPF_RESET_SEL) ? false : r[ARCLE_REC_ACTIVE] != 0;
      switch (kind) {  // (a compare tree the compiler keeps as BRANCHES: Color and the critical ops leave after two or three compares; written
                       // as a table look-up + conditional assignments it became ~60 scalar selects for every op, round6 §2h)
        case ARCLE_OP_FLOODFILL:  // (a tuple that is not a single cell fills nothing, color.py:92: no plane is needed)
          if (!tuple || (tuple_any && (ingress == ING_POINT || (imin(imax(pay[0], pay[2]), H - 1) == imin(pay[0], pay[2]) && imin(imax(pay[1], pay[3]), W - 1) == imin(pay[1], pay[3])))))
            staged = ARCLE_PL_GRID;
          break;
        case ARCLE_OP_CROP_GRID: staged = ARCLE_PL_GRID; break;
        case ARCLE_OP_COPY: staged = arg ? ARCLE_PL_GRID : ARCLE_PL_INPUT; break;
        case ARCLE_OP_PASTE: staged = ARCLE_PL_CLIP; break;
        case ARCLE_OP_MOVE:
        case ARCLE_OP_ROTATE:
        case ARCLE_OP_FLIP:
          if (!tuple || tuple_any) staged = ARCLE_PL_GRID;
          else if (will_be_active) {
            staged = ARCLE_PL_BACKGROUND;
            staged_obj = true;
          }
          break;
        default: break;
      }
      if (staged >= 0) x.stage_g(x.A, staged);
      if (staged_obj) {
        x.stage_g(x.B, ARCLE_PL_OBJECT);
        x.stage_g(x.C, ARCLE_PL_OBJECT_SEL);
      }
    }

    // ---- selection -> S (bytes in LDS) + its reductions -------------------------------------------------------------------------
    bool any_nz, any_pos;
    int ssum, x0, x1, y0, y1, amax_cell;
    if (mask_ingress) {
      const bool packed = ingress == ING_BITS;  // boolean masks, bit f of the env's row of PS / 8 bytes = cell f
      const int8_t* const src = reinterpret_cast<const int8_t*>(p.sel) + (size_t)env * (size_t)(packed ? x.PS >> 3 : P);
      const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
      int l_nz = 0, l_pos = 0, l_sum = 0, lx0 = 1 << 20, lx1 = -1, ly0 = 1 << 20, ly1 = -1;
      uint32_t l_amax = 0;
      BIG_EACH_CHUNK(x, c) {
        Chunk v = zero_chunk();
        const int f0 = 16 * c;
        x.count();  // (the mask chunk: 16 bytes, or 2 of a bit row — counted as a chunk)
        if (packed) {
          uint32_t m = (uint32_t)(uint8_t)src[2 * c] | ((uint32_t)(uint8_t)src[2 * c + 1] << 8);
          m &= (1u << imin(imax(P - f0, 0), 16)) - 1u;  // (bits behind the plane's last cell are not the caller's to set)
#pragma unroll
          for (int q = 0; q < 4; q++)  // four bits -> four 0 / 1 bytes: bit i of the nibble lands on bit 8 i of n + (n << 7) + (n << 14) + (n << 21)
            v.w[q] = bx::mul24((m >> (4 * q)) & 15u, 0x00204081u) & 0x01010101u;
        } else if (f0 + 16 <= P) {
          v = aligned ? ldg(src, c) : ldu(src + f0);
        } else {
#pragma unroll
          for (int k = 0; k < 16; k++)
            if (f0 + k < P) v.b[k] = src[f0 + k];
        }
        stg(x.S, c, v);
        const uint32_t any_bits = v.w[0] | v.w[1] | v.w[2] | v.w[3];
        if (any_bits && x.wide()) {
          // the chunk's share of the reductions on whole words (W >= 16: its cells lie in at most two plane rows): which bytes are non-zero as
          // a 16-bit map (multiply-gather of the bytes' low bits), rows / first and last column of either part by count-zeros, the sum by
          // a dot product with ones, the arg-max as "first 1" for a boolean chunk — the general chain (~12 instructions a cell) only
          // for chunks that hold other values
          uint32_t m16 = 0, posw = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            m16 |= ((bx::mul32(nz_bytes(v.w[q]) & 0x01010101u, 0x01020408u) >> 24) & 15u) << (4 * q);
            posw |= pos_bytes(v.w[q]);
            l_sum = bx::dot4_i8(v.w[q], l_sum);
          }
          const int i0 = div_w(f0, x.wm), j0 = f0 - i0 * W, n0 = imin(16, W - j0);
          const uint32_t part0 = m16 & ((1u << n0) - 1u), part1 = m16 >> n0;
          l_nz = 1;
          l_pos |= posw != 0u;
          if (part0) {
            lx0 = imin(lx0, i0);
            lx1 = imax(lx1, i0);
            ly0 = imin(ly0, j0 + __builtin_ctz(part0));
            ly1 = imax(ly1, j0 + 31 - __builtin_clz(part0));
          }
          if (part1) {
            lx0 = imin(lx0, i0 + 1);
            lx1 = imax(lx1, i0 + 1);
            ly0 = imin(ly0, __builtin_ctz(part1));
            ly1 = imax(ly1, 31 - __builtin_clz(part1));
          }
          if (!(any_bits & 0xfefefefeu)) {  // only 0 / 1: np.argmax = the first 1 (cells behind P hold 0)
            const uint32_t key = (129u << 16) | (uint32_t)(0xffff - (f0 + __builtin_ctz(m16)));
            if (key > l_amax) l_amax = key;
          } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
              const uint32_t key = ((uint32_t)(v.b[k] + 128) << 16) | (uint32_t)(0xffff - (f0 + k));
              if (f0 + k < P && key > l_amax) l_amax = key;
            }
          }
        } else if (any_bits) {
          int i = div_w(f0, x.wm), j = f0 - i * W;
#pragma unroll
          for (int k = 0; k < 16; k++) {
            const int s = v.b[k];
            if (s != 0) {
              l_nz = 1;
              l_pos |= s > 0;
              lx0 = imin(lx0, i);
              lx1 = imax(lx1, i);
              ly0 = imin(ly0, j);
              ly1 = imax(ly1, j);
            }
            l_sum += s;
            const uint32_t key = ((uint32_t)(s + 128) << 16) | (uint32_t)(0xffff - (f0 + k));
            if (f0 + k < P && key > l_amax) l_amax = key;
            if (++j == W) {
              j = 0;
              ++i;
            }
          }
        } else {
          const uint32_t key = (128u << 16) | (uint32_t)(0xffff - f0);  // a chunk of zeros: its first cell stands for it in the arg-max
          if (f0 < P && key > l_amax) l_amax = key;
        }
      }
      Red* q = x.red;
      if (bx::HAS_WAVE_OPS) {
        // Every wavefront reduces its lanes' shares across lanes — two ballots, six DPP reductions (row shifts 1 / 2 / 4 / 8, then the row
        // broadcasts 15 and 31: lane 63 holds the result) — and lane 0 leaves the eight words in the wavefront's slot; ONE barrier (which
        // also publishes the S tile and the staged plane), then every thread combines the slots of the workgroup's wavefronts (one: none
        // to combine).  No LDS atomics: the compiler's rewrite of a same-address atomic costs ~25 instructions each (its default, a
        // scalar loop over the lanes, ~8 per lane), profiles/round6_experiments.txt §2f / §2i.
        uint32_t* const slots = reinterpret_cast<uint32_t*>(x.sc);  // (the row writers' scalar / layout area: idle until a step's epilogue)
        const uint32_t w_fl = (bx::wave_any(l_nz != 0) ? 1u : 0u) | (bx::wave_any(l_pos != 0) ? 2u : 0u);
        const int w_sum = bx::wave_add(l_sum), w_x0 = bx::wave_min(lx0), w_x1 = bx::wave_max(lx1), w_y0 = bx::wave_min(ly0), w_y1 = bx::wave_max(ly1);
        const uint32_t w_amax = bx::wave_umax(l_amax);
        const int nw = x.NT >> 6;
        if (nw > 1 && (tid & 63) == 0) {
          uint32_t* const sl = slots + 8 * (tid >> 6);
          sl[0] = w_fl;
          sl[1] = (uint32_t)w_sum;
          sl[2] = (uint32_t)w_x0;
          sl[3] = (uint32_t)w_x1;
          sl[4] = (uint32_t)w_y0;
          sl[5] = (uint32_t)w_y1;
          sl[6] = w_amax;
        }
        bx::sync();
        uint32_t fl = w_fl, am = w_amax;
        ssum = w_sum;
        x0 = w_x0;
        x1 = w_x1;
        y0 = w_y0;
        y1 = w_y1;
        if (nw > 1) {
          fl = 0u;
          am = 0u;
          ssum = 0;
          x0 = y0 = 1 << 20;
          x1 = y1 = -1;
          for (int w = 0; w < nw; w++) {
            const uint32_t* const sl = slots + 8 * w;
            fl |= sl[0];
            ssum += (int)sl[1];
            x0 = imin(x0, (int)sl[2]);
            x1 = imax(x1, (int)sl[3]);
            y0 = imin(y0, (int)sl[4]);
            y1 = imax(y1, (int)sl[5]);
            am = sl[6] > am ? sl[6] : am;
          }
          fl = (uint32_t)bx::uniform((int)fl);
          am = (uint32_t)bx::uniform((int)am);
          ssum = bx::uniform(ssum);
          x0 = bx::uniform(x0);
          x1 = bx::uniform(x1);
          y0 = bx::uniform(y0);
          y1 = bx::uniform(y1);
        }
        any_nz = (fl & 1u) != 0;
        any_pos = (fl & 2u) != 0;
        amax_cell = 0xffff - (int)(am & 0xffffu);
      } else {  // (the CPU emulation of the workgroup: host threads are no wavefront — atomics on the reduction block)
        if (l_nz) {
          bx::lds_or(&q->any_nz, 1);
          if (l_pos) bx::lds_or(&q->any_pos, 1);
          bx::lds_min(&q->x0, lx0);
          bx::lds_max(&q->x1, lx1);
          bx::lds_min(&q->y0, ly0);
          bx::lds_max(&q->y1, ly1);
        }
        if (l_sum) bx::lds_add(&q->sum, l_sum);
        bx::lds_umax(&q->amax, l_amax);
        bx::sync();
        any_nz = q->any_nz != 0;
        any_pos = q->any_pos != 0;
        ssum = q->sum;
        x0 = q->x0;
        x1 = q->x1;
        y0 = q->y0;
        y1 = q->y1;
        amax_cell = 0xffff - (int)(q->amax & 0xffffu);
      }
      if ((flags & ARCLE_STEP_CONTINUE_RULE) && (kind == ARCLE_OP_MOVE || kind == ARCLE_OP_ROTATE || kind == ARCLE_OP_FLIP)) {
        // the O2ARC trace harness (tests/o2arc_check.py:169-170): an object op whose logged selection equals the env's current
        // `selected` plane continues the active object, i.e. is sent with an empty selection
        bool differs = false;
        BIG_EACH_CHUNK(x, c) {
          const Chunk a = x.gl(ARCLE_PL_SELECTED, c), b = ldg(x.S, c);
          differs |= ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3])) != 0;
        }
        if (differs) q->neq = 1;
        bx::sync();
        const bool same = q->neq == 0;
        bx::sync();
        if (tid == 0) q->neq = 0;
        if (same) {
          x.fill(x.S, zero_chunk());
          any_nz = any_pos = false;
          ssum = 0;
          amax_cell = 0;
        }
        bx::sync();
      }
    } else {
      // BBoxWrapper.action (bbox.py:22-30): sorted corners, slices clip at H, W / PointWrapper.action (bbox.py:43-49).  Coordinates
      // outside the wrappers' action space select nothing and raise ARCLE_ST_BAD_SELECTION.
      int xa, xb, ya, yb;
      bool any;
      if (ingress == ING_POINT) {
        xa = xb = pay[0];
        ya = yb = pay[1];
        any = (uint32_t)xa < (uint32_t)H && (uint32_t)ya < (uint32_t)W;
        if (!any) st |= ARCLE_ST_BAD_SELECTION;
      } else {
        xa = imin(pay[0], pay[2]);
        xb = imin(imax(pay[0], pay[2]), H - 1);
        ya = imin(pay[1], pay[3]);
        yb = imin(imax(pay[1], pay[3]), W - 1);
        any = (uint32_t)xa < (uint32_t)H && (uint32_t)ya < (uint32_t)W;
        if (!any && (xa | ya) < 0) st |= ARCLE_ST_BAD_SELECTION;
      }
      if (x.wide()) {
        BIG_EACH_CHUNK(x, c) {
          Chunk m = any ? rect_mask16(c, W, x.wm, xa, xb + 1, ya, yb + 1) : zero_chunk();
#pragma unroll
          for (int q = 0; q < 4; q++) m.w[q] &= 0x01010101u;
          stg(x.S, c, m);
        }
      } else
      BIG_EACH_CHUNK(x, c)
        stg(x.S, c, build_chunk(c, W, x.wm, P, [&](int, int i, int j) { return (any && i >= xa && i <= xb && j >= ya && j <= yb) ? 1 : 0; }));
      any_nz = any_pos = any;
      ssum = any ? (xb - xa + 1) * (yb - ya + 1) : 0;
      x0 = xa;
      x1 = xb;
      y0 = ya;
      y1 = yb;
      amax_cell = any ? xa * W + ya : 0;
      bx::sync();
    }

    BIG_STOP(2);
    // ---- reset_sel / keep_sel (object.py:10-41): scalar part first — an object op reads `active` after the wrapper ran ----
    const int8_t active_before = r[ARCLE_REC_ACTIVE];
    if (oflags & ARCLE_OPF_RESET_SEL) r[ARCLE_REC_ACTIVE] = 0;

    // ---- Rotate / Flip: the geometry first, so that a transform the reference raises in skips the whole step ----
    bool fresh = false, obj_go = false, new_geom = false;
    int oh = 0, ow = 0, nx = 0, ny = 0, nh = 0, nw = 0, npar = 0, ai = 0, bj = 0, c0 = 0;
    if (kind == ARCLE_OP_MOVE || kind == ARCLE_OP_ROTATE || kind == ARCLE_OP_FLIP) {
      // _init_objsel, object.py:60-111: a fresh selection, or the stored object when active, or a total no-op (:110-111)
      int ox, oy, bx0, bx1, by0, by1;
      fresh = any_nz;
      if (fresh) {
        bx0 = x0; bx1 = x1; by0 = y0; by1 = y1;
        oh = x1 - x0 + 1; ow = y1 - y0 + 1; ox = x0; oy = y0;
        obj_go = true;
      } else if (r[ARCLE_REC_ACTIVE]) {
        oh = r[ARCLE_REC_OBJECT_DIM]; ow = r[ARCLE_REC_OBJECT_DIM + 1];
        ox = r[ARCLE_REC_OBJECT_POS]; oy = r[ARCLE_REC_OBJECT_POS + 1];
        bx0 = ox; bx1 = i8w(i8w(ox + oh) - 1); by0 = oy; by1 = i8w(i8w(oy + ow) - 1);  // :102-107, int8 arithmetic
        obj_go = true;
      } else {
        ox = oy = bx0 = bx1 = by0 = by1 = 0;
      }
      nx = ox; ny = oy; nh = oh; nw = ow;
      npar = fresh ? 0 : r[ARCLE_REC_PARITY];
      bool domain_error = false;
      if (obj_go && kind == ARCLE_OP_ROTATE) {  // gen_rotate(k), object.py:177-213 — the float centre arithmetic on doubled integers
        const int k = arg;
        if (k & 1) {
          const int sx2 = fresh ? bx1 + bx0 : i8w(bx1 + bx0);
          const int sy2 = fresh ? by1 + by0 : i8w(by1 + by0);
          if ((oh & 1) == (ow & 1)) {
            nx = floordiv2(sx2 - sy2 + 2 * oy);
            ny = floordiv2(sy2 - sx2 + 2 * ox);
          } else {
            npar = (npar + k) % 2;
            const int sig = (k + 2) % 4 - 2, mod = 1 - npar;
            nx = floordiv2(sx2 + imin(sig * (sy2 - 2 * by0), sig * (sy2 - 2 * by1)) + 2 * mod);
            ny = floordiv2(sy2 + imin(-sig * (sx2 - 2 * bx0), -sig * (sx2 - 2 * bx1)) + 2 * mod);
          }
          nh = ow;
          nw = oh;
          new_geom = true;
          if (ow > H || oh > W || nx < -128 || nx > 127 || ny < -128 || ny > 127) domain_error = true;
        }
        if (k == 1) { ai = -1; bj = W; c0 = ow - 1; }                          // rot90:  new[i,j] = old[j, w-1-i]
        else if (k == 2) { ai = -W; bj = -1; c0 = (oh - 1) * W + (ow - 1); }   // rot180: old[h-1-i, w-1-j]
        else { ai = 1; bj = -W; c0 = (oh - 1) * W; }                           // rot270: old[h-1-j, i]
      } else if (obj_go && kind == ARCLE_OP_FLIP) {  // gen_flip(axis), object.py:265-276; object_dim is NOT updated (:270-273)
        if (arg == 0) { ai = W; bj = -1; c0 = ow - 1; }                        // fliplr: old[i, w-1-j]
        else if (arg == 1) { ai = -W; bj = 1; c0 = (oh - 1) * W; }             // flipud: old[h-1-i, j]
        else if (arg == 2) { ai = 1; bj = W; c0 = 0; nh = ow; nw = oh; }       // D0 transpose: old[j, i]
        else { ai = -1; bj = -W; c0 = (oh - 1) * W + (ow - 1); nh = ow; nw = oh; }  // D1: old[h-1-j, w-1-i]
        if (arg >= 2 && (ow > H || oh > W)) domain_error = true;
      }
      if (domain_error) {  // the reference raised inside the op (ValueError at object.py:45 / int8 overflow): the step did not happen
        st |= ARCLE_ST_ROTATE_DOMAIN;
        r[ARCLE_REC_ACTIVE] = active_before;
        break;
      }
    }

    // ---- the wrappers' plane writes: reset_sel / keep_sel set `selected` BEFORE the wrapped op runs and an object op that places its
    // object overwrites it — so the plane is written once, behind the op, with whichever value is final (1 = zeros, 2 = the selection;
    // place() clears the request).  No op below stores a plane ahead of its last barrier (a barrier also waits for the stores issued before
    // it): gathers go to LDS first, the stores come at the end — neutral in time on MI355X (profiles/round5_experiments.txt §19), kept for
    // the single write of `selected`. ----
    int sel_pending = 0;
    if (oflags & ARCLE_OPF_KEEP_SEL) sel_pending = 2;
    else if ((oflags & ARCLE_OPF_RESET_SEL) && !((flags & ARCLE_STEP_ELIDE_SELECTED) && active_before == 0)) sel_pending = 1;
    // (the only ops that recycle the S tile — Rotate / Flip that proceed — end in place(): a pending keep_sel never needs S after them)

    BIG_STOP(3);
    int eq = -1;  // grid == answer, evaluated at most once
    bool grid_moved = true;  // (conservative: every op below that stores the grid plane ends with a barrier before the compare)
    switch (kind) {  // transition(): self.operations[op](state, action)   o2arcenv.py:149-151
      case ARCLE_OP_COLOR: {  // color.py:70-74 — whole H x W plane, grid_dim ignored
        if (!any_nz) break;
        BIG_EACH_CHUNK(x, c) {
          const Chunk s = ldg(x.S, c);
          if (!(s.w[0] | s.w[1] | s.w[2] | s.w[3])) continue;
          Chunk gr = x.gl(ARCLE_PL_GRID, c);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t m = x.sel_nz(s.w[q]);
            gr.w[q] = (gr.w[q] & ~m) | ((((uint32_t)arg & 0xffu) * 0x01010101u) & m);
          }
          x.gs(ARCLE_PL_GRID, c, gr);
        }
        break;
      }
      case ARCLE_OP_FLOODFILL: {  // color.py:88-100
        if (ssum != 1) break;
        const int sx = div_w(amax_cell, x.wm), sy = amax_cell - sx * W;
        const int gh = r[ARCLE_REC_GRID_DIM], gw = r[ARCLE_REC_GRID_DIM + 1];
        if (sx >= gh || sy >= gw) break;
        flood_fill(x, gh, gw, sx, sy, arg);  // (the grid is in A)
        break;
      }
      case ARCLE_OP_MOVE:
      case ARCLE_OP_ROTATE:
      case ARCLE_OP_FLIP: {
        if (!obj_go) break;
        int8_t *O = x.B, *Q = x.C;  // object / object_sel tiles
        bool bg_in_A = false;        // A holds the background itself (else: the grid, and the background is where(sel > 0, 0, grid))
        const bool transform = kind != ARCLE_OP_MOVE;
        if (fresh && x.wide()) {
          // A fresh selection, whole chunks (object.py:67-99 fused with the op): the lifted object is the grid under the selection at the flat
          // shift delta = x0 * W + y0, so nothing is lifted into tiles first.  Move: the three object planes are formed straight from A / S
          // and place() reads A / S at the combined shift — ONE gather pass.  Rotate / Flip: the transformed tiles are gathered straight
          // from A / S (source index + delta) — one gather pass and one barrier, then place().
          if (staged != ARCLE_PL_GRID) {
            x.stage_g(x.A, ARCLE_PL_GRID);
            bx::sync();
          }
          const int delta = x0 * W + y0;
          r[ARCLE_REC_OBJECT_DIM] = (int8_t)oh;
          r[ARCLE_REC_OBJECT_DIM + 1] = (int8_t)ow;
          r[ARCLE_REC_OBJECT_POS] = (int8_t)x0;
          r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)y0;
          r[ARCLE_REC_ACTIVE] = 1;
          r[ARCLE_REC_PARITY] = 0;
          if (!transform) {  // gen_move(d), object.py:230-240
            const int dx = (arg == 0) ? -1 : (arg == 1) ? 1 : 0;
            const int dy = (arg == 2) ? 1 : (arg == 3) ? -1 : 0;
            r[ARCLE_REC_OBJECT_POS] = (int8_t)i8w(x0 + dx);  // :238, int8 wrap
            r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)i8w(y0 + dy);
            BIG_EACH_CHUNK(x, c) {
              const Chunk sv = shifted16(x.S, 16 * c + delta, x.PS), av = shifted16(x.A, 16 * c + delta, x.PS);
              const Chunk in = rect_mask16(c, W, x.wm, 0, oh, 0, ow);
              Chunk ob, qs, gr = ldg(x.A, c);
              const Chunk sm = ldg(x.S, c);
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const uint32_t m = in.w[q] & x.sel_pos(sv.w[q]);  // :78 sel > 0
                ob.w[q] = av.w[q] & m;                            // :81
                qs.w[q] = 0x01010101u & m;                        // :84
                gr.w[q] &= ~x.sel_pos(sm.w[q]);                   // :87-88 background = where(sel > 0, 0, grid)
              }
              x.gs(ARCLE_PL_OBJECT, c, ob);
              x.gs(ARCLE_PL_OBJECT_SEL, c, qs);
              x.gs(ARCLE_PL_BACKGROUND, c, gr);
            }
            place_t<true, true>(x, r, x.A, x.S, nullptr, nullptr, delta);
          } else {
            BIG_EACH_CHUNK(x, c) {
              const Chunk gs_ = gather_affine16(x.S, c, W, x.wm, P, nh, nw, c0 + delta, ai, bj), ga = gather_affine16(x.A, c, W, x.wm, P, nh, nw, c0 + delta, ai, bj);
              Chunk ob, qs;
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const uint32_t m = x.sel_pos(gs_.w[q]);
                ob.w[q] = ga.w[q] & m;
                qs.w[q] = 0x01010101u & m;
              }
              stg(x.B, c, ob);
              stg(x.C, c, qs);
            }
            if (new_geom) {
              r[ARCLE_REC_OBJECT_POS] = (int8_t)nx;
              r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)ny;
              r[ARCLE_REC_OBJECT_DIM] = (int8_t)nh;
              r[ARCLE_REC_OBJECT_DIM + 1] = (int8_t)nw;
              r[ARCLE_REC_PARITY] = (int8_t)npar;
            }
            bx::sync();
            BIG_EACH_CHUNK(x, c) {
              Chunk gr = ldg(x.A, c);
              const Chunk sm = ldg(x.S, c);
#pragma unroll
              for (int q = 0; q < 4; q++) gr.w[q] &= ~x.sel_pos(sm.w[q]);
              x.gs(ARCLE_PL_BACKGROUND, c, gr);
              x.gs(ARCLE_PL_OBJECT, c, ldg(x.B, c));
              x.gs(ARCLE_PL_OBJECT_SEL, c, ldg(x.C, c));
            }
            place_t<true, false>(x, r, x.A, x.S, x.B, x.C, -1);
          }
          sel_pending = 0;  // (place() writes the whole `selected` plane)
          break;
        }
        if (fresh) {  // object.py:67-99 (W < 16: cell by cell, through object tiles)
          if (staged != ARCLE_PL_GRID) {
            x.stage_g(x.A, ARCLE_PL_GRID);
            bx::sync();
          }
          BIG_EACH_CHUNK(x, c) {
            Chunk qs = zero_chunk();
            // (every gather below reads LDS UNCONDITIONALLY at a clamped index and selects afterwards: the 16 cells' reads are then
            // independent and issue back to back — reads under a branch wait for one another, profiles/round5_experiments.txt §15)
            const Chunk ob = build_chunk(c, W, x.wm, P, [&](int f, int i, int j) {
              const bool in = i < oh && j < ow;
              const int s = in ? (x0 + i) * W + (y0 + j) : 0;
              const int8_t sv = x.S[s], av = x.A[s];
              const bool part = in && sv > 0;          // :78 sel > 0
              qs.b[f & 15] = part ? (int8_t)1 : (int8_t)0;  // :84
              return part ? av : (int8_t)0;            // :81
            });
            stg(x.B, c, ob);
            stg(x.C, c, qs);
          }
          r[ARCLE_REC_OBJECT_DIM] = (int8_t)oh;
          r[ARCLE_REC_OBJECT_DIM + 1] = (int8_t)ow;
          r[ARCLE_REC_OBJECT_POS] = (int8_t)x0;
          r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)y0;
          r[ARCLE_REC_ACTIVE] = 1;
          r[ARCLE_REC_PARITY] = 0;
          // (selected = sel, :96 — place() below rewrites the whole plane)
        } else {  // :102-107 the stored object continues (the selection is empty: S is all zero)
          if (!staged_obj) {
            if (staged >= 0) bx::sync();  // (a mask selection that turned out empty: every thread is done with the guessed plane)
            x.stage_g(x.A, ARCLE_PL_BACKGROUND);
            x.stage_g(x.B, ARCLE_PL_OBJECT);
            x.stage_g(x.C, ARCLE_PL_OBJECT_SEL);
          }
          bg_in_A = true;
        }
        bx::sync();
        if (!transform) {  // gen_move(d), object.py:230-240
          const int dx = (arg == 0) ? -1 : (arg == 1) ? 1 : 0;
          const int dy = (arg == 2) ? 1 : (arg == 3) ? -1 : 0;
          r[ARCLE_REC_OBJECT_POS] = (int8_t)i8w(r[ARCLE_REC_OBJECT_POS] + dx);  // :238, int8 wrap
          r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)i8w(r[ARCLE_REC_OBJECT_POS + 1] + dy);
          if (fresh) {  // the lifted tiles and the background go out now, behind the last barrier (this thread's own chunks)
            BIG_EACH_CHUNK(x, c) {
              x.gs(ARCLE_PL_OBJECT, c, ldg(x.B, c));
              x.gs(ARCLE_PL_OBJECT_SEL, c, ldg(x.C, c));
              Chunk gr = ldg(x.A, c);  // background = where(sel > 0, 0, grid)  :87-88; place() forms it again from A and S
              const Chunk sm = ldg(x.S, c);
#pragma unroll
              for (int q = 0; q < 4; q++) gr.w[q] &= ~x.sel_pos(sm.w[q]);
              x.gs(ARCLE_PL_BACKGROUND, c, gr);
            }
          }
        } else {
          // dst[:nh,:nw] = T(src[:h,:w]), rest 0 (_pad_assign, object.py:43-47): object B -> S, then object_sel C -> B.  A thread
          // overwrites S only at its OWN chunks, after it has formed the background of those chunks from them (fresh selections).
          BIG_EACH_CHUNK(x, c) {
            if (fresh) {  // background = where(sel > 0, 0, grid)  :87-88, into A in place (the lift is done with the grid)
              Chunk gr = ldg(x.A, c);
              const Chunk sm = ldg(x.S, c);
#pragma unroll
              for (int q = 0; q < 4; q++) gr.w[q] &= ~x.sel_pos(sm.w[q]);
              stg(x.A, c, gr);
            }
            const Chunk t = x.wide() ? gather_affine16(x.B, c, W, x.wm, P, nh, nw, c0, ai, bj) : build_chunk(c, W, x.wm, P, [&](int, int i, int j) {
              const bool in = i < nh && j < nw;
              const int8_t v = x.B[in ? c0 + ai * i + bj * j : 0];
              return in ? v : (int8_t)0;
            });
            stg(x.S, c, t);
          }
          bg_in_A = true;
          bx::sync();
          BIG_EACH_CHUNK(x, c) {
            const Chunk t = x.wide() ? gather_affine16(x.C, c, W, x.wm, P, nh, nw, c0, ai, bj) : build_chunk(c, W, x.wm, P, [&](int, int i, int j) {
              const bool in = i < nh && j < nw;
              const int8_t v = x.C[in ? c0 + ai * i + bj * j : 0];
              return in ? v : (int8_t)0;
            });
            stg(x.B, c, t);
          }
          O = x.S;
          Q = x.B;
          if (new_geom) {
            r[ARCLE_REC_OBJECT_POS] = (int8_t)nx;
            r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)ny;
            r[ARCLE_REC_OBJECT_DIM] = (int8_t)nh;
            r[ARCLE_REC_OBJECT_DIM + 1] = (int8_t)nw;
            r[ARCLE_REC_PARITY] = (int8_t)npar;
          }
          bx::sync();
          BIG_EACH_CHUNK(x, c) {  // the transformed tiles (and a fresh background) go out behind the last barrier
            if (fresh) x.gs(ARCLE_PL_BACKGROUND, c, ldg(x.A, c));
            x.gs(ARCLE_PL_OBJECT, c, ldg(x.S, c));
            x.gs(ARCLE_PL_OBJECT_SEL, c, ldg(x.B, c));
          }
        }
        sel_pending = 0;  // (place() writes the whole `selected` plane)
        place(x, r, x.A, bg_in_A ? nullptr : x.S, O, Q);
        break;
      }
      case ARCLE_OP_COPY: {  // gen_copy(source), object.py:291-312
        if (!any_pos) break;
        const int so = arg ? ARCLE_REC_GRID_DIM : ARCLE_REC_INPUT_DIM;
        const int ss_h = r[so], ss_w = r[so + 1];
        if (x1 > ss_h || y1 > ss_w) break;  // :301 (sic: > not >=)
        const int h = x1 - x0 + 1, w = y1 - y0 + 1;
        // (the source plane is in A)
        if (x.wide()) {
          BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_CLIP, c, cut_out16(x, c, x0, y0, h, w));  // :310-312 where=logical_and(src, sel)
        } else
        BIG_EACH_CHUNK(x, c)
          x.gs(ARCLE_PL_CLIP, c, build_chunk(c, W, x.wm, P, [&](int, int i, int j) {
                const bool in = i < h && j < w;
                const int s = in ? (x0 + i) * W + (y0 + j) : 0;
                const int8_t sv = x.S[s], av = x.A[s];
                return (in && sv != 0) ? av : (int8_t)0;  // :310-312 where=logical_and(src, sel)
              }));
        r[ARCLE_REC_CLIP_DIM] = (int8_t)h;
        r[ARCLE_REC_CLIP_DIM + 1] = (int8_t)w;
        break;
      }
      case ARCLE_OP_PASTE: {  // gen_paste(paste_blank), object.py:317-348
        if (!any_pos) break;
        const int h = r[ARCLE_REC_CLIP_DIM], w = r[ARCLE_REC_CLIP_DIM + 1];
        if (h == 0 || w == 0) break;  // :334
        const int ex = imin(x0 + h, H), ey = imin(y0 + w, W);  // :340-341 clipped to H x W, not grid_dim
        // (the clip plane is in A)
        const int c_first = (x0 * W) >> 4, c_last = imin(nch - 1, (ex * W) >> 4);
        if (x.wide()) {  // whole chunks: the clip read at the flat shift -(x0 * W + y0), the pasted rectangle as a byte mask
          const int d2 = x0 * W + y0;
          const uint32_t blank = arg ? ~0u : 0u;
          BIG_EACH_CHUNK(x, c) {
            if (c < c_first || c > c_last) continue;
            Chunk gr = x.gl(ARCLE_PL_GRID, c);
            const Chunk pv = shifted16(x.A, 16 * c - d2, x.PS), in = rect_mask16(c, W, x.wm, x0, ex, y0, ey);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const uint32_t m = in.w[q] & (pos_bytes(pv.w[q]) | blank);  // :345-348 (paste_blank: every cell of the rectangle)
              gr.w[q] = (pv.w[q] & m) | (gr.w[q] & ~m);
            }
            x.gs(ARCLE_PL_GRID, c, gr);
          }
        } else
        BIG_EACH_CHUNK(x, c) {
            if (c < c_first || c > c_last) continue;
          const Chunk gr = x.gl(ARCLE_PL_GRID, c);
          x.gs(ARCLE_PL_GRID, c, build_chunk(c, W, x.wm, P, [&](int f, int i, int j) {
                const bool in = i >= x0 && i < ex && j >= y0 && j < ey;
                const int8_t pv = x.A[in ? (i - x0) * W + (j - y0) : 0];
                return (in && (arg || pv > 0)) ? pv : gr.b[f & 15];  // :345-348
              }));
        }
        break;
      }
      case ARCLE_OP_COPY_FROM_INPUT: {  // critical.py:28-29
        BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_GRID, c, x.gl(ARCLE_PL_INPUT, c));
        r[ARCLE_REC_GRID_DIM] = r[ARCLE_REC_INPUT_DIM];
        r[ARCLE_REC_GRID_DIM + 1] = r[ARCLE_REC_INPUT_DIM + 1];
        break;
      }
      case ARCLE_OP_RESET_GRID: {  // critical.py:17
        BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_GRID, c, zero_chunk());
        break;
      }
      case ARCLE_OP_RESIZE_GRID: {  // critical.py:39-46
        if (!any_nz) break;
        BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_GRID, c, zero_chunk());
        r[ARCLE_REC_GRID_DIM] = (int8_t)(x1 - x0 + 1);
        r[ARCLE_REC_GRID_DIM + 1] = (int8_t)(y1 - y0 + 1);
        break;
      }
      case ARCLE_OP_CROP_GRID: {  // critical.py:56-66
        if (!any_nz) break;
        const int h = x1 - x0 + 1, w = y1 - y0 + 1;  // (the grid is in A)
        if (x.wide()) {
          BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_GRID, c, cut_out16(x, c, x0, y0, h, w));
        } else
        BIG_EACH_CHUNK(x, c)
          x.gs(ARCLE_PL_GRID, c, build_chunk(c, W, x.wm, P, [&](int, int i, int j) {
                const bool in = i < h && j < w;
                const int s = in ? (x0 + i) * W + (y0 + j) : 0;
                const int8_t sv = x.S[s], av = x.A[s];
                return (in && sv != 0) ? av : (int8_t)0;
              }));
        r[ARCLE_REC_GRID_DIM] = (int8_t)h;
        r[ARCLE_REC_GRID_DIM + 1] = (int8_t)w;
        break;
      }
      case ARCLE_OP_RESIZE_TO_ANSWER: {  // arcenv.py:31-35
        const int ah = r[ARCLE_REC_ANSWER_DIM], aw = r[ARCLE_REC_ANSWER_DIM + 1];
        r[ARCLE_REC_GRID_DIM] = (int8_t)ah;
        r[ARCLE_REC_GRID_DIM + 1] = (int8_t)aw;
        BIG_EACH_CHUNK(x, c) {
          const Chunk gr = x.gl(ARCLE_PL_GRID, c);
          x.gs(ARCLE_PL_GRID, c, build_chunk(c, W, x.wm, P, [&](int f, int i, int j) { return (i < ah && j < aw) ? gr.b[f & 15] : (int8_t)0; }));
        }
        break;
      }
      case ARCLE_OP_SUBMIT: {  // base.py:172-183
        grid_moved = false;
        int trials = r[ARCLE_REC_TRIALS];
        if (trials != 0) {
          trials = i8w(trials - 1);  // :174 int8 wrap
          r[ARCLE_REC_TRIALS] = (int8_t)trials;
          submit_inc = 1;
          if (flags & ARCLE_STEP_RESET_ON_SUBMIT) {
            // base.py:179-180: init_state() rebinds current_state inside submit — the decrement, the `terminated` of a correct
            // answer and the trials-exhausted check all land on the discarded dict; the caller sees the
            // re-initialised state, reward() is evaluated on it, and the env's counters go on
            init_planes(x, x.g(ARCLE_PL_INPUT), false);
            init_rec(r, p.max_trial);
            grid_moved = true;
            break;
          }
          eq = grid_equals_answer(x, r) ? 1 : 0;
          if (eq) r[ARCLE_REC_TERMINATED] = 1;
        }
        if (trials == 0) r[ARCLE_REC_TERMINATED] = 1;
        break;
      }
      default:  // ARCLE_OP_HOST: a device no-op, the step is counted
        break;
    }
    if (sel_pending == 2) {
      BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_SELECTED, c, ldg(x.S, c));
    } else if (sel_pending == 1) {
      BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_SELECTED, c, zero_chunk());
    }

    BIG_STOP(4);
    // reward(): only the LAST op of the table can be rewarded (o2arcenv.py:121-128)
    if (opi == p.n_ops - 1) {
      if (eq < 0) {
        if (grid_moved) bx::sync();  // the grid plane this workgroup just stored
        eq = grid_equals_answer(x, r) ? 1 : 0;
      }
      reward = eq;
    }
    counted = true;
  } while (0);

  if (counted) {
    cnt0 += 1;  // o2arcenv.py:142
    cnt1 += submit_inc;
  }
  if ((flags & ARCLE_STEP_DENSE) && p.dense) {
    // the research env's dense reward (agents/env.py:44-58) as an exact integer pair (correct cells, total cells) of the state the step
    // produced; (0, 0) = "no dense term" for a step that executed no action (the auto-reset step of an env, a skipped step).  Computed
    // from the planes every time (the one-wavefront kernels keep a per-env cache of the pair; here the compare is one pass of a phase-bound
    // kernel).
    int correct = 0, total = 0;
    if (counted) {
      const int gh = r[ARCLE_REC_GRID_DIM], gw = r[ARCLE_REC_GRID_DIM + 1], ah = r[ARCLE_REC_ANSWER_DIM], aw = r[ARCLE_REC_ANSWER_DIM + 1];
      const int mh = imin(gh, ah), mw = imin(gw, aw);
      if (tid == 0) x.red->sum = 0;
      bx::sync();  // (also: every plane store of the step is visible to the workgroup)
      int mine = 0;
      const int lastc = imin(nch, (imax(mh, 0) * W + 15) >> 4);
      BIG_EACH_CHUNK(x, c) {
        if (c >= lastc) break;
        const Chunk a = x.gl(ARCLE_PL_GRID, c), b = x.gl(ARCLE_PL_ANSWER, c);
        if (x.wide()) {  // (whole words: equal bytes under the common rectangle's mask, counted)
          const Chunk in = rect_mask16(c, W, x.wm, 0, mh, 0, mw);
#pragma unroll
          for (int q = 0; q < 4; q++) mine += __builtin_popcount(~nz_bytes(a.w[q] ^ b.w[q]) & in.w[q] & 0x01010101u);
          continue;
        }
        int f = 16 * c;
        int i = div_w(f, x.wm), j = f - i * W;
#pragma unroll
        for (int k = 0; k < 16; k++) {
          mine += (i < mh && j < mw && a.b[k] == b.b[k]) ? 1 : 0;
          if (++j == W) {
            j = 0;
            ++i;
          }
        }
      }
      if (mine) bx::lds_add(&x.red->sum, mine);
      bx::sync();
      correct = x.red->sum;
      total = mh * mw;
      if ((gh <= ah) == (gw <= aw)) total += ah * aw > gh * gw ? ah * aw - gh * gw : gh * gw - ah * aw;
      else total += (gh > ah ? gh - ah : ah - gh) * mw + (gw > aw ? gw - aw : aw - gw) * mh;
    }
    if (tid == 0) {
      p.dense[2 * (size_t)env] = correct;
      p.dense[2 * (size_t)env + 1] = total;
    }
  }
  const int term = r[ARCLE_REC_TERMINATED] != 0;
  const bool truncated = (flags & ARCLE_STEP_TRUNCATE) && cnt0 >= p.step_limit;
  if (accounting) {
    // byte accounting (arcle_enable_accounting): every 16-byte access of the step the threads counted, + the env's scalars (record in / out,
    // counters, action, outputs).  "issued" = those bytes; the other figure leaves the row padding out (chunks x 16 x P / PS).  Rows written
    // by the FLAT_OBS / PACK_OBS epilogue below are not in it (as in the one-wavefront kernels, where the host adds them per launch).
    if (tid == 0) x.red->sum = 0;
    bx::sync();
    if (x.io) bx::lds_add(&x.red->sum, (int)x.io);
    bx::sync();
    if (tid == 0) {
      const uint32_t chunks = (uint32_t)x.red->sum, scal = 2u * ARCLE_REC_BYTES + 16u + 20u + 5u;
      p.acct[env] += (uint32_t)(((uint64_t)chunks * 16u * (uint32_t)P) / (uint32_t)x.PS) + scal;
      p.acct[(size_t)p.n_envs + env] += chunks * 16u + scal;
    }
  }
  if (tid == 0) {
    Chunk rc;
#pragma unroll
    for (int k = 0; k < 4; k++) rc.w[k] = r.w[k];
    stg(p.rec, env, rc);
    p.cnt[2 * (size_t)env] = cnt0;
    p.cnt[2 * (size_t)env + 1] = cnt1;
    p.reward[env] = reward;
    p.term[env] = (uint8_t)term;
    if ((flags & ARCLE_STEP_TRUNCATE) && p.trunc) p.trunc[env] = (uint8_t)truncated;
    if (st) bx::status_or(p.status, st);
  }
  if (flags & (ARCLE_STEP_FLAT_OBS | ARCLE_STEP_PACK_OBS)) {
    bx::sync();  // every plane store of the step is visible to the workgroup
    emit_rows(x, r, flags, reward, term, cnt0, cnt1, truncated, st);
  }
}

// the generic form: every run-time parameter honoured (tuning launches, transition_rows, accounting, the row epilogues, the emulator)
ARCLE_BIG_DEV void step_env(const BigParams& p, const int env, int8_t* lds) { step_env_t<Ctx, ING_T_ANY>(p, env, lds); }

// ---- reset kernels (one workgroup per env) ---------------------------------------------------------------------------------------------
// mode 0: arcle_reset (init_state from PL_INPUT / REC_INPUT_DIM); 1: arcle_reset_from_table (task_idx); 2: arcle_reset_sampled
ARCLE_BIG_DEV void reset_env(const BigParams& p, int env, int mode, int8_t* lds) {
  if (p.rmask && !p.rmask[env]) return;
  const Ctx x(p, env, lds);
  Chunk rc;
  if (mode == 0) {
    rc = ldg(p.rec, env);
    bx::sync();
    init_planes(x, x.g(ARCLE_PL_INPUT), false);
  } else {
    rc = zero_chunk();
    int t, rot_k = 0;
    uint64_t perm = ARCLE_BIG_PERM_IDENTITY;
    uint32_t ep = 0;
    if (mode == 1) {
      t = p.task_idx[env];
      if (t < 0 || t >= p.n_tasks) {
        if (x.tid == 0) bx::status_or(p.status, ARCLE_ST_BAD_TASK);
        return;
      }
      if (p.aug_k) rot_k = p.aug_k[env] & 3;
      if (p.aug_perm) {
        perm = ARCLE_BIG_PERM_IDENTITY & ~0xFFFFFFFFFFull;  // (nibbles 10..15 stay the identity)
        for (int c = 0; c < 10; c++) perm |= (uint64_t)(p.aug_perm[16 * (size_t)env + c] & 15u) << (4 * c);
      }
    } else {
      ep = (uint32_t)p.episode[env];
      const Draw d = draw_task(p, env, ep);
      t = d.entry;
      rot_k = d.rot_k;
      perm = d.perm;
      bx::sync();
    }
    if (!load_task(x, rc.b, t, rot_k, perm, mode == 2)) {  // an explicit quarter turn that does not fit a non-square plane
      if (x.tid == 0) bx::status_or(p.status, ARCLE_ST_AUG_DOMAIN);
      return;
    }
    if (x.tid == 0) {
      if (mode == 2) p.episode[env] = (int32_t)(ep + 1u);
      if (p.cur_task) p.cur_task[env] = t;
    }
  }
  init_rec(rc.b, p.max_trial);
  if (x.tid == 0) {
    stg(p.rec, env, rc);
    p.cnt[2 * (size_t)env] = 0;
    p.cnt[2 * (size_t)env + 1] = 0;
  }
}

// ---- stand-alone row kernels -----------------------------------------------------------------------------------------------------------
// mode 0: arcle_flatten_obs / arcle_get_state_rows (rows of the resident state, no tail); 1: arcle_pack_obs (reward / term arrays given)
ARCLE_BIG_DEV void rows_env(const BigParams& p, int env, int mode, int8_t* lds) {
  const Ctx x(p, env, lds);
  if (mode == 2) {  // arcle_pack_mask_bits: int8 [N][P] masks (truthy = non-zero) -> bit rows of PS / 8 bytes (p.pack_out)
    const int8_t* const src = reinterpret_cast<const int8_t*>(p.sel) + (size_t)env * (size_t)x.P;
    uint8_t* const dst = p.pack_out + (size_t)env * (size_t)(x.PS >> 3);
    BIG_EACH_CHUNK(x, c) {
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (16 * c + k < x.P && src[16 * c + k] != 0) m |= 1u << k;
      dst[2 * c] = (uint8_t)m;
      dst[2 * c + 1] = (uint8_t)(m >> 8);
    }
    return;
  }
  const Chunk rc = ldg(p.rec, env);
  if (mode == 0) emit_rows(x, rc.b, ARCLE_STEP_FLAT_OBS, 0, 0, 0, 0, false, 0);
  else emit_rows(x, rc.b, ARCLE_STEP_PACK_OBS, p.reward[env], p.term[env], 0, 0, false, 0);
}

// arcle_set_state_rows: the inverse of the full (unfiltered) flat row — planes and the record's state fields from row `env` of rows_in
// (any alignment / stride); the task side (answer, answer_dim) and the counters stay
ARCLE_BIG_DEV void set_rows_env(const BigParams& p, int env, int8_t* lds) {
  if (p.rmask && !p.rmask[env]) return;
  const Ctx x(p, env, lds);
  const int8_t* const row = p.rows_in + (size_t)env * p.rows_in_stride;
  Chunk rc = ldg(p.rec, env);
  if (p.res_rec) {  // a scratch env of arcle_transition_rows: the task side comes from a resident env, the counters start at zero
    int src = p.src_env ? p.src_env[env] : env;
    if (src < 0 || src >= p.n_resident) src = 0;  // (the step launch flags the row and skips it)
    rc = ldg(p.res_rec, src);
    BIG_EACH_CHUNK(x, c) x.gs(ARCLE_PL_ANSWER, c, ldg(p.res_answer + (size_t)src * x.PS, c));
    if (x.tid == 0) p.cnt[2 * (size_t)env] = p.cnt[2 * (size_t)env + 1] = 0;
  }
  if (x.tid == 0) {
    flat_layout(p, 0, *x.lay);
    stg(x.sc, 0, rc);  // the record; the row's scalar segments overwrite their fields below
  }
  bx::sync();
  const Layout& L = *x.lay;
  for (int s = 0; s < L.n; s++) {
    const Seg sg = L.s[s];
    if (sg.plane < 0) {
      if (x.tid == 0)
        for (int k = 0; k < sg.len; k++) x.sc[sg.soff + k] = row[sg.start + k];
    } else {
      const int8_t* const src = row + sg.start;
      BIG_EACH_CHUNK(x, c) {
        Chunk v = zero_chunk();
        const int f0 = 16 * c;
        if (f0 + 16 <= x.P) {
          v = ldu(src + f0);
        } else {
#pragma unroll
          for (int k = 0; k < 16; k++)
            if (f0 + k < x.P) v.b[k] = src[f0 + k];
        }
        x.count();  // (the row bytes read)
        x.gs(sg.plane, c, v);
      }
    }
  }
  if (x.tid == 0) stg(p.rec, env, ldg(x.sc, 0));
}

}  // namespace arcle_big
