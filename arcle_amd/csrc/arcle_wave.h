// arcle_wave.h — the per-environment body of the ARCLE step kernel, written for ONE CDNA4 wavefront
// (64 lanes) per environment.
//
// Execution model
//   * one wavefront owns one env; the operation index is therefore wave-uniform and the op dispatch is
//     a scalar branch (no intra-wave divergence on the op);
//   * lane L owns the 16 consecutive cells [16L, 16L+16) of the row-major H x W plane (one aligned
//     dwordx4 per plane per lane — every plane access of the wave is a single coalesced 16 B/lane
//     instruction; the per-env plane stride is H*W rounded up to 16 B);
//   * cell predicates live in 16-bit per-lane masks (bit k <-> cell 16L+k); byte planes live in 4 VGPRs;
//   * geometric ops (object lift / place, Copy, Paste, Crop) are *uniform flat shifts* of a plane:
//     the plane is staged once in the wave's private LDS tile and read back at a wave-uniform byte
//     offset (dword reads + v_alignbyte); Rotate/Flip are affine index gathers from the LDS tile;
//   * FloodFill runs on a flat bit-board (32 cells per lane) with cross-lane shifts.
//
// This header is compiled by hipcc for gfx950 (arcle_hip.hip).  tests/emu/ compiles the very same
// header with g++ against a lock-step 64-thread emulation of the cross-lane primitives (namespace xl)
// so that the kernel LOGIC can be checked against the oracle without a GPU; that emulator is test
// infrastructure and is never part of the product library.
//
// Reference semantics restated here are cited per function (paths relative to /root/reference).
#pragma once
#include <stdint.h>

#include "../../include/arcle_hip.h"

#ifndef ARCLE_DEV
#error "include through arcle_hip.hip (or the test emulator), which defines ARCLE_DEV and namespace xl"
#endif

namespace arcle {

enum { INGRESS_MASK = 0, INGRESS_BBOX = 1, INGRESS_POINT = 2 };

struct StepParams {
  int8_t* plane[ARCLE_N_PLANES];
  int8_t* rec;
  int32_t* cnt;
  const int32_t* op;
  const void* sel;  // ingress payload: int8 [N][P] | int32 [N][4] | int32 [N][2]
  int32_t* reward;
  uint8_t* term;
  uint32_t* status;
  uint32_t* acct;        // optional per-env algorithmic-byte accumulator
  const uint8_t* rmask;  // reset kernels only
  const int32_t* task_idx;                   // reset-from-table kernel only
  const int8_t *tbl_in, *tbl_ans;            // task table planes [n_tasks][PS]
  const int8_t *tbl_in_dim, *tbl_ans_dim;    // task table dims   [n_tasks][2]
  int32_t n_tasks;
  int32_t n_steps;  // rollout kernel only: steps per launch
  int8_t* flat_out;  // flatten kernel only
  int32_t flat_len;
  int32_t n_envs, H, W, P, PS;  // PS = plane stride in bytes (P rounded up to 16)
  int32_t n_ops, max_trial, ingress;
  uint32_t flags;
  uint32_t div_magic;  // floor(65536/W)+1 : (n*div_magic)>>16 == n/W for n < 1040 (checked at create)
  int32_t nseg;        // max row segments a 16-cell lane window can span
  const uint32_t* d_ops;  // device copy of the op table, ARCLE_MAX_OPS entries (unused slots 0)
};

// 16 bytes of a plane = 4 VGPRs; a first-class vector value so that it always lives in registers
#if defined(__clang__)
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
#else
typedef uint32_t U4 __attribute__((vector_size(16)));
#endif

struct I2 {
  int32_t x, y;
};

struct WaveLDS {
  uint32_t a[256];  // 1024 B staging tile (bytes of one plane)
  uint32_t b[256];  // second tile (object_sel during Rotate/Flip)
};

// ------------------------------------------------------------------------------------------------
// small bit helpers (per lane)
// ------------------------------------------------------------------------------------------------
ARCLE_DEV int imin(int a, int b) { return a < b ? a : b; }
ARCLE_DEV int imax(int a, int b) { return a > b ? a : b; }
ARCLE_DEV int i8w(int x) { return (int)(int8_t)(uint8_t)(x & 0xff); }  // wrap to int8
ARCLE_DEV int floordiv2(int a) { return a >> 1; }                       // arithmetic shift == floor(a/2)
ARCLE_DEV uint32_t bits_range(int a, int b) { return (2u << b) - (1u << a); }  // bits a..b, 0<=a<=b<=30

// bit7-per-byte flags: byte != 0
ARCLE_DEV uint32_t nzflags(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }
// 0x80 flags -> 4-bit nibble
ARCLE_DEV uint32_t flags2nib(uint32_t t) {
  uint32_t y = t >> 7;
  return (y | (y >> 7) | (y >> 14) | (y >> 21)) & 0xfu;
}
ARCLE_DEV uint32_t nz16(const U4& v) {
  return flags2nib(nzflags(v[0])) | (flags2nib(nzflags(v[1])) << 4) | (flags2nib(nzflags(v[2])) << 8) |
         (flags2nib(nzflags(v[3])) << 12);
}
// signed int8 > 0  <=>  non-zero and sign bit clear
ARCLE_DEV uint32_t pos16(const U4& v) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) m |= flags2nib(nzflags(v[i]) & ~v[i]) << (4 * i);
  return m;
}
// 0xff for every byte that is > 0 as int8
ARCLE_DEV U4 posbytes(const U4& v) {
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t t = nzflags(v[i]) & ~v[i];  // 0x80 in qualifying bytes
    r[i] = t | (t - (t >> 7));
  }
  return r;
}
// 4-bit nibble -> 0xff byte mask per set bit
ARCLE_DEV uint32_t nib2bytes(uint32_t nib) {
  uint32_t x = (nib * 0x00204081u) & 0x01010101u;
  return (x << 8) - x;
}
ARCLE_DEV U4 expand16(uint32_t m) {
  U4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = nib2bytes((m >> (4 * i)) & 0xfu);
  return r;
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
ARCLE_DEV U4 u4_splat(uint32_t byte) {
  U4 r;
  r[0] = r[1] = r[2] = r[3] = (byte & 0xffu) * 0x01010101u;
  return r;
}
ARCLE_DEV uint32_t eq16(const U4& v, uint32_t byte) {  // bytes == byte
  uint32_t c = (byte & 0xffu) * 0x01010101u, m = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) m |= (flags2nib(nzflags(v[i] ^ c)) ^ 0xfu) << (4 * i);
  return m;
}
ARCLE_DEV uint32_t u4_byte(const U4& v, int k) {  // dynamic byte extract
  uint32_t w = (k & 8) ? ((k & 4) ? v[3] : v[2]) : ((k & 4) ? v[1] : v[0]);
  return (w >> (8 * (k & 3))) & 0xffu;
}

// ------------------------------------------------------------------------------------------------
// wave context
// ------------------------------------------------------------------------------------------------
struct Wave {
  const StepParams& p;
  WaveLDS* lds;
  int env, lane;
  int r0, c0;        // row / col of this lane's first cell
  uint32_t valid16;  // cells of this lane that exist (flat index < P)
  bool live;         // lane holds at least one cell (lane < PS/16)
  // 16 <= W <= 32: the lane's window covers row r0 from column c0 (k1 cells, mask lm) and then row r0+1
  // from column 0 (mask hm); rectangle masks then cost a dozen VALU ops (rect16 fast path)
  bool fastw;
  int k1;
  uint32_t lm, hm;
  // rollout mode: the env's planes stay in registers across steps; load/store then never touch HBM
  bool resident;
  mutable U4 cache[ARCLE_N_PLANES];
  mutable uint32_t dirty;  // planes of `cache` that differ from HBM
#ifdef ARCLE_TRACE_WAVES
  mutable uint64_t t_desc, t_sel;
#endif
  int ingress;    // INGRESS_* (a compile-time constant of the kernel instantiation)
  uint32_t poff;  // byte offset of this lane's 16 cells inside a plane: env*PS + 16*lane (< 4 GiB, checked at create)

  // fw: 1 = the instantiation is only launched for 16 <= W <= 32 (generic rectangle code compiled out),
  //     0 = generic
  ARCLE_DEV Wave(const StepParams& p_, WaveLDS* l, int env_, int lane_, int ingress_, int fw)
      : p(p_), lds(l), env(env_), lane(lane_) {
    ingress = ingress_;
    resident = false;
    dirty = 0;
    poff = (uint32_t)env * (uint32_t)p.PS + 16u * (uint32_t)lane;
    uint32_t f0 = 16u * (uint32_t)lane;
    r0 = (int)((f0 * p.div_magic) >> 16);
    c0 = (int)f0 - r0 * p.W;
    int nv = imin(imax(p.P - (int)f0, 0), 16);
    valid16 = (1u << nv) - 1u;
    live = (int)f0 < p.PS;
    fastw = fw != 0;
    k1 = imin(16, p.W - c0);
    lm = (1u << k1) - 1u;
    hm = 0xffffu & ~lm;
  }

  // ---- plane I/O: one aligned 16 B access per lane -------------------------------------------
  ARCLE_DEV U4 load_hbm(int pl) const {
    U4 v = u4_zero();
    if (live) v = *reinterpret_cast<const U4*>(p.plane[pl] + poff);
    return v;
  }
  ARCLE_DEV void store_hbm(int pl, const U4& v) const {
    if (live) xl::store16(p.plane[pl] + poff, v);
  }
  ARCLE_DEV U4 load(int pl) const { return resident ? cache[pl] : load_hbm(pl); }
  ARCLE_DEV void store(int pl, const U4& v) const {
    if (resident) {
      cache[pl] = v;
      dirty |= 1u << pl;
    } else {
      store_hbm(pl, v);
    }
  }

  // ---- 16-bit mask of this lane's cells inside rows [x1,x2] x cols [y1,y2] (inclusive) ----------
  ARCLE_DEV uint32_t rect16(int x1, int x2, int y1, int y2) const {
    if (fastw) {
      // column mask of the rectangle (wave-uniform, scalar ALU); empty rectangle -> 0
      uint32_t cm = (x1 <= x2 && y1 <= y2) ? ((2u << y2) - (1u << y1)) : 0u;
      uint32_t dx = (uint32_t)(x2 - x1);
      uint32_t s0 = ((uint32_t)(r0 - x1) <= dx) ? ((cm >> c0) & lm) : 0u;
      uint32_t s1 = ((uint32_t)(r0 + 1 - x1) <= dx) ? ((cm << k1) & hm) : 0u;
      return (s0 | s1) & valid16;
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
#pragma unroll
    for (int i = 0; i < 4; i++) buf[4 * lane + i] = v[i];
    xl::lds_fence();
  }
  // out[f] = staged[f + S] for this lane's 16 cells.  Cells whose source falls outside the tile get
  // unspecified bytes — every caller masks the result with a rectangle that excludes them.
  ARCLE_DEV U4 shifted(const uint32_t* buf, int S) const {
    int base = 4 * lane + (S >> 2);
    uint32_t sh = (uint32_t)S & 3u;
    uint32_t w[5];
#pragma unroll
    for (int j = 0; j < 5; j++) w[j] = buf[xl::lds_idx(base + j, 256)];
    U4 r;
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = xl::alignbyte(w[j + 1], w[j], sh);
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

// ------------------------------------------------------------------------------------------------
// selection ingress: action['selection'] as a cell mask (bbox.py:22-30, :43-49 fused on device)
// ------------------------------------------------------------------------------------------------
struct Sel {
  uint32_t nz;   // truthy cells (np.any(sel), ma mask, logical_and)
  uint32_t pos;  // cells with sel > 0
  U4 vals;       // the raw int8 values (for `selected = sel`, object.py:96 / keep_sel :38)
  bool any_nz, any_pos;
  int x0, x1, y0, y1;  // _get_bbox of the truthy cells (object.py:49-58); valid iff any_nz
  bool is_rect;        // built from a bbox / point tuple: every cell of the bbox is 1
};

ARCLE_DEV void sel_from_rect(const Wave& w, Sel& s, int x1, int x2, int y1, int y2) {
  s.is_rect = true;
  s.any_nz = s.any_pos = (x1 <= x2 && y1 <= y2);
  s.x0 = x1;
  s.x1 = x2;
  s.y0 = y1;
  s.y1 = y2;
  s.nz = s.pos = s.any_nz ? w.rect16(x1, x2, y1, y2) : 0u;
}

// the raw int8 selection values (`selected = sel`, keep_sel object.py:38; mask ingress keeps what it loaded)
ARCLE_DEV U4 sel_values(const Sel& s) {
  if (!s.is_rect) return s.vals;
  U4 e = expand16(s.nz);
#pragma unroll
  for (int i = 0; i < 4; i++) e[i] &= 0x01010101u;
  return e;
}

// The selection payload of this env, fetched in the same latency window as the record / op / counters
// (all four are independent of each other): bbox = 4 ints, point = 2 ints, mask = this lane's 16 cells.
ARCLE_DEV U4 load_payload(const Wave& w, size_t step = 0) {  // step: rollout kernels index [step][env]
  const StepParams& p = w.p;
  U4 v = u4_zero();
  const size_t e = step * (size_t)p.n_envs + (size_t)w.env;
  if (w.ingress == INGRESS_BBOX) {
    v = *reinterpret_cast<const U4*>(reinterpret_cast<const int32_t*>(p.sel) + 4 * e);
  } else if (w.ingress == INGRESS_POINT) {
    const uint32_t* b = reinterpret_cast<const uint32_t*>(p.sel) + 2 * e;
    v[0] = b[0];
    v[1] = b[1];
  } else {
    // full mask, contiguous int8 [N][P] as the caller holds it (no 16 B alignment guarantee)
    const int8_t* src = reinterpret_cast<const int8_t*>(p.sel) + (size_t)w.env * p.P + 16 * w.lane;
    if ((p.P & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.sel) & 3) == 0)) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (16 * w.lane + 4 * i < p.P) v[i] = *reinterpret_cast<const uint32_t*>(src + 4 * i);
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++)
        if (16 * w.lane + k < p.P) v[k >> 2] |= (uint32_t)(uint8_t)src[k] << (8 * (k & 3));
    }
  }
  return v;
}

ARCLE_DEV void ingest_selection(const Wave& w, Sel& s, const U4& payload) {
  const StepParams& p = w.p;
  if (w.ingress == INGRESS_BBOX) {
    // BBoxWrapper.action (bbox.py:22-30): sort the corners, sel[x1:x2+1, y1:y2+1] = 1 (slices clip at H, W;
    // negative coordinates are outside the wrapper's Discrete action space and select nothing here)
    int bx1 = (int)xl::uniform(payload[0]), by1 = (int)xl::uniform(payload[1]);
    int bx2 = (int)xl::uniform(payload[2]), by2 = (int)xl::uniform(payload[3]);
    int xa = imin(bx1, bx2), xb = imin(imax(bx1, bx2), p.H - 1);
    int ya = imin(by1, by2), yb = imin(imax(by1, by2), p.W - 1);
    if (xa < 0 || ya < 0) xa = xb + 1;
    sel_from_rect(w, s, xa, xb, ya, yb);
    return;
  }
  if (w.ingress == INGRESS_POINT) {
    // PointWrapper.action (bbox.py:43-49)
    int x = (int)xl::uniform(payload[0]), y = (int)xl::uniform(payload[1]);
    bool ok = x >= 0 && x < p.H && y >= 0 && y < p.W;
    sel_from_rect(w, s, x, ok ? x : x - 1, y, y);
    return;
  }
  const U4 v = payload;
  s.is_rect = false;
  s.vals = v;
  s.nz = nz16(v) & w.valid16;
  s.pos = pos16(v) & w.valid16;
  s.any_nz = w.any(s.nz != 0);
  s.any_pos = w.any(s.pos != 0);
  s.x0 = s.x1 = s.y0 = s.y1 = 0;
  if (s.any_nz) {
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
// state held in registers during one step
// ------------------------------------------------------------------------------------------------
enum {
  WR_GRID = 1 << ARCLE_PL_GRID,
  WR_SELECTED = 1 << ARCLE_PL_SELECTED,
  WR_CLIP = 1 << ARCLE_PL_CLIP,
  WR_OBJECT = 1 << ARCLE_PL_OBJECT,
  WR_OBJECT_SEL = 1 << ARCLE_PL_OBJECT_SEL,
  WR_BACKGROUND = 1 << ARCLE_PL_BACKGROUND
};

struct Rec {  // the 16-byte scalar record, unpacked
  int in_h, in_w, gh, gw, ch, cw, oh, ow, ox, oy, trials, term, active, parity, ah, aw;
};
ARCLE_DEV int sb(uint32_t w, int i) { return (int)(int8_t)((w >> (8 * i)) & 0xffu); }
ARCLE_DEV void rec_unpack(const U4& v, Rec& r) {
  r.in_h = sb(v[0], 0); r.in_w = sb(v[0], 1); r.gh = sb(v[0], 2); r.gw = sb(v[0], 3);
  r.ch = sb(v[1], 0); r.cw = sb(v[1], 1); r.oh = sb(v[1], 2); r.ow = sb(v[1], 3);
  r.ox = sb(v[2], 0); r.oy = sb(v[2], 1); r.trials = sb(v[2], 2); r.term = sb(v[2], 3);
  r.active = sb(v[3], 0); r.parity = sb(v[3], 1); r.ah = sb(v[3], 2); r.aw = sb(v[3], 3);
}
ARCLE_DEV uint32_t pk(int a, int b, int c, int d) {
  return ((uint32_t)a & 0xffu) | (((uint32_t)b & 0xffu) << 8) | (((uint32_t)c & 0xffu) << 16) | (((uint32_t)d & 0xffu) << 24);
}
ARCLE_DEV U4 rec_pack(const Rec& r) {
  U4 v;
  v[0] = pk(r.in_h, r.in_w, r.gh, r.gw);
  v[1] = pk(r.ch, r.cw, r.oh, r.ow);
  v[2] = pk(r.ox, r.oy, r.trials, r.term);
  v[3] = pk(r.active, r.parity, r.ah, r.aw);
  return v;
}

struct Planes {
  U4 grid, selected, clip, object, object_sel, background;
  uint32_t wr;      // WR_* planes to write back
  bool have_grid;   // `grid` holds the current grid plane
  uint32_t bytes;   // algorithmic HBM bytes of this step (SURVEY.md §8d accounting)
};

ARCLE_DEV void need_grid(const Wave& w, Planes& s) {
  if (!s.have_grid) {
    s.grid = w.load(ARCLE_PL_GRID);
    s.have_grid = true;
    s.bytes += w.p.P;  // algorithmic accounting: the op semantically reads the grid
  }
}

// grid[:gh,:gw] == answer with equal dims (base.py:176-177, o2arcenv.py:124-127)
ARCLE_DEV bool grid_equals_answer(const Wave& w, Planes& s, const Rec& r) {
  if (r.gh != r.ah || r.gw != r.aw) return false;
  need_grid(w, s);
  U4 a = w.load(ARCLE_PL_ANSWER);
  s.bytes += w.p.P;
  U4 m = expand16(w.rect16(0, r.gh - 1, 0, r.gw - 1));
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
ARCLE_DEV void place(const Wave& w, Planes& s, const Rec& r, const uint32_t* tile, uint32_t osel, int S0,
                     bool osel_full = false) {
  const int W = w.p.W;
  s.grid = s.background;
  s.selected = u4_zero();
  int xh = i8w(r.ox + r.oh), yw = i8w(r.oy + r.ow);  // int8 + int8 (object.py:127)
  if (xh > 0 && r.ox < r.gh && yw > 0 && r.oy < r.gw) {
    int stx = imax(0, r.ox), edx = imin(r.gh, xh), sty = imax(0, r.oy), edy = imin(r.gw, yw);
    uint32_t R = w.rect16(stx, edx - 1, sty, edy - 1);
    int S = S0 - (r.ox * W + r.oy);
    U4 po = w.shifted(tile, S);
    const U4 rb = expand16(R);
    s.grid = u4_sel(u4_and(rb, posbytes(po)), po, s.background);  // where=(p>0)  object.py:138
    // object.py:165; when object_sel covers the whole object tile (rectangle selection) the placed mask IS R
    const U4 e = osel_full ? rb : expand16(w.shifted_bits(osel, S) & R);
#pragma unroll
    for (int i = 0; i < 4; i++) s.selected[i] = e[i] & 0x01010101u;
  }
  s.wr |= WR_GRID | WR_SELECTED;
  s.have_grid = true;
  s.bytes += 2 * w.p.P;
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
  U4 lmb = expand16(w.lm);  // bytes of the first row segment
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
  bool ok;        // false: inactive and nothing selected -> the op is a no-op
  bool fresh;     // a new selection was lifted
  uint32_t osel;  // object_sel cell mask, in the frame of `tile` (see S0)
  int S0;         // staged object cell f is at tile[f + S0]
  bool osel_full; // object_sel == the whole h x w object tile (selection was a rectangle)
};

// After this call lds->a holds the object bytes (shifted by S0) and s.object/object_sel/background are
// current in registers.  For a fresh selection the tile is the masked GRID (S0 = x0*W + y0), so that
// Move can place straight from it without a second staging.
ARCLE_DEV Lift init_objsel(const Wave& w, Planes& s, Rec& r, const Sel& sel) {
  Lift L;
  const int W = w.p.W, P = w.p.P;
  if (sel.any_nz) {  // object.py:67-99
    need_grid(w, s);
    int h = sel.x1 - sel.x0 + 1, wd = sel.y1 - sel.y0 + 1;
    U4 pm = expand16(sel.pos);
    // every read of the tile below is masked by a rectangle inside the selection's bbox image, so for a
    // rectangle selection (all cells of the bbox selected) the grid itself can be staged
    w.stage(w.lds->a, sel.is_rect ? s.grid : u4_and(s.grid, pm));
    int S0 = sel.x0 * W + sel.y0;
    uint32_t orect = w.rect16(0, h - 1, 0, wd - 1);
    const U4 ob = expand16(orect);
    s.object = u4_and(w.shifted(w.lds->a, S0), ob);
    const U4 e = sel.is_rect ? ob : expand16(w.shifted_bits(sel.pos, S0) & orect);
#pragma unroll
    for (int i = 0; i < 4; i++) s.object_sel[i] = e[i] & 0x01010101u;
    s.background = u4_andn(s.grid, pm);
    r.oh = h;
    r.ow = wd;
    r.ox = sel.x0;
    r.oy = sel.y0;
    r.active = 1;
    r.parity = 0;
    s.wr |= WR_OBJECT | WR_OBJECT_SEL | WR_BACKGROUND;
    s.bytes += 3 * P;
    L.ok = true;
    L.fresh = true;
    L.osel = sel.pos;  // in the grid frame, consistent with the tile
    L.S0 = S0;
    L.osel_full = sel.is_rect;
    return L;
  }
  if (r.active) {  // object.py:102-107
    s.object = w.load(ARCLE_PL_OBJECT);
    s.object_sel = w.load(ARCLE_PL_OBJECT_SEL);
    s.background = w.load(ARCLE_PL_BACKGROUND);
    s.bytes += 3 * P;
    w.stage(w.lds->a, s.object);
    L.ok = true;
    L.fresh = false;
    L.osel = nz16(s.object_sel);
    L.S0 = 0;
    L.osel_full = false;
    return L;
  }
  L.ok = false;
  L.fresh = false;
  L.osel = 0;
  L.S0 = 0;
  L.osel_full = false;
  return L;
}

// dst[:nh,:nw] = T(src[:h,:w]), rest 0 (_pad_assign object.py:43-47) for both object and object_sel.
// src index = ai*i + bj*j + c0 (affine in the destination cell), read from the LDS tiles.
ARCLE_DEV void tile_transform(const Wave& w, Planes& s, int nh, int nw, int ai, int bj, int c0) {
  const int W = w.p.W;
  w.stage(w.lds->a, s.object);
  w.stage(w.lds->b, s.object_sel);
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
  s.object = o;
  s.object_sel = os;
  s.wr |= WR_OBJECT | WR_OBJECT_SEL;
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

ARCLE_DEV void op_floodfill(const Wave& w, Planes& s, const Rec& r, const Sel& sel, int color) {
  const StepParams& p = w.p;
  int seed;
  if (sel.is_rect) {  // np.sum(sel) == 1  <=>  1x1 rectangle
    if (!(sel.any_nz && sel.x0 == sel.x1 && sel.y0 == sel.y1)) return;
    seed = sel.x0 * p.W + sel.y0;
  } else {
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
  int sx = (int)(((uint32_t)seed * p.div_magic) >> 16), sy = seed - sx * p.W;
  if (sx >= r.gh || sy >= r.gw) return;  // color.py:96
  need_grid(w, s);
  // colour at the seed: lane seed>>4, byte seed&15
  uint32_t mine = u4_byte(s.grid, seed & 15);
  uint32_t col = xl::uniform(xl::shfl(mine, seed >> 4));
  uint32_t inside = w.rect16(0, r.gh - 1, 0, r.gw - 1);
  uint32_t M = to32(w, eq16(s.grid, col) & inside);
  uint32_t notfirst = to32(w, w.rect16(0, p.H - 1, 1, p.W - 1));
  uint32_t notlast = to32(w, w.rect16(0, p.H - 1, 0, p.W - 2));
  uint32_t F = (w.lane == (seed >> 5)) ? (1u << (seed & 31)) : 0u;
  if (w.fastw) {
    // 16 <= W <= 32: every neighbour shift (1 or W bits) only needs the adjacent lanes' words, fetched with two DPP
    // wave shifts per propagation step (no LDS round trip); 4 steps per convergence ballot (the closure is monotone,
    // extra steps are harmless)
    const uint32_t Wb = (uint32_t)p.W;
    for (int it = 0; it < ARCLE_MAX_CELLS / 4 + 1; it++) {
      const uint32_t F0 = F;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t prev = xl::lane_prev(F), next = xl::lane_next(F);  // lane j-1 / j+1, 0 outside the wave
        const uint32_t l1 = (F << 1) | (prev >> 31), r1 = (F >> 1) | (next << 31);
        const uint32_t lW = (Wb == 32) ? prev : ((F << Wb) | (prev >> (32 - Wb)));
        const uint32_t rW = (Wb == 32) ? next : ((F >> Wb) | (next << (32 - Wb)));
        F |= ((l1 & notfirst) | (r1 & notlast) | lW | rW) & M;
      }
      if (!w.any(F != F0)) break;
    }
  } else {
    for (int it = 0; it < ARCLE_MAX_CELLS; it++) {
      uint32_t grow = (board_shl(w, F, 1) & notfirst) | (board_shr(w, F, 1) & notlast) | board_shl(w, F, p.W) |
                      board_shr(w, F, p.W);
      uint32_t Fn = F | (grow & M);
      bool changed = w.any(Fn != F);
      F = Fn;
      if (!changed) break;
    }
  }
  uint32_t vis = to16(w, F);
  s.grid = u4_sel(expand16(vis), u4_splat((uint32_t)color), s.grid);
  s.wr |= WR_GRID;
  s.bytes += p.P;
}

// the 16-byte record is identical in every lane: tell the compiler so (scalar control flow)
ARCLE_DEV U4 load_rec(const StepParams& p, int env) {
  U4 rv = *reinterpret_cast<const U4*>(p.rec + (size_t)env * ARCLE_REC_BYTES);
#pragma unroll
  for (int i = 0; i < 4; i++) rv[i] = xl::uniform(rv[i]);
  return rv;
}

// ------------------------------------------------------------------------------------------------
// init_state (base.py:155-166 + o2arcenv.py:16-34 / arcenv.py:81-89), counters as in reset (base.py:73-79)
// ------------------------------------------------------------------------------------------------
ARCLE_DEV void init_state(const Wave& w, Rec& r, I2& cnt) {
  const StepParams& p = w.p;
  U4 in = w.load(ARCLE_PL_INPUT);
  w.store(ARCLE_PL_GRID, in);
  U4 z = u4_zero();
  if (p.plane[ARCLE_PL_SELECTED]) w.store(ARCLE_PL_SELECTED, z);
  if (p.plane[ARCLE_PL_CLIP]) w.store(ARCLE_PL_CLIP, z);
  if (p.plane[ARCLE_PL_OBJECT]) w.store(ARCLE_PL_OBJECT, z);
  if (p.plane[ARCLE_PL_OBJECT_SEL]) w.store(ARCLE_PL_OBJECT_SEL, z);
  if (p.plane[ARCLE_PL_BACKGROUND]) w.store(ARCLE_PL_BACKGROUND, z);
  r.gh = r.in_h;
  r.gw = r.in_w;
  r.ch = r.cw = r.oh = r.ow = r.ox = r.oy = 0;
  r.trials = i8w(p.max_trial);
  r.term = r.active = r.parity = 0;
  cnt.x = 0;
  cnt.y = 0;
}

ARCLE_DEV void store_rec_cnt(const StepParams& p, int env, int lane, const Rec& r, const I2& cnt) {
  if (lane == 0) {
    *reinterpret_cast<U4*>(p.rec + (size_t)env * ARCLE_REC_BYTES) = rec_pack(r);
    *reinterpret_cast<I2*>(p.cnt + 2 * (size_t)env) = cnt;
  }
}

ARCLE_DEV void wave_reset(const StepParams& p, WaveLDS* lds, int env, int lane) {
  if (p.rmask && !p.rmask[env]) return;
  Wave w(p, lds, env, lane, INGRESS_BBOX, 0);
  Rec r;
  rec_unpack(load_rec(p, env), r);
  I2 cnt;
  init_state(w, r, cnt);
  xl::lds_fence();  // (emulator) every lane has read the record before lane 0 rewrites it
  store_rec_cnt(p, env, lane, r, cnt);
}

// reset() with a caller-chosen task (base.py:95-108): the (input, answer) pair comes from the device task table
ARCLE_DEV void wave_reset_table(const StepParams& p, WaveLDS* lds, int env, int lane) {
  if (p.rmask && !p.rmask[env]) return;
  const int t = (int)xl::uniform((uint32_t)p.task_idx[env]);
  if (t < 0 || t >= p.n_tasks) {
    if (lane == 0) xl::atomic_or(p.status, ARCLE_ST_BAD_TASK);
    return;
  }
  Wave w(p, lds, env, lane, INGRESS_BBOX, 0);
  U4 in = u4_zero(), an = u4_zero();
  if (w.live) {
    in = *reinterpret_cast<const U4*>(p.tbl_in + (size_t)t * p.PS + 16 * lane);
    an = *reinterpret_cast<const U4*>(p.tbl_ans + (size_t)t * p.PS + 16 * lane);
  }
  w.store(ARCLE_PL_INPUT, in);
  w.store(ARCLE_PL_ANSWER, an);
  w.store(ARCLE_PL_GRID, in);
  U4 z = u4_zero();
  if (p.plane[ARCLE_PL_SELECTED]) w.store(ARCLE_PL_SELECTED, z);
  if (p.plane[ARCLE_PL_CLIP]) w.store(ARCLE_PL_CLIP, z);
  if (p.plane[ARCLE_PL_OBJECT]) w.store(ARCLE_PL_OBJECT, z);
  if (p.plane[ARCLE_PL_OBJECT_SEL]) w.store(ARCLE_PL_OBJECT_SEL, z);
  if (p.plane[ARCLE_PL_BACKGROUND]) w.store(ARCLE_PL_BACKGROUND, z);
  if (lane == 0) {
    Rec r;
    r.in_h = r.gh = p.tbl_in_dim[2 * t];
    r.in_w = r.gw = p.tbl_in_dim[2 * t + 1];
    r.ah = p.tbl_ans_dim[2 * t];
    r.aw = p.tbl_ans_dim[2 * t + 1];
    r.ch = r.cw = r.oh = r.ow = r.ox = r.oy = 0;
    r.trials = i8w(p.max_trial);
    r.term = r.active = r.parity = 0;
    *reinterpret_cast<U4*>(p.rec + (size_t)env * ARCLE_REC_BYTES) = rec_pack(r);
    p.cnt[2 * (size_t)env + ARCLE_CNT_STEPS] = 0;
    p.cnt[2 * (size_t)env + ARCLE_CNT_SUBMIT] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// one step() of one env:  O2ARCv2Env.step (o2arcenv.py:130-147) / ARCEnv.step / RawARCEnv.step
// ------------------------------------------------------------------------------------------------
// Descriptor of slot `op` for the three canonical tables, computed in scalar registers instead of fetched: the
// table lookup is a dependent scalar load on every wave's critical path (~0.5 us measured), and these tables are
// what O2ARCv2Env / ARCEnv / RawARCEnv install (o2arcenv.py:88-113, arcenv.py:123-137, arcenv.py:26-41).  The host
// selects the instantiation only when the installed table equals the canonical one; any other table uses the lookup.
enum { TBL_LOOKUP = 0, TBL_O2ARC = 1, TBL_ARC = 2, TBL_RAW = 3 };

ARCLE_DEV uint32_t packed_slot(uint64_t lo, uint64_t hi, int i) {  // 8-bit fields: kind | arg << 4 | flags << 6
  const uint64_t w = (i < 8) ? lo : hi;
  const uint32_t f = (uint32_t)(w >> (8 * (i & 7))) & 0xffu;
  return ARCLE_OP_DESC(f & 0xfu, (f >> 4) & 0x3u, f >> 6);
}
#define ARCLE_PK(kind, arg, flags) ((uint64_t)((kind) | ((arg) << 4) | ((flags) << 6)))
#define ARCLE_PK8(a, b, c, d, e, f, g, h) \
  ((a) | ((b) << 8) | ((c) << 16) | ((d) << 24) | ((e) << 32) | ((f) << 40) | ((g) << 48) | ((h) << 56))

template <int TBL>
ARCLE_DEV uint32_t decode_op(const StepParams& p, int op) {
  if (TBL == TBL_O2ARC) {  // 0-9 Color|R, 10-19 FloodFill|R, 20.. below
    if (op < 10) return ARCLE_OP_DESC(ARCLE_OP_COLOR, op, ARCLE_OPF_RESET_SEL);
    if (op < 20) return ARCLE_OP_DESC(ARCLE_OP_FLOODFILL, op - 10, ARCLE_OPF_RESET_SEL);
    const uint64_t lo = ARCLE_PK8(ARCLE_PK(ARCLE_OP_MOVE, 0, 0), ARCLE_PK(ARCLE_OP_MOVE, 1, 0), ARCLE_PK(ARCLE_OP_MOVE, 2, 0),
                                  ARCLE_PK(ARCLE_OP_MOVE, 3, 0), ARCLE_PK(ARCLE_OP_ROTATE, 1, 0), ARCLE_PK(ARCLE_OP_ROTATE, 3, 0),
                                  ARCLE_PK(ARCLE_OP_FLIP, 0, 0), ARCLE_PK(ARCLE_OP_FLIP, 1, 0));
    const uint64_t hi = ARCLE_PK8(ARCLE_PK(ARCLE_OP_COPY, 0, 1), ARCLE_PK(ARCLE_OP_COPY, 1, 1), ARCLE_PK(ARCLE_OP_PASTE, 1, 1),
                                  ARCLE_PK(ARCLE_OP_COPY_FROM_INPUT, 0, 1), ARCLE_PK(ARCLE_OP_RESET_GRID, 0, 1),
                                  ARCLE_PK(ARCLE_OP_RESIZE_GRID, 0, 1), ARCLE_PK(ARCLE_OP_SUBMIT, 0, 0), (uint64_t)0);
    return packed_slot(lo, hi, op - 20);
  }
  if (TBL == TBL_ARC) {
    if (op < 10) return ARCLE_OP_DESC(ARCLE_OP_COLOR, op, 0);
    if (op < 20) return ARCLE_OP_DESC(ARCLE_OP_FLOODFILL, op - 10, 0);
    const uint64_t lo = ARCLE_PK8(ARCLE_PK(ARCLE_OP_COPY, 0, 0), ARCLE_PK(ARCLE_OP_COPY, 1, 0), ARCLE_PK(ARCLE_OP_PASTE, 1, 0),
                                  ARCLE_PK(ARCLE_OP_COPY_FROM_INPUT, 0, 0), ARCLE_PK(ARCLE_OP_RESET_GRID, 0, 0),
                                  ARCLE_PK(ARCLE_OP_RESIZE_GRID, 0, 0), ARCLE_PK(ARCLE_OP_SUBMIT, 0, 0), (uint64_t)0);
    return packed_slot(lo, 0, op - 20);
  }
  if (TBL == TBL_RAW) {
    if (op < 10) return ARCLE_OP_DESC(ARCLE_OP_COLOR, op, 0);
    return op == 10 ? ARCLE_OP_DESC(ARCLE_OP_RESIZE_TO_ANSWER, 0, 0) : ARCLE_OP_DESC(ARCLE_OP_SUBMIT, 0, 0);
  }
  // any other table: scalar load through the constant cache (a per-lane vector fetch of the 256 B table hot-spots one
  // L2 channel)
  return p.d_ops[op];
}

struct StepOut {
  int reward;      // 0/1
  bool term;       // bool(state['terminated'])
  uint32_t bytes;  // algorithmic HBM bytes of the step (0 for skipped steps)
};

// Everything of step() between "record/op/payload are in registers" and "record/counters/outputs go back to
// memory": autoreset, op decode, the operation itself, reward.  Planes are read/written through w.load/w.store, so
// the same code serves the single-step kernel (HBM) and the rollout kernel (register-resident planes).
template <int ING, int FW, int TBL>
ARCLE_DEV StepOut step_core(const Wave& w, Rec& r, I2& cnt0, const U4& payload, const int op) {
  const StepParams& p = w.p;
  const int P = p.P, W = p.W, lane = w.lane;
  StepOut out;
  out.reward = 0;
  out.bytes = 0;
  if ((p.flags & ARCLE_STEP_AUTORESET) && r.term != 0) {
    init_state(w, r, cnt0);
    out.term = 0;
    out.bytes = (uint32_t)(7 * P + 2 * ARCLE_REC_BYTES);
    return out;
  }
  bool bad_op = op < 0 || op >= p.n_ops;
  const uint32_t desc = bad_op ? 0u : decode_op<TBL>(p, op);
  if (!bad_op) bad_op = ARCLE_OP_KIND(desc) == ARCLE_OP_NONE;
#ifdef ARCLE_TRACE_WAVES
  w.t_desc = xl::clock();
#endif
  if (bad_op) {
    // reference: IndexError / TypeError before any mutation
    if (lane == 0) xl::atomic_or(p.status, ARCLE_ST_BAD_OP);
    out.term = r.term != 0;
    return out;
  }
  const int kind = (int)ARCLE_OP_KIND(desc), arg = (int)ARCLE_OP_ARG(desc);
  const uint32_t oflags = ARCLE_OP_FLAGS(desc);

  Planes s;
  s.wr = 0;
  s.have_grid = false;
  s.bytes = 2 * ARCLE_REC_BYTES + 24;  // record R/W + action in + reward/term out
  int submit_inc = 0;
  bool domain_error = false;
  int eq = -1;  // grid == answer, evaluated at most once (Submit and reward see the same state)

  Sel sel;
  ingest_selection(w, sel, payload);
  if (w.ingress == INGRESS_MASK) s.bytes += P;
#ifdef ARCLE_TRACE_WAVES
  w.t_sel = xl::clock();
#endif

  const Rec r_before = r;
  if (oflags & ARCLE_OPF_RESET_SEL) {  // object.py:20-25
    // with ARCLE_STEP_ELIDE_SELECTED an env that enters the step inactive is known to hold an all-zero `selected`
    // plane already (see include/arcle_hip.h): the zero-fill would rewrite zeros with zeros
    if (!((p.flags & ARCLE_STEP_ELIDE_SELECTED) && r.active == 0)) {
      s.selected = u4_zero();
      s.wr |= WR_SELECTED;
    }
    s.bytes += P;  // semantic accounting is unchanged
    r.active = 0;
  }
  if (oflags & ARCLE_OPF_KEEP_SEL) {  // object.py:36-40
    s.selected = sel_values(sel);
    if (!(s.wr & WR_SELECTED)) s.bytes += P;
    s.wr |= WR_SELECTED;
  }

  switch (kind) {  // transition(): self.operations[op](state, action)   o2arcenv.py:149-151
    case ARCLE_OP_COLOR: {  // color.py:70-74 — whole HxW plane, grid_dim ignored
      if (sel.any_nz) {
        need_grid(w, s);
        s.grid = u4_sel(expand16(sel.nz), u4_splat((uint32_t)arg), s.grid);
        s.wr |= WR_GRID;
        s.bytes += P;
      }
      break;
    }
    case ARCLE_OP_FLOODFILL:
      op_floodfill(w, s, r, sel, arg);
      break;
    case ARCLE_OP_MOVE: {  // object.py:230-240
      Lift L = init_objsel(w, s, r, sel);
      if (!L.ok) break;
      const int dx = (arg == 0) ? -1 : (arg == 1) ? 1 : 0;
      const int dy = (arg == 2) ? 1 : (arg == 3) ? -1 : 0;
      r.ox = i8w(r.ox + dx);  // :238, int8 wrap
      r.oy = i8w(r.oy + dy);
      place(w, s, r, w.lds->a, L.osel, L.S0, L.osel_full);
      break;
    }
    case ARCLE_OP_ROTATE:
    case ARCLE_OP_FLIP: {  // object.py:177-213 / :265-276
      // pre-compute the geometry so that an out-of-domain transform (the reference raises) skips the step
      int h, wd, x, y, xmin, xmax, ymin, ymax;
      bool fresh = sel.any_nz;
      if (fresh) {
        xmin = sel.x0; xmax = sel.x1; ymin = sel.y0; ymax = sel.y1;
        h = xmax - xmin + 1; wd = ymax - ymin + 1; x = xmin; y = ymin;
      } else if (r.active) {
        h = r.oh; wd = r.ow; x = r.ox; y = r.oy;
        xmin = x; xmax = i8w(i8w(x + h) - 1); ymin = y; ymax = i8w(i8w(y + wd) - 1);  // :102-107
      } else {
        break;  // :110-111 total no-op
      }
      int nx = x, ny = y, nh = h, nw = wd, npar = fresh ? 0 : r.parity;
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
      if (w.fastw) {
        // ---- lean path (16 <= W <= 32): the transformed tile is gathered straight from the source plane
        //      (the grid for a fresh selection, the stored object when continuing) ------------------------
        const bool rect_sel = fresh && sel.is_rect;
        int S0 = 0;
        U4 src_sel = u4_zero();
        if (fresh) {  // _init_objsel, object.py:67-99, fused with the transform
          need_grid(w, s);
          U4 pm = expand16(sel.pos);
          s.background = u4_andn(s.grid, pm);
          w.stage(w.lds->a, rect_sel ? s.grid : u4_and(s.grid, pm));
          if (!rect_sel) {
#pragma unroll
            for (int i = 0; i < 4; i++) src_sel[i] = pm[i] & 0x01010101u;
          }
          S0 = xmin * W + ymin;
          r.ox = xmin; r.oy = ymin; r.oh = h; r.ow = wd; r.active = 1; r.parity = 0;
          s.wr |= WR_BACKGROUND;
          s.bytes += 3 * P;
        } else {  // object.py:102-107
          U4 so = w.load(ARCLE_PL_OBJECT);
          src_sel = w.load(ARCLE_PL_OBJECT_SEL);
          s.background = w.load(ARCLE_PL_BACKGROUND);
          s.bytes += 5 * P;
          w.stage(w.lds->a, so);
        }
        if (!rect_sel) w.stage(w.lds->b, src_sel);
        const int B0 = ai * w.r0 + bj * w.c0 + c0 + S0;
        const int B1 = ai * (w.r0 + 1) - bj * w.k1 + c0 + S0;
        const uint32_t orect = w.rect16(0, nh - 1, 0, nw - 1);
        const U4 ob = expand16(orect);
        s.object = u4_and(gather_affine(w, w.lds->a, B0, B1, bj), ob);
        if (rect_sel) {
#pragma unroll
          for (int i = 0; i < 4; i++) s.object_sel[i] = ob[i] & 0x01010101u;
        } else {
          s.object_sel = u4_and(gather_affine(w, w.lds->b, B0, B1, bj), ob);
        }
        s.wr |= WR_OBJECT | WR_OBJECT_SEL;
        if (kind == ARCLE_OP_ROTATE && (arg & 1)) {
          r.ox = nx; r.oy = ny; r.oh = nh; r.ow = nw; r.parity = npar;
        }
        w.stage(w.lds->a, s.object);
        // Flip D0/D1 leave object_dim = (h,w) while the tile is (w,h) (:270-273): only when the two agree is the
        // placed selection exactly the destination rectangle
        place(w, s, r, w.lds->a, rect_sel ? orect : nz16(s.object_sel), 0, rect_sel && nh == r.oh && nw == r.ow);
        break;
      }
      Lift L = init_objsel(w, s, r, sel);
      tile_transform(w, s, nh, nw, ai, bj, c0);
      if (kind == ARCLE_OP_ROTATE && (arg & 1)) {
        r.ox = nx; r.oy = ny; r.oh = nh; r.ow = nw; r.parity = npar;
      }
      if (!L.fresh) s.bytes += 2 * P;  // object, object_sel written back (already counted when fresh)
      w.stage(w.lds->a, s.object);
      place(w, s, r, w.lds->a, nz16(s.object_sel), 0);
      break;
    }
    case ARCLE_OP_COPY: {  // object.py:291-312
      if (!sel.any_pos) break;
      int ss_h = arg ? r.gh : r.in_h, ss_w = arg ? r.gw : r.in_w;
      if (sel.x1 > ss_h || sel.y1 > ss_w) break;  // :301 (sic: > not >=)
      U4 src;
      if (arg) {
        need_grid(w, s);
        src = s.grid;
      } else {
        src = w.load(ARCLE_PL_INPUT);
        s.bytes += P;
      }
      int h = sel.x1 - sel.x0 + 1, wd = sel.y1 - sel.y0 + 1;
      w.stage(w.lds->a, u4_and(src, expand16(sel.nz)));  // where=logical_and(src, sel)
      s.clip = u4_and(w.shifted(w.lds->a, sel.x0 * W + sel.y0), expand16(w.rect16(0, h - 1, 0, wd - 1)));
      r.ch = h;
      r.cw = wd;
      s.wr |= WR_CLIP;
      s.bytes += P;
      break;
    }
    case ARCLE_OP_PASTE: {  // object.py:317-348
      if (!sel.any_pos) break;
      int h = r.ch, wd = r.cw;
      if (h == 0 || wd == 0) break;  // :334
      int ex = imin(sel.x0 + h, p.H), ey = imin(sel.y0 + wd, p.W);  // :340-341 clipped to HxW, not grid_dim
      need_grid(w, s);
      s.clip = w.load(ARCLE_PL_CLIP);
      s.bytes += P;
      w.stage(w.lds->a, s.clip);
      U4 pc = w.shifted(w.lds->a, -(sel.x0 * W + sel.y0));
      uint32_t R = w.rect16(sel.x0, ex - 1, sel.y0, ey - 1);
      if (!arg) R &= pos16(pc);  // paste_blank=False: where=(patch>0)
      s.grid = u4_sel(expand16(R), pc, s.grid);
      s.wr |= WR_GRID;
      s.bytes += P;
      break;
    }
    case ARCLE_OP_COPY_FROM_INPUT: {  // critical.py:28-29
      s.grid = w.load(ARCLE_PL_INPUT);
      s.have_grid = true;
      r.gh = r.in_h;
      r.gw = r.in_w;
      s.wr |= WR_GRID;
      s.bytes += 2 * P;
      break;
    }
    case ARCLE_OP_RESET_GRID: {  // critical.py:17
      s.grid = u4_zero();
      s.have_grid = true;
      s.wr |= WR_GRID;
      s.bytes += P;
      break;
    }
    case ARCLE_OP_RESIZE_GRID: {  // critical.py:39-46
      if (!sel.any_nz) break;
      s.grid = u4_zero();
      s.have_grid = true;
      r.gh = sel.x1 - sel.x0 + 1;
      r.gw = sel.y1 - sel.y0 + 1;
      s.wr |= WR_GRID;
      s.bytes += P;
      break;
    }
    case ARCLE_OP_CROP_GRID: {  // critical.py:56-66
      if (!sel.any_nz) break;
      need_grid(w, s);
      int h = sel.x1 - sel.x0 + 1, wd = sel.y1 - sel.y0 + 1;
      w.stage(w.lds->a, u4_and(s.grid, expand16(sel.nz)));
      s.grid = u4_and(w.shifted(w.lds->a, sel.x0 * W + sel.y0), expand16(w.rect16(0, h - 1, 0, wd - 1)));
      r.gh = h;
      r.gw = wd;
      s.wr |= WR_GRID;
      s.bytes += P;
      break;
    }
    case ARCLE_OP_RESIZE_TO_ANSWER: {  // arcenv.py:31-35
      need_grid(w, s);
      r.gh = r.ah;
      r.gw = r.aw;
      s.grid = u4_and(s.grid, expand16(w.rect16(0, r.ah - 1, 0, r.aw - 1)));
      s.wr |= WR_GRID;
      s.bytes += P;
      break;
    }
    case ARCLE_OP_SUBMIT: {  // base.py:172-183 (reset_on_submit=False)
      if (r.trials != 0) {
        r.trials = i8w(r.trials - 1);  // :174 int8 wrap
        submit_inc = 1;
        eq = grid_equals_answer(w, s, r) ? 1 : 0;
        if (eq) r.term = 1;
      }
      if (r.trials == 0) r.term = 1;
      break;
    }
    default:
      break;
  }

  if (domain_error) {  // the reference raised inside the op: the step did not happen
    if (lane == 0) xl::atomic_or(p.status, ARCLE_ST_ROTATE_DOMAIN);
    r = r_before;
    out.term = r.term != 0;
    return out;
  }

  // reward(): only the LAST op of the table can be rewarded (o2arcenv.py:121-128)
  int reward = 0;
  if (op == p.n_ops - 1) {
    if (eq < 0) eq = grid_equals_answer(w, s, r) ? 1 : 0;
    reward = eq;
  }

  // write back the planes the op changed
  xl::lds_fence();
  if (s.wr & WR_GRID) w.store(ARCLE_PL_GRID, s.grid);
  if (s.wr & WR_SELECTED) w.store(ARCLE_PL_SELECTED, s.selected);
  if (s.wr & WR_CLIP) w.store(ARCLE_PL_CLIP, s.clip);
  if (s.wr & WR_OBJECT) w.store(ARCLE_PL_OBJECT, s.object);
  if (s.wr & WR_OBJECT_SEL) w.store(ARCLE_PL_OBJECT_SEL, s.object_sel);
  if (s.wr & WR_BACKGROUND) w.store(ARCLE_PL_BACKGROUND, s.background);
  cnt0.x += 1;  // o2arcenv.py:142
  cnt0.y += submit_inc;
  out.reward = reward;
  out.term = r.term != 0;
  out.bytes = s.bytes;
  return out;
}

template <int ING, int FW, int TBL>
ARCLE_DEV void wave_step(const StepParams& p_in, WaveLDS* lds, int env, int lane) {
  // fetch every kernel argument the step needs in one burst at wave start (instead of lazily, one cold constant-cache
  // miss at a time along the critical path) and keep them in SGPRs: 8.3 -> 7.8 us per launch
  StepParams p = p_in;
#pragma unroll
  for (int i = 0; i < ARCLE_N_PLANES; i++) xl::pin_ptr(p.plane[i]);
  xl::pin_ptr(p.reward);
  xl::pin_ptr(p.term);
  xl::pin_ptr(p.acct);
  xl::pin_ptr(p.status);
  xl::pin_u32(p.flags);
  xl::pin_i32(p.H);
  xl::pin_i32(p.W);
  xl::pin_i32(p.P);
  xl::pin_i32(p.PS);
  xl::pin_i32(p.n_ops);
  xl::pin_u32(p.div_magic);
#ifdef ARCLE_TRACE_WAVES  // diagnostic build: per-wave start/end shader clocks into the acct buffer (as uint64[N][2])
  const uint64_t t_start = xl::clock();
#endif
  Wave w(p, lds, env, lane, ING, FW);
  // ---- one latency window: record, op index, counters and the selection payload are independent loads ----
  U4 rv = *reinterpret_cast<const U4*>(p.rec + (size_t)env * ARCLE_REC_BYTES);
  uint32_t opv = (uint32_t)p.op[env];
  I2 cnt0 = *reinterpret_cast<const I2*>(p.cnt + 2 * (size_t)env);
  U4 payload = load_payload(w);
  xl::keep(rv, payload, opv, cnt0.x);  // all four are in flight before the first use
#pragma unroll
  for (int i = 0; i < 4; i++) rv[i] = xl::uniform(rv[i]);
  Rec r;
  rec_unpack(rv, r);
  const int op = (int)xl::uniform(opv);
#ifdef ARCLE_TRACE_WAVES
  const uint64_t t_win1 = xl::clock();  // first latency window complete
#endif
  StepOut out = step_core<ING, FW, TBL>(w, r, cnt0, payload, op);
#ifdef ARCLE_TRACE_WAVES
  const uint64_t t_core = xl::clock();  // op applied, plane stores issued
#endif
  // ---- epilogue: record, counters and the step outputs --------------------------------------------
  if (lane == 0) {
    *reinterpret_cast<U4*>(p.rec + (size_t)env * ARCLE_REC_BYTES) = rec_pack(r);
    *reinterpret_cast<I2*>(p.cnt + 2 * (size_t)env) = cnt0;
    p.reward[env] = out.reward;
    p.term[env] = (uint8_t)out.term;
#ifdef ARCLE_TRACE_WAVES
    if (p.acct) {
      uint64_t* tr = reinterpret_cast<uint64_t*>(p.acct) + 8 * (size_t)env;
      tr[0] = t_start;
      tr[1] = t_win1;
      tr[2] = t_core;
      tr[3] = xl::clock();
      tr[4] = w.t_desc;
      tr[5] = w.t_sel;
    }
#else
    if (p.acct) p.acct[env] += out.bytes;
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// n_steps consecutive step()s of one env in ONE launch (a rollout / trace replay: the caller already holds the
// whole action sequence, e.g. tests/o2arc_check.py:139-199 of the reference or a scripted policy).  The env's
// planes and record live in registers for the whole rollout: HBM sees the planes once in and (if changed) once
// out, plus 20 B of action in and 5 B of reward/terminated out per step.
//   sel: int32 [n_steps][n_envs][4|2]   op: int32 [n_steps][n_envs]
//   reward: int32 [n_steps][n_envs]     term: uint8 [n_steps][n_envs]
// ------------------------------------------------------------------------------------------------
template <int ING, int FW, int TBL>
ARCLE_DEV void wave_rollout(const StepParams& p, WaveLDS* lds, int env, int lane) {
  Wave w(p, lds, env, lane, ING, FW);
#pragma unroll
  for (int pl = 0; pl < ARCLE_N_PLANES; pl++) w.cache[pl] = p.plane[pl] ? w.load_hbm(pl) : u4_zero();
  w.resident = true;
  Rec r;
  rec_unpack(load_rec(p, env), r);
  I2 cnt = *reinterpret_cast<const I2*>(p.cnt + 2 * (size_t)env);
  const size_t N = (size_t)p.n_envs;
  U4 next_payload = load_payload(w, 0);
  uint32_t next_op = (uint32_t)p.op[env];
  for (int t = 0; t < p.n_steps; t++) {
    const U4 payload = next_payload;
    const int op = (int)xl::uniform(next_op);
    if (t + 1 < p.n_steps) {  // the next action is in flight while this one executes
      next_payload = load_payload(w, (size_t)t + 1);
      next_op = (uint32_t)p.op[((size_t)t + 1) * N + env];
    }
    const StepOut out = step_core<ING, FW, TBL>(w, r, cnt, payload, op);
    if (lane == 0) {
      p.reward[(size_t)t * N + env] = out.reward;
      p.term[(size_t)t * N + env] = (uint8_t)out.term;
    }
  }
#pragma unroll
  for (int pl = 0; pl < ARCLE_N_PLANES; pl++)
    if (w.dirty & (1u << pl)) w.store_hbm(pl, w.cache[pl]);
  store_rec_cnt(p, env, lane, r, cnt);
}

// ------------------------------------------------------------------------------------------------
// flattened observation row of one env (Gymnasium FlattenObservation key order; GPTPolicy.py:17-35)
// ------------------------------------------------------------------------------------------------
ARCLE_DEV void flat_plane(const Wave& w, int8_t* row, int& off, int pl) {
  if (!w.p.plane[pl]) return;
  U4 v = w.load_hbm(pl);
  int8_t* d = row + off + 16 * w.lane;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if ((w.valid16 >> k) & 1u) d[k] = (int8_t)u4_byte(v, k);
  off += w.p.P;
}
ARCLE_DEV void flat_scalar(const Wave& w, int8_t* row, int& off, const int8_t* rec, int field, int n) {
  if (w.lane < n) row[off + w.lane] = rec[field + w.lane];
  off += n;
}
ARCLE_DEV void wave_flatten(const StepParams& p, WaveLDS* lds, int env, int lane) {
  Wave w(p, lds, env, lane, INGRESS_BBOX, 0);
  int8_t* row = p.flat_out + (size_t)env * p.flat_len;
  const int8_t* rec = p.rec + (size_t)env * ARCLE_REC_BYTES;
  const bool o2 = p.plane[ARCLE_PL_SELECTED] != nullptr, clip = p.plane[ARCLE_PL_CLIP] != nullptr;
  int off = 0;
  flat_plane(w, row, off, ARCLE_PL_CLIP);
  if (clip) flat_scalar(w, row, off, rec, ARCLE_REC_CLIP_DIM, 2);
  flat_plane(w, row, off, ARCLE_PL_GRID);
  flat_scalar(w, row, off, rec, ARCLE_REC_GRID_DIM, 2);
  flat_plane(w, row, off, ARCLE_PL_INPUT);
  flat_scalar(w, row, off, rec, ARCLE_REC_INPUT_DIM, 2);
  if (o2) {
    flat_scalar(w, row, off, rec, ARCLE_REC_ACTIVE, 1);
    flat_plane(w, row, off, ARCLE_PL_BACKGROUND);
    flat_plane(w, row, off, ARCLE_PL_OBJECT);
    flat_scalar(w, row, off, rec, ARCLE_REC_OBJECT_DIM, 2);
    flat_scalar(w, row, off, rec, ARCLE_REC_OBJECT_POS, 2);
    flat_plane(w, row, off, ARCLE_PL_OBJECT_SEL);
    flat_scalar(w, row, off, rec, ARCLE_REC_PARITY, 1);
    flat_plane(w, row, off, ARCLE_PL_SELECTED);
  }
  flat_scalar(w, row, off, rec, ARCLE_REC_TERMINATED, 1);
  flat_scalar(w, row, off, rec, ARCLE_REC_TRIALS, 1);
}

}  // namespace arcle
