/*
 * arcle_oracle.c — CPU restatement (plain C, scalar loops) of the reference's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under arcle_amd/ may import, link or call this file;
 * it exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check
 * (and time) the algorithm the HIP kernels must reproduce bit-for-bit.
 *
 * Parity status: PINNED.  The reference ships no usable golden vectors for this path
 * (its trace pickles are in .MISSING_LARGE_BLOBS), so the pin is
 *   (1) oracle/diff_vs_reference.py — imports the unmodified reference from /root/reference
 *       (with the gymnasium/pygame stubs of oracle/stubs) in the build container and
 *       differential-fuzzes every op of every env kind against this file, and
 *   (2) tests/golden/ (.npz files) — input/output vectors captured from that same reference import
 *       by tests/golden/make_golden.py, replayed against this file by tests/test_oracle_golden.py.
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference).
 * The data layout (planes / rec / cnt) is the one declared in include/arcle_hip.h so that
 * oracle state and device state can be compared array-for-array.
 *
 * Deliberately written per-cell / per-env with 2-D loops and an explicit-stack DFS: it shares
 * no code and no data-parallel structure with the HIP kernels.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/arcle_hip.h"

/* The oracle's own bound: any H x W with H, W <= 127 (grid dims are int8 in the reference's state dict, base.py:162-166), i.e. also
 * the grids beyond ARCLE_MAX_CELLS that the HIP library steps with one workgroup per env (arcle_big.hip). */
#define ORACLE_MAX_CELLS (127 * 127)

typedef struct oracle_env {
  int32_t n_envs, H, W;
  int32_t max_trial;
  int32_t n_ops;
  uint32_t ops[ARCLE_MAX_OPS];
  int8_t* plane[ARCLE_N_PLANES]; /* host pointers, [n_envs][H*W], NULL if absent */
  int8_t* rec;                   /* [n_envs][16] */
  int32_t* cnt;                  /* [n_envs][2]  */
  uint32_t status;               /* ARCLE_ST_* sticky bits */
} oracle_env;

/* one env's view */
typedef struct view {
  int H, W;
  int8_t *input, *grid, *selected, *clip, *object, *object_sel, *background, *answer;
  int8_t* rec;
  int32_t* cnt;
} view;

#define AT(p, i, j) ((p)[(i) * v->W + (j)])
static inline int8_t i8(int x) { return (int8_t)(uint8_t)(x & 0xff); } /* wrap mod 256 */

static view make_view(const oracle_env* e, int n) {
  view v;
  size_t off = (size_t)n * e->H * e->W;
  v.H = e->H;
  v.W = e->W;
#define PL(id) (e->plane[id] ? e->plane[id] + off : NULL)
  v.input = PL(ARCLE_PL_INPUT);
  v.grid = PL(ARCLE_PL_GRID);
  v.selected = PL(ARCLE_PL_SELECTED);
  v.clip = PL(ARCLE_PL_CLIP);
  v.object = PL(ARCLE_PL_OBJECT);
  v.object_sel = PL(ARCLE_PL_OBJECT_SEL);
  v.background = PL(ARCLE_PL_BACKGROUND);
  v.answer = PL(ARCLE_PL_ANSWER);
#undef PL
  v.rec = e->rec + (size_t)n * ARCLE_REC_BYTES;
  v.cnt = e->cnt + (size_t)n * 2;
  return v;
}

/* ---- helpers -------------------------------------------------------------------------- */

/* np.any(sel)  — truthiness, used by color.py:72, object.py:67, critical.py:39,56 */
static int any_truthy(const view* v, const int8_t* sel) {
  for (int k = 0; k < v->H * v->W; k++)
    if (sel[k] != 0) return 1;
  return 0;
}
/* np.any(sel>0) — object.py:294, :326 */
static int any_positive(const view* v, const int8_t* sel) {
  for (int k = 0; k < v->H * v->W; k++)
    if (sel[k] > 0) return 1;
  return 0;
}

/* _get_bbox, object.py:49-58: first/last row and column holding a truthy cell. */
static void get_bbox(const view* v, const int8_t* img, int* xmin, int* xmax, int* ymin, int* ymax) {
  int x0 = -1, x1 = -1, y0 = -1, y1 = -1;
  for (int i = 0; i < v->H; i++) {
    int any = 0;
    for (int j = 0; j < v->W; j++) any |= (AT(img, i, j) != 0);
    if (any) {
      if (x0 < 0) x0 = i;
      x1 = i;
    }
  }
  for (int j = 0; j < v->W; j++) {
    int any = 0;
    for (int i = 0; i < v->H; i++) any |= (AT(img, i, j) != 0);
    if (any) {
      if (y0 < 0) y0 = j;
      y1 = j;
    }
  }
  *xmin = x0;
  *xmax = x1;
  *ymin = y0;
  *ymax = y1;
}

/* reset_sel wrapper, object.py:20-25 */
static void reset_sel(view* v) {
  memset(v->selected, 0, (size_t)v->H * v->W);
  v->rec[ARCLE_REC_ACTIVE] = 0;
}
/* keep_sel wrapper, object.py:36-40 */
static void keep_sel(view* v, const int8_t* sel) { memcpy(v->selected, sel, (size_t)v->H * v->W); }

/* ---- color.py ------------------------------------------------------------------------- */

/* gen_color, color.py:70-74: masked fill over the whole HxW plane (grid_dim ignored). */
static void op_color(view* v, const int8_t* sel, int c) {
  if (!any_truthy(v, sel)) return;
  for (int k = 0; k < v->H * v->W; k++)
    if (sel[k] != 0) v->grid[k] = (int8_t)c;
}

/* gen_flood_fill, color.py:88-100 with dfs color.py:8-30 (explicit stack instead of Python
 * recursion; the visited set of a DFS does not depend on the visiting order). */
static void op_floodfill(view* v, const int8_t* sel, int c) {
  long sum = 0; /* np.sum(sel) promotes int8 to the platform int */
  for (int k = 0; k < v->H * v->W; k++) sum += sel[k];
  if (sum != 1) return;
  int best = 0; /* np.argmax: first occurrence of the maximum */
  for (int k = 1; k < v->H * v->W; k++)
    if (sel[k] > sel[best]) best = k;
  int x = best / v->W, y = best % v->W;
  int gh = v->rec[ARCLE_REC_GRID_DIM], gw = v->rec[ARCLE_REC_GRID_DIM + 1];
  if (x >= gh || y >= gw) return;
  uint8_t visit[ORACLE_MAX_CELLS];
  int16_t stack[ORACLE_MAX_CELLS * 4 + 4];
  memset(visit, 0, (size_t)v->H * v->W);
  int8_t col = AT(v->grid, x, y);
  int sp = 0;
  stack[sp++] = (int16_t)(x * v->W + y);
  static const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
  while (sp > 0) {
    int k = stack[--sp];
    if (visit[k]) continue;
    visit[k] = 1;
    int cx = k / v->W, cy = k % v->W;
    for (int d = 0; d < 4; d++) {
      int nx = cx + dx[d], ny = cy + dy[d];
      if (nx >= 0 && nx < gh && ny >= 0 && ny < gw && AT(v->grid, nx, ny) == col &&
          !visit[nx * v->W + ny])
        stack[sp++] = (int16_t)(nx * v->W + ny);
    }
  }
  for (int k = 0; k < v->H * v->W; k++)
    if (visit[k]) v->grid[k] = (int8_t)c;
}

/* ---- object.py ------------------------------------------------------------------------ */

/* _init_objsel, object.py:60-111.  Returns 1 and the bbox if the op proceeds, 0 for the
 * "inactive and nothing selected" no-op (object.py:110-111). */
static int init_objsel(view* v, const int8_t* sel, int* xmin, int* xmax, int* ymin, int* ymax) {
  int8_t* r = v->rec;
  if (any_truthy(v, sel)) {
    get_bbox(v, sel, xmin, xmax, ymin, ymax);
    int h = *xmax - *xmin + 1, w = *ymax - *ymin + 1;
    r[ARCLE_REC_OBJECT_DIM] = (int8_t)h;
    r[ARCLE_REC_OBJECT_DIM + 1] = (int8_t)w;
    memset(v->object, 0, (size_t)v->H * v->W);
    memset(v->object_sel, 0, (size_t)v->H * v->W);
    for (int i = 0; i < h; i++)
      for (int j = 0; j < w; j++) {
        int part = AT(sel, *xmin + i, *ymin + j) > 0; /* object.py:78 */
        if (part) {
          AT(v->object, i, j) = AT(v->grid, *xmin + i, *ymin + j); /* :81 */
          AT(v->object_sel, i, j) = 1;                             /* :84 */
        }
      }
    for (int k = 0; k < v->H * v->W; k++) v->background[k] = (sel[k] > 0) ? 0 : v->grid[k]; /* :87-88 */
    r[ARCLE_REC_OBJECT_POS] = (int8_t)*xmin;
    r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)*ymin;
    r[ARCLE_REC_ACTIVE] = 1;
    r[ARCLE_REC_PARITY] = 0;
    memcpy(v->selected, sel, (size_t)v->H * v->W); /* :96 (values kept, cast to int8) */
    return 1;
  }
  if (r[ARCLE_REC_ACTIVE]) { /* :102-107, int8 arithmetic */
    int x = r[ARCLE_REC_OBJECT_POS], y = r[ARCLE_REC_OBJECT_POS + 1];
    int h = r[ARCLE_REC_OBJECT_DIM], w = r[ARCLE_REC_OBJECT_DIM + 1];
    *xmin = x;
    *xmax = i8(i8(x + h) - 1);
    *ymin = y;
    *ymax = i8(i8(y + w) - 1);
    return 1;
  }
  return 0;
}

/* _apply_patch (object.py:113-138) and _apply_sel (object.py:140-165). */
static void apply_patch_and_sel(view* v) {
  int8_t* r = v->rec;
  int x = r[ARCLE_REC_OBJECT_POS], y = r[ARCLE_REC_OBJECT_POS + 1];
  int h = r[ARCLE_REC_OBJECT_DIM], w = r[ARCLE_REC_OBJECT_DIM + 1];
  int gh = r[ARCLE_REC_GRID_DIM], gw = r[ARCLE_REC_GRID_DIM + 1];
  memcpy(v->grid, v->background, (size_t)v->H * v->W); /* :125 */
  memset(v->selected, 0, (size_t)v->H * v->W);         /* :153 */
  int xh = i8(x + h), yw = i8(y + w);                   /* int8 + int8 */
  if (xh > 0 && x < gh && yw > 0 && y < gw) {
    int stx = x > 0 ? x : 0, edx = gh < xh ? gh : xh;
    int sty = y > 0 ? y : 0, edy = gw < yw ? gw : yw;
    for (int i = stx; i < edx; i++)
      for (int j = sty; j < edy; j++) {
        int8_t p = AT(v->object, i - x, j - y);
        if (p > 0) AT(v->grid, i, j) = p;                       /* :138 where=(p>0) */
        AT(v->selected, i, j) = AT(v->object_sel, i - x, j - y); /* :165 */
      }
  }
}

/* tile transform used by Rotate / Flip: dst[:nh,:nw] = f(src[:h,:w]), rest 0 (_pad_assign :43-47) */
enum { T_ROT90 = 1, T_ROT180 = 2, T_ROT270 = 3, T_FLIPH = 4, T_FLIPV = 5, T_D0 = 6, T_D1 = 7 };
static int tile_transform(view* v, int8_t* plane, int h, int w, int t) {
  int8_t tmp[ORACLE_MAX_CELLS];
  int transposing = (t == T_ROT90 || t == T_ROT270 || t == T_D0 || t == T_D1);
  int nh = transposing ? w : h, nw = transposing ? h : w;
  if (nh > v->H || nw > v->W) return -1; /* reference: ValueError at object.py:45 */
  memset(tmp, 0, (size_t)v->H * v->W);
  for (int i = 0; i < nh; i++)
    for (int j = 0; j < nw; j++) {
      int si, sj;
      switch (t) {
        case T_ROT90: si = j; sj = w - 1 - i; break;          /* np.rot90(x,1)[i,j] = x[j, w-1-i] */
        case T_ROT180: si = h - 1 - i; sj = w - 1 - j; break; /* np.rot90(x,2) */
        case T_ROT270: si = h - 1 - j; sj = i; break;         /* np.rot90(x,3)[i,j] = x[h-1-j, i] */
        case T_FLIPH: si = i; sj = w - 1 - j; break;          /* np.fliplr */
        case T_FLIPV: si = h - 1 - i; sj = j; break;          /* np.flipud */
        case T_D0: si = j; sj = i; break;                     /* rot90(fliplr(x)) = x.T */
        default: si = h - 1 - j; sj = w - 1 - i; break;       /* fliplr(rot90(x)) = anti-transpose */
      }
      tmp[i * v->W + j] = AT(plane, si, sj);
    }
  memcpy(plane, tmp, (size_t)v->H * v->W);
  return 0;
}

static int floordiv2(int a) { return (a >= 0) ? a / 2 : -((-a + 1) / 2); }
static int imin(int a, int b) { return a < b ? a : b; }

/* snapshot / restore of one env, used where the reference raises half-way through an op */
typedef struct snap {
  int8_t planes[4][ORACLE_MAX_CELLS];
  int8_t rec[ARCLE_REC_BYTES];
} snap;
static void snap_take(const view* v, snap* s) {
  size_t P = (size_t)v->H * v->W;
  memcpy(s->planes[0], v->selected, P);
  memcpy(s->planes[1], v->object, P);
  memcpy(s->planes[2], v->object_sel, P);
  memcpy(s->planes[3], v->background, P);
  memcpy(s->rec, v->rec, ARCLE_REC_BYTES);
}
static void snap_restore(view* v, const snap* s) {
  size_t P = (size_t)v->H * v->W;
  memcpy(v->selected, s->planes[0], P);
  memcpy(v->object, s->planes[1], P);
  memcpy(v->object_sel, s->planes[2], P);
  memcpy(v->background, s->planes[3], P);
  memcpy(v->rec, s->rec, ARCLE_REC_BYTES);
}

/* gen_rotate(k), object.py:177-213.  All quantities of the float centre arithmetic
 * (:187-206) are multiples of 0.5, so it is evaluated exactly on doubled integers.
 * Returns -1 (ARCLE_ST_ROTATE_DOMAIN set; step_one restores the pre-step state) where the reference raises:
 * tile does not fit HxW (ValueError at :45 via :210) or the new position is not an int8. */
static int op_rotate(view* v, const int8_t* sel, int k, uint32_t* status) {
  int xmin, xmax, ymin, ymax;
  int8_t* r = v->rec;
  int fresh = any_truthy(v, sel);
  if (!init_objsel(v, sel, &xmin, &xmax, &ymin, &ymax)) return 0;
  int h = r[ARCLE_REC_OBJECT_DIM], w = r[ARCLE_REC_OBJECT_DIM + 1];
  int x = r[ARCLE_REC_OBJECT_POS], y = r[ARCLE_REC_OBJECT_POS + 1];
  if (k % 2 != 0) {
    /* cx = (xmax + xmin) * 0.5: np.int64 sums after a fresh selection, int8 (wrapping) sums when
     * continuing with the stored object (:102-107) */
    int sx2 = fresh ? xmax + xmin : i8(xmax + xmin);
    int sy2 = fresh ? ymax + ymin : i8(ymax + ymin);
    int nx, ny;
    if ((h & 1) == (w & 1)) { /* :186-192 */
      nx = floordiv2(sx2 - sy2 + 2 * y);
      ny = floordiv2(sy2 - sx2 + 2 * x);
    } else { /* :195-206 */
      r[ARCLE_REC_PARITY] = (int8_t)((r[ARCLE_REC_PARITY] + k) % 2);
      int sig = (k + 2) % 4 - 2;
      int mod = 1 - r[ARCLE_REC_PARITY];
      int mx2 = sx2 + imin(sig * (sy2 - 2 * ymin), sig * (sy2 - 2 * ymax)) + 2 * mod;
      int my2 = sy2 + imin(-sig * (sx2 - 2 * xmin), -sig * (sx2 - 2 * xmax)) + 2 * mod;
      nx = floordiv2(mx2);
      ny = floordiv2(my2);
    }
    if (w > v->H || h > v->W || nx < -128 || nx > 127 || ny < -128 || ny > 127) {
      __atomic_fetch_or(status, ARCLE_ST_ROTATE_DOMAIN, __ATOMIC_RELAXED);
      return -1; /* caller restores the pre-step state */
    }
    r[ARCLE_REC_OBJECT_POS] = (int8_t)nx;
    r[ARCLE_REC_OBJECT_POS + 1] = (int8_t)ny;
    r[ARCLE_REC_OBJECT_DIM] = (int8_t)w;
    r[ARCLE_REC_OBJECT_DIM + 1] = (int8_t)h;
  }
  int t = (k == 1) ? T_ROT90 : (k == 2) ? T_ROT180 : T_ROT270;
  tile_transform(v, v->object, h, w, t);     /* :210 */
  tile_transform(v, v->object_sel, h, w, t); /* :211 */
  apply_patch_and_sel(v);                    /* :212-213 */
  return 0;
}

/* gen_move(d), object.py:230-240 */
static void op_move(view* v, const int8_t* sel, int d) {
  static const int dirX[4] = {-1, +1, 0, 0}, dirY[4] = {0, 0, +1, -1};
  int a, b, c, e;
  if (!init_objsel(v, sel, &a, &b, &c, &e)) return;
  int8_t* r = v->rec;
  r[ARCLE_REC_OBJECT_POS] = i8(r[ARCLE_REC_OBJECT_POS] + dirX[d]);         /* :238 int8 wrap */
  r[ARCLE_REC_OBJECT_POS + 1] = i8(r[ARCLE_REC_OBJECT_POS + 1] + dirY[d]);
  apply_patch_and_sel(v);
}

/* gen_flip(axis), object.py:265-276.  object_dim is NOT updated, also for D0/D1 (:270-273). */
static int op_flip(view* v, const int8_t* sel, int axis, uint32_t* status) {
  int a, b, c, e;
  int8_t* r = v->rec;
  if (!init_objsel(v, sel, &a, &b, &c, &e)) return 0;
  int h = r[ARCLE_REC_OBJECT_DIM], w = r[ARCLE_REC_OBJECT_DIM + 1];
  if (axis >= 2 && (w > v->H || h > v->W)) { /* D0/D1 transpose the tile: ValueError at :45 */
    __atomic_fetch_or(status, ARCLE_ST_ROTATE_DOMAIN, __ATOMIC_RELAXED);
    return -1;
  }
  int t = axis == 0 ? T_FLIPH : axis == 1 ? T_FLIPV : axis == 2 ? T_D0 : T_D1;
  tile_transform(v, v->object, h, w, t);
  tile_transform(v, v->object_sel, h, w, t);
  apply_patch_and_sel(v);
  return 0;
}

/* gen_copy(source), object.py:291-312 */
static void op_copy(view* v, const int8_t* sel, int src_is_grid) {
  if (!any_positive(v, sel)) return; /* :294 */
  int xmin, xmax, ymin, ymax;
  get_bbox(v, sel, &xmin, &xmax, &ymin, &ymax);
  int8_t* r = v->rec;
  int ss_h = r[src_is_grid ? ARCLE_REC_GRID_DIM : ARCLE_REC_INPUT_DIM];
  int ss_w = r[(src_is_grid ? ARCLE_REC_GRID_DIM : ARCLE_REC_INPUT_DIM) + 1];
  if (xmax > ss_h || ymax > ss_w) return; /* :301 (sic: > not >=) */
  int h = xmax - xmin + 1, w = ymax - ymin + 1;
  const int8_t* src = src_is_grid ? v->grid : v->input;
  int8_t tmp[ORACLE_MAX_CELLS];
  memset(tmp, 0, (size_t)v->H * v->W); /* :307 */
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++) {
      int8_t s = AT(src, xmin + i, ymin + j);
      if (s != 0 && AT(sel, xmin + i, ymin + j) != 0) tmp[i * v->W + j] = s; /* :310-312 */
    }
  memcpy(v->clip, tmp, (size_t)v->H * v->W);
  r[ARCLE_REC_CLIP_DIM] = (int8_t)h; /* :308 */
  r[ARCLE_REC_CLIP_DIM + 1] = (int8_t)w;
}

/* gen_paste(paste_blank), object.py:317-348 */
static void op_paste(view* v, const int8_t* sel, int paste_blank) {
  if (!any_positive(v, sel)) return; /* :326 */
  int xmin, xmax, ymin, ymax;
  get_bbox(v, sel, &xmin, &xmax, &ymin, &ymax);
  int8_t* r = v->rec;
  int h = r[ARCLE_REC_CLIP_DIM], w = r[ARCLE_REC_CLIP_DIM + 1];
  if (xmin >= v->H || ymin >= v->W || h == 0 || w == 0) return; /* :334 */
  int edx = xmin + h < v->H ? xmin + h : v->H;                   /* :340-341: clipped to HxW */
  int edy = ymin + w < v->W ? ymin + w : v->W;
  for (int i = xmin; i < edx; i++)
    for (int j = ymin; j < edy; j++) {
      int8_t p = AT(v->clip, i - xmin, j - ymin);
      if (paste_blank || p > 0) AT(v->grid, i, j) = p; /* :345-348 */
    }
}

/* ---- critical.py / arcenv.py ----------------------------------------------------------- */

static void op_reset_grid(view* v) { memset(v->grid, 0, (size_t)v->H * v->W); } /* critical.py:17 */

static void op_copy_from_input(view* v) { /* critical.py:28-29 */
  v->rec[ARCLE_REC_GRID_DIM] = v->rec[ARCLE_REC_INPUT_DIM];
  v->rec[ARCLE_REC_GRID_DIM + 1] = v->rec[ARCLE_REC_INPUT_DIM + 1];
  memcpy(v->grid, v->input, (size_t)v->H * v->W);
}

static void op_resize_grid(view* v, const int8_t* sel) { /* critical.py:39-46 */
  if (!any_truthy(v, sel)) return;
  int xmin, xmax, ymin, ymax;
  get_bbox(v, sel, &xmin, &xmax, &ymin, &ymax);
  memset(v->grid, 0, (size_t)v->H * v->W);
  v->rec[ARCLE_REC_GRID_DIM] = (int8_t)(xmax - xmin + 1);
  v->rec[ARCLE_REC_GRID_DIM + 1] = (int8_t)(ymax - ymin + 1);
}

static void op_crop_grid(view* v, const int8_t* sel) { /* critical.py:56-66 */
  if (!any_truthy(v, sel)) return;
  int xmin, xmax, ymin, ymax;
  get_bbox(v, sel, &xmin, &xmax, &ymin, &ymax);
  int h = xmax - xmin + 1, w = ymax - ymin + 1;
  int8_t tmp[ORACLE_MAX_CELLS];
  memset(tmp, 0, (size_t)v->H * v->W);
  for (int i = 0; i < h; i++)
    for (int j = 0; j < w; j++) {
      int8_t g = AT(v->grid, xmin + i, ymin + j);
      if (AT(sel, xmin + i, ymin + j) != 0 && g != 0) tmp[i * v->W + j] = g;
    }
  memcpy(v->grid, tmp, (size_t)v->H * v->W);
  v->rec[ARCLE_REC_GRID_DIM] = (int8_t)h;
  v->rec[ARCLE_REC_GRID_DIM + 1] = (int8_t)w;
}

static void op_resize_to_answer(view* v) { /* arcenv.py:31-35 */
  int h = v->rec[ARCLE_REC_ANSWER_DIM], w = v->rec[ARCLE_REC_ANSWER_DIM + 1];
  v->rec[ARCLE_REC_GRID_DIM] = (int8_t)h;
  v->rec[ARCLE_REC_GRID_DIM + 1] = (int8_t)w;
  for (int i = 0; i < v->H; i++)
    for (int j = 0; j < v->W; j++)
      if (i >= h || j >= w) AT(v->grid, i, j) = 0;
}

/* answer.shape == grid_dim and grid[:h,:w] == answer  (base.py:177, o2arcenv.py:124-127) */
static int grid_equals_answer(const view* v) {
  int gh = v->rec[ARCLE_REC_GRID_DIM], gw = v->rec[ARCLE_REC_GRID_DIM + 1];
  if (gh != v->rec[ARCLE_REC_ANSWER_DIM] || gw != v->rec[ARCLE_REC_ANSWER_DIM + 1]) return 0;
  for (int i = 0; i < gh; i++)
    for (int j = 0; j < gw; j++)
      if (AT(v->grid, i, j) != AT(v->answer, i, j)) return 0;
  return 1;
}

/* AbstractARCEnv.submit, base.py:172-183 (reset_on_submit=False, the default) */
static void op_submit(view* v) {
  int8_t* r = v->rec;
  if (r[ARCLE_REC_TRIALS] != 0) {
    r[ARCLE_REC_TRIALS] = i8(r[ARCLE_REC_TRIALS] - 1); /* :174 int8 wrap */
    v->cnt[ARCLE_CNT_SUBMIT] += 1;
    if (grid_equals_answer(v)) r[ARCLE_REC_TERMINATED] = 1;
  }
  if (r[ARCLE_REC_TRIALS] == 0) r[ARCLE_REC_TERMINATED] = 1; /* :182-183 */
}

/* init_state: base.py:155-166 + o2arcenv.py:16-34 / arcenv.py:81-89; counters base.py:73-79 */
static void init_state(const oracle_env* e, view* v) {
  size_t P = (size_t)v->H * v->W;
  memcpy(v->grid, v->input, P);
  if (v->selected) memset(v->selected, 0, P);
  if (v->clip) memset(v->clip, 0, P);
  if (v->object) memset(v->object, 0, P);
  if (v->object_sel) memset(v->object_sel, 0, P);
  if (v->background) memset(v->background, 0, P);
  int8_t* r = v->rec;
  r[ARCLE_REC_GRID_DIM] = r[ARCLE_REC_INPUT_DIM];
  r[ARCLE_REC_GRID_DIM + 1] = r[ARCLE_REC_INPUT_DIM + 1];
  r[ARCLE_REC_CLIP_DIM] = r[ARCLE_REC_CLIP_DIM + 1] = 0;
  r[ARCLE_REC_OBJECT_DIM] = r[ARCLE_REC_OBJECT_DIM + 1] = 0;
  r[ARCLE_REC_OBJECT_POS] = r[ARCLE_REC_OBJECT_POS + 1] = 0;
  r[ARCLE_REC_TRIALS] = i8(e->max_trial);
  r[ARCLE_REC_TERMINATED] = 0;
  r[ARCLE_REC_ACTIVE] = 0;
  r[ARCLE_REC_PARITY] = 0;
  v->cnt[ARCLE_CNT_STEPS] = 0;
  v->cnt[ARCLE_CNT_SUBMIT] = 0;
}

/* ---- public entry points (called through ctypes by oracle/oracle.py) -------------------- */

oracle_env* oracle_create(int n_envs, int H, int W, int max_trial) {
  if (n_envs <= 0 || H <= 0 || W <= 0 || H * W > ORACLE_MAX_CELLS || H > 127 || W > 127) return NULL;
  oracle_env* e = (oracle_env*)calloc(1, sizeof *e);
  e->n_envs = n_envs;
  e->H = H;
  e->W = W;
  e->max_trial = max_trial;
  return e;
}
void oracle_destroy(oracle_env* e) { free(e); }

int oracle_bind(oracle_env* e, int8_t** planes, int8_t* rec, int32_t* cnt) {
  for (int p = 0; p < ARCLE_N_PLANES; p++) e->plane[p] = planes[p];
  e->rec = rec;
  e->cnt = cnt;
  return (e->plane[ARCLE_PL_INPUT] && e->plane[ARCLE_PL_GRID] && e->plane[ARCLE_PL_ANSWER] && rec && cnt)
             ? 0
             : ARCLE_ERR_ARG;
}

int oracle_set_op_table(oracle_env* e, const uint32_t* descs, int n_ops) {
  if (n_ops <= 0 || n_ops > ARCLE_MAX_OPS) return ARCLE_ERR_CONFIG;
  for (int i = 0; i < n_ops; i++) {
    uint32_t k = ARCLE_OP_KIND(descs[i]), f = ARCLE_OP_FLAGS(descs[i]);
    if (k >= ARCLE_N_OP_KINDS) return ARCLE_ERR_CONFIG;
    int need_sel = (f & (ARCLE_OPF_RESET_SEL | ARCLE_OPF_KEEP_SEL)) != 0;
    int need_obj = (k == ARCLE_OP_MOVE || k == ARCLE_OP_ROTATE || k == ARCLE_OP_FLIP);
    int need_clip = (k == ARCLE_OP_COPY || k == ARCLE_OP_PASTE);
    if ((need_sel || need_obj) && !e->plane[ARCLE_PL_SELECTED]) return ARCLE_ERR_CONFIG;
    if (need_obj && !(e->plane[ARCLE_PL_OBJECT] && e->plane[ARCLE_PL_OBJECT_SEL] && e->plane[ARCLE_PL_BACKGROUND]))
      return ARCLE_ERR_CONFIG;
    if (need_clip && !e->plane[ARCLE_PL_CLIP]) return ARCLE_ERR_CONFIG;
    e->ops[i] = descs[i];
  }
  e->n_ops = n_ops;
  return 0;
}

int oracle_reset(oracle_env* e, const uint8_t* mask) {
  for (int n = 0; n < e->n_envs; n++) {
    if (mask && !mask[n]) continue;
    view v = make_view(e, n);
    init_state(e, &v);
  }
  return 0;
}

uint32_t oracle_get_status(oracle_env* e, int clear) {
  uint32_t s = e->status;
  if (clear) e->status = 0;
  return s;
}

/* O2ARCv2Env.step, o2arcenv.py:130-147 (ARCEnv.step arcenv.py:155-172, RawARCEnv.step :60-76) for
 * env n with a full selection mask. */
static void step_one(oracle_env* e, int n, const int8_t* sel, int op, int32_t* reward, uint8_t* term,
                     uint32_t flags) {
  view v = make_view(e, n);
  if ((flags & ARCLE_STEP_AUTORESET) && v.rec[ARCLE_REC_TERMINATED]) {
    init_state(e, &v);
    *reward = 0;
    *term = 0;
    return;
  }
  if (op < 0 || op >= e->n_ops || ARCLE_OP_KIND(e->ops[op]) == ARCLE_OP_NONE) {
    __atomic_fetch_or(&e->status, ARCLE_ST_BAD_OP, __ATOMIC_RELAXED); /* reference: IndexError / TypeError before any mutation */
    *reward = 0;
    *term = (uint8_t)(v.rec[ARCLE_REC_TERMINATED] != 0);
    return;
  }
  uint32_t d = e->ops[op];
  int kind = (int)ARCLE_OP_KIND(d), arg = (int)ARCLE_OP_ARG(d);
  uint32_t f = ARCLE_OP_FLAGS(d);
  snap before;
  int may_raise = (kind == ARCLE_OP_ROTATE || kind == ARCLE_OP_FLIP);
  if (may_raise) snap_take(&v, &before);
  if (f & ARCLE_OPF_RESET_SEL) reset_sel(&v);
  if (f & ARCLE_OPF_KEEP_SEL) keep_sel(&v, sel);
  int rc = 0;
  switch (kind) { /* transition(): self.operations[op](state, action)  o2arcenv.py:149-151 */
    case ARCLE_OP_COLOR: op_color(&v, sel, arg); break;
    case ARCLE_OP_FLOODFILL: op_floodfill(&v, sel, arg); break;
    case ARCLE_OP_MOVE: op_move(&v, sel, arg); break;
    case ARCLE_OP_ROTATE: rc = op_rotate(&v, sel, arg, &e->status); break;
    case ARCLE_OP_FLIP: rc = op_flip(&v, sel, arg, &e->status); break;
    case ARCLE_OP_COPY: op_copy(&v, sel, arg); break;
    case ARCLE_OP_PASTE: op_paste(&v, sel, arg); break;
    case ARCLE_OP_COPY_FROM_INPUT: op_copy_from_input(&v); break;
    case ARCLE_OP_RESET_GRID: op_reset_grid(&v); break;
    case ARCLE_OP_RESIZE_GRID: op_resize_grid(&v, sel); break;
    case ARCLE_OP_CROP_GRID: op_crop_grid(&v, sel); break;
    case ARCLE_OP_RESIZE_TO_ANSWER: op_resize_to_answer(&v); break;
    case ARCLE_OP_SUBMIT: op_submit(&v); break;
    default: break;
  }
  if (rc < 0) { /* the reference raised inside the op: the step did not happen (state restored) */
    snap_restore(&v, &before);
    *reward = 0;
    *term = (uint8_t)(v.rec[ARCLE_REC_TERMINATED] != 0);
    return;
  }
  /* reward(): o2arcenv.py:121-128 — only the LAST op of the table can be rewarded */
  *reward = (op == e->n_ops - 1 && grid_equals_answer(&v)) ? 1 : 0;
  v.cnt[ARCLE_CNT_STEPS] += 1; /* :142 */
  *term = (uint8_t)(v.rec[ARCLE_REC_TERMINATED] != 0);
}

/* Envs are independent, so the env loops below may be split over host threads (OpenMP, used only for the
 * "all cores" CPU baseline; oracle_set_threads(1) — the default — keeps the plain serial loop). */
static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }

int oracle_step_mask(oracle_env* e, const int8_t* sel, const int32_t* op, int32_t* reward, uint8_t* term,
                     uint32_t flags) {
  size_t P = (size_t)e->H * e->W;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int n = 0; n < e->n_envs; n++) step_one(e, n, sel + n * P, op[n], &reward[n], &term[n], flags);
  return 0;
}

/* BBoxWrapper.action, bbox.py:22-30 (non-negative coordinates; slices clip at H, W) */
int oracle_step_bbox(oracle_env* e, const int32_t* bbox, const int32_t* op, int32_t* reward, uint8_t* term,
                     uint32_t flags) {
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int n = 0; n < e->n_envs; n++) {
    int8_t sel[ORACLE_MAX_CELLS];
    int x1 = bbox[4 * n], y1 = bbox[4 * n + 1], x2 = bbox[4 * n + 2], y2 = bbox[4 * n + 3];
    if (x1 > x2) { int t = x1; x1 = x2; x2 = t; }
    if (y1 > y2) { int t = y1; y1 = y2; y2 = t; }
    if (x1 < 0 || y1 < 0) __atomic_fetch_or(&e->status, ARCLE_ST_BAD_SELECTION, __ATOMIC_RELAXED);
    memset(sel, 0, (size_t)e->H * e->W);
    for (int i = x1 < 0 ? e->H : x1; i <= x2 && i < e->H; i++)
      for (int j = y1 < 0 ? e->W : y1; j <= y2 && j < e->W; j++) sel[i * e->W + j] = 1;
    step_one(e, n, sel, op[n], &reward[n], &term[n], flags);
  }
  return 0;
}

/* PointWrapper.action, bbox.py:43-49 (out-of-range points select nothing) */
int oracle_step_point(oracle_env* e, const int32_t* xy, const int32_t* op, int32_t* reward, uint8_t* term,
                      uint32_t flags) {
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int n = 0; n < e->n_envs; n++) {
    int8_t sel[ORACLE_MAX_CELLS];
    int x = xy[2 * n], y = xy[2 * n + 1];
    memset(sel, 0, (size_t)e->H * e->W);
    if (x >= 0 && x < e->H && y >= 0 && y < e->W) sel[x * e->W + y] = 1;
    else __atomic_fetch_or(&e->status, ARCLE_ST_BAD_SELECTION, __ATOMIC_RELAXED);
    step_one(e, n, sel, op[n], &reward[n], &term[n], flags);
  }
  return 0;
}
