/*
 * arcle_hip.h — C ABI of the MI355X-native ARCLE hot path (libarcle_hip.so).
 *
 * This is the drop-in boundary for the reference's data-parallel hot path:
 *   O2ARCv2Env.step()/transition()      /root/reference/arcle/envs/o2arcenv.py:130-151
 *   ARCEnv.step()/transition()          /root/reference/arcle/envs/arcenv.py:155-176
 *   RawARCEnv.step()                    /root/reference/arcle/envs/arcenv.py:60-76
 *   AbstractARCEnv.submit()/init_state  /root/reference/arcle/envs/base.py:155-183
 *   the op closures of arcle/actions    color.py:62-103, object.py:10-349, critical.py:8-66
 *   BBoxWrapper/PointWrapper.action     /root/reference/arcle/wrappers/bbox.py:22-30,43-49
 *
 * The reference has no FFI of its own (it is pure Python); the functions below are what
 * a ctypes binding inside the reference's `step()` would call.
 *
 * Conventions
 *   - every pointer in `arcle_buffers` and every per-step array is a DEVICE pointer
 *     (HBM of the GPU the handle was created on); plain C types only, no torch types.
 *   - all functions return 0 on success or a negative arcle_status; they never throw.
 *   - kernels are enqueued on the `hipStream_t` passed as `void* stream` (NULL = default
 *     stream) and are asynchronous; outputs are valid after the stream is synchronised.
 *   - a handle is not thread-safe; serialise calls on one handle (the reference env is
 *     single-threaded as well).
 *
 * State layout in HBM (structure of arrays, all int8, one row per env):
 *   plane[p]  : int8 [n_envs][PS]    env e's H x W cells are the first H*W bytes (row-major: row, col) of row e; the
 *                                    row stride PS = arcle_config.plane_stride (default: H*W rounded up to 128 B,
 *                                    e.g. 1024 for 30x30); bytes [H*W, PS) of every row are zero padding.
 *                                    p in arcle_plane
 *   rec       : int8 [n_envs][16]    packed per-env scalars, byte offsets ARCLE_REC_*
 *   cnt       : int32[n_envs][2]     {action_steps, submit_count}
 */
#ifndef ARCLE_HIP_H
#define ARCLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARCLE_ABI_VERSION 5
#define ARCLE_MAX_OPS 64
#define ARCLE_MAX_CELLS 1024 /* H*W <= 1024: one 64-lane wavefront x 16 cells holds a plane (the one-wavefront-per-env kernels);
                                larger planes (H, W <= 127) are served by the workgroup-per-env kernels — see "Grids beyond
                                ARCLE_MAX_CELLS" below */
#define ARCLE_MAX_SIDE 127   /* H, W <= 127: grid dims are int8 in the record (and in the reference's state dict, base.py:162-166) */
/* default per-env plane stride: H*W rounded up to a whole number of 128-byte lines (30x30 -> 1024 B), so that no two
 * envs share a cache line of a plane and every plane store writes full lines */
#define ARCLE_DEFAULT_PLANE_STRIDE(P) (((P) + 127) & ~127)

/* ---- planes: keys of the reference state dict (o2arcenv.py:16-34, base.py:155-166) ---- */
enum arcle_plane {
  ARCLE_PL_INPUT = 0,      /* state['input']                                        */
  ARCLE_PL_GRID = 1,       /* state['grid']                                         */
  ARCLE_PL_SELECTED = 2,   /* state['selected']                    (O2ARCv2Env)     */
  ARCLE_PL_CLIP = 3,       /* state['clip']                        (O2ARCv2Env/ARCEnv) */
  ARCLE_PL_OBJECT = 4,     /* state['object_states']['object']     (O2ARCv2Env)     */
  ARCLE_PL_OBJECT_SEL = 5, /* state['object_states']['object_sel'] (O2ARCv2Env)     */
  ARCLE_PL_BACKGROUND = 6, /* state['object_states']['background'] (O2ARCv2Env)     */
  ARCLE_PL_ANSWER = 7,     /* env.answer zero-padded to HxW (info['answer'], base.py:150) */
  ARCLE_N_PLANES = 8
};

/* ---- per-env scalar record, byte offsets into rec[env][16] (all int8) ---- */
#define ARCLE_REC_INPUT_DIM 0   /* [2] state['input_dim']                     */
#define ARCLE_REC_GRID_DIM 2    /* [2] state['grid_dim']                      */
#define ARCLE_REC_CLIP_DIM 4    /* [2] state['clip_dim']                      */
#define ARCLE_REC_OBJECT_DIM 6  /* [2] object_states['object_dim']            */
#define ARCLE_REC_OBJECT_POS 8  /* [2] object_states['object_pos'] (signed)   */
#define ARCLE_REC_TRIALS 10     /* [1] state['trials_remain']                 */
#define ARCLE_REC_TERMINATED 11 /* [1] state['terminated']                    */
#define ARCLE_REC_ACTIVE 12     /* [1] object_states['active']                */
#define ARCLE_REC_PARITY 13     /* [1] object_states['rotation_parity']       */
#define ARCLE_REC_ANSWER_DIM 14 /* [2] env.answer.shape                       */
#define ARCLE_REC_BYTES 16

#define ARCLE_CNT_STEPS 0  /* env.action_steps == info['steps']          */
#define ARCLE_CNT_SUBMIT 1 /* env.submit_count == info['submit_count']   */

/* ---- op descriptors: one uint32 per slot of the env's operation table ----
 * desc = kind | (arg << 8) | (flags << 16).  The table is what
 * AbstractARCEnv.create_operations() returns (base.py:140-142); slot index == the
 * integer `action['operation']`. */
enum arcle_op_kind {
  ARCLE_OP_NONE = 0,             /* empty slot (never valid to execute)                     */
  ARCLE_OP_COLOR = 1,            /* gen_color(arg)        color.py:62-77                    */
  ARCLE_OP_FLOODFILL = 2,        /* gen_flood_fill(arg)   color.py:79-103 (+dfs :8-30)      */
  ARCLE_OP_MOVE = 3,             /* gen_move(arg) 0=U 1=D 2=R 3=L    object.py:218-243      */
  ARCLE_OP_ROTATE = 4,           /* gen_rotate(arg) k=1,2,3 (CCW)    object.py:167-216      */
  ARCLE_OP_FLIP = 5,             /* gen_flip(axis) 0=H 1=V 2=D0 3=D1 object.py:245-279      */
  ARCLE_OP_COPY = 6,             /* gen_copy(src) 0="I" 1="O"        object.py:281-314      */
  ARCLE_OP_PASTE = 7,            /* gen_paste(paste_blank=arg)       object.py:316-349      */
  ARCLE_OP_COPY_FROM_INPUT = 8,  /* copy_from_input       critical.py:19-29                 */
  ARCLE_OP_RESET_GRID = 9,       /* reset_grid            critical.py:8-17                  */
  ARCLE_OP_RESIZE_GRID = 10,     /* resize_grid           critical.py:31-46                 */
  ARCLE_OP_CROP_GRID = 11,       /* crop_grid             critical.py:48-66                 */
  ARCLE_OP_RESIZE_TO_ANSWER = 12,/* RawARCEnv resize_to_answer  arcenv.py:31-35             */
  ARCLE_OP_SUBMIT = 13,          /* AbstractARCEnv.submit base.py:172-183                   */
  ARCLE_OP_HOST = 14,            /* slot of an arbitrary host callable (base.py:140-142): a device no-op — the step is
                                    counted, the host layer applies the callable to the fetched state               */
  ARCLE_N_OP_KINDS = 15
};
#define ARCLE_OPF_RESET_SEL 1u /* wrapped by reset_sel  object.py:10-26 */
#define ARCLE_OPF_KEEP_SEL 2u  /* wrapped by keep_sel   object.py:28-41 */
#define ARCLE_OP_DESC(kind, arg, flags) \
  ((uint32_t)(kind) | ((uint32_t)(arg) << 8) | ((uint32_t)(flags) << 16))
#define ARCLE_OP_KIND(d) ((d) & 0xffu)
#define ARCLE_OP_ARG(d) (((d) >> 8) & 0xffu)
#define ARCLE_OP_FLAGS(d) (((d) >> 16) & 0xffu)

/* ---- step flags ---- */
/* Envs whose `terminated` is already 1 when the step starts are re-initialised from their
 * task (init_state, base.py:155-166 + o2arcenv.py:16-34) instead of executing the action;
 * reward 0, terminated 0 for that step (Gymnasium "next-step" autoreset; not in the
 * reference, which keeps mutating a terminated env — that is the default here too). */
#define ARCLE_STEP_AUTORESET 1u
/* reset_sel (object.py:20-25) need not rewrite a `selected` plane that is already zero.  Whenever the op table holds
 * no keep_sel-wrapped op, `active == 0` implies `selected == 0` (object ops set both, reset_sel / init_state clear both,
 * nothing else writes `selected`), so with this flag the zero-fill is skipped when the env enters the step with
 * active == 0.  The state after the step is bit-identical; only a redundant 1-plane write disappears.  Do NOT set it
 * for states written from outside (e.g. a state dict uploaded for transition()) that may violate the invariant;
 * arcle_set_op_table() reports through arcle_can_elide_selected() whether the installed table permits it. */
#define ARCLE_STEP_ELIDE_SELECTED 2u
/* also writes truncated[env] = (action_steps >= step_limit) into the array installed with arcle_set_truncation
 * (gymnasium TimeLimit(max_episode_steps) as the reference's training script applies it, agents/train.py:67) */
#define ARCLE_STEP_TRUNCATE 4u
/* like ARCLE_STEP_AUTORESET, but the env starts its next episode on a NEW task drawn on the device from the task table
 * (arcle_set_task_table + arcle_set_sampler): problem and pair uniform, optional colour-permutation / rot90
 * augmentation (agents/env.py:31-42), all keyed by (seed, global env id, episode number) */
#define ARCLE_STEP_RESAMPLE 8u
/* also writes dense[env] = (cells of grid that match the answer inside the common rectangle, total cells as in
 * agents/env.py:44-58) into the int32 [n_envs][2] array installed with arcle_set_dense_output: the research env's dense
 * reward is  sparse*100 - 1 + dense[0]/dense[1]  (formed by the host, exact integers on the device) */
#define ARCLE_STEP_DENSE 16u
/* mask ingress only: an object op (Move/Rotate/Flip) whose selection equals the env's current `selected` plane is
 * executed with an empty selection, i.e. continues the active object — the rule of the reference's O2ARC trace harness
 * (tests/o2arc_check.py:169-170) */
#define ARCLE_STEP_CONTINUE_RULE 32u
/* reset(options={'reset_on_submit': True}) (base.py:87-93,179-180): a Submit with trials left
 * re-initialises the env from its input inside the op; the caller sees the fresh state, terminated stays 0 */
#define ARCLE_STEP_RESET_ON_SUBMIT 64u
/* the step flags served by the feature instantiations of the step kernel (the plain ones stay lean) */
/* the step kernel also writes the flattened observation row (arcle_set_flat_output: FlattenObservation, optionally after
 * FilterO2ARC) of the state it produced — fused into the same launch */
#define ARCLE_STEP_FLAT_OBS 128u
/* the step kernel also writes the env's packed per-step row  grid | grid_dim | reward | terminated  (arcle_set_packed_output; the
 * record a central learner gathers from every GPU) — fused into the same launch */
#define ARCLE_STEP_PACK_OBS 256u
/* with ARCLE_STEP_FLAT_OBS: the row buffer still holds every env's row of the PREVIOUS step (same buffer, installed once, every
 * step of this handle run with ARCLE_STEP_FLAT_OBS since the rows were last written in full — by arcle_flatten_obs into that buffer or
 * by a step without this flag).  The writer then rewrites only the scalars and the segments of the planes this step stored; an
 * op touches one to five of the seven planes, so most of a row is left as it is.  The rows are byte-identical to full rewrites. */
#define ARCLE_STEP_ROWS_INCREMENTAL 512u
#define ARCLE_STEP_FEATURE_FLAGS                                                                                        \
  (ARCLE_STEP_RESAMPLE | ARCLE_STEP_DENSE | ARCLE_STEP_CONTINUE_RULE | ARCLE_STEP_RESET_ON_SUBMIT | ARCLE_STEP_FLAT_OBS | \
   ARCLE_STEP_PACK_OBS | ARCLE_STEP_ROWS_INCREMENTAL)

/* ---- augmentation of a task at reset (arcle_set_sampler / ARCLE_STEP_RESAMPLE / arcle_reset_sampled) ---- */
#define ARCLE_AUG_PERMUTE 1u /* random permutation of the colours 0..9 (applied to input and answer) */
#define ARCLE_AUG_ROT90 2u   /* np.rot90(., k) with k uniform in 0..3                                 */

/* ---- sticky device status bits (arcle_get_status) ---- */
#define ARCLE_ST_BAD_OP 1u       /* operation index out of range / empty slot: step skipped
                                    (reference: IndexError / TypeError)                   */
#define ARCLE_ST_BAD_TASK 4u      /* arcle_reset_from_table: task index outside the table: env left untouched */
#define ARCLE_ST_BAD_SELECTION 8u /* a point outside the H x W plane or a negative bbox / point coordinate: the reference
                                    raises IndexError (or NumPy wraps the negative index); here the selection is
                                    empty, the step runs, and this bit reports it                          */
#define ARCLE_ST_ROTATE_DOMAIN 2u /* Rotate produced a position outside int8 or a tile that
                                    does not fit HxW (reference: ValueError/garbage,
                                    SURVEY.md A.6-2, A.6-6): step skipped                  */

#define ARCLE_ST_AUG_DOMAIN 16u   /* arcle_reset_from_table_aug: an explicit rot90 by an odd count does not fit a non-square
                                    H x W plane: env left untouched (device-drawn augmentations drop the quarter turn instead) */

enum arcle_status {
  ARCLE_OK = 0,
  ARCLE_ERR_ARG = -1,     /* NULL / out-of-range argument                                */
  ARCLE_ERR_CONFIG = -2,  /* unsupported H, W, n_ops or op table needing absent planes   */
  ARCLE_ERR_HIP = -3,     /* a HIP runtime call failed; see arcle_last_error()           */
  ARCLE_ERR_NO_DEVICE = -4
};

typedef struct arcle_config {
  int32_t n_envs;    /* envs owned by this handle (this GPU's shard)                     */
  int32_t H, W;      /* max_grid_size (base.py:49); H, W <= ARCLE_MAX_SIDE (any such size: see "Grids beyond ARCLE_MAX_CELLS") */
  int32_t max_trial; /* base.py:51; stored as int8 in trials_remain                      */
  int32_t device;    /* HIP device ordinal, -1 = current device                          */
  int32_t plane_stride; /* bytes between consecutive envs of a plane (PS): 0 = default (ARCLE_DEFAULT_PLANE_STRIDE(H*W));
                           otherwise a multiple of 16 with H*W <= PS <= 1024 (<= 16256 for grids beyond ARCLE_MAX_CELLS) */
} arcle_config;

typedef struct arcle_buffers {
  int8_t* plane[ARCLE_N_PLANES]; /* NULL for planes the env kind does not have           */
  int8_t* rec;                   /* [n_envs][16]                                         */
  int32_t* cnt;                  /* [n_envs][2]                                          */
} arcle_buffers;

typedef struct arcle_env arcle_env; /* opaque handle */

/* ---- Grids beyond ARCLE_MAX_CELLS -----------------------------------------------------------------------------------------------
 * The reference takes any max_grid_size (base.py:37-49).  A handle with H * W <= ARCLE_MAX_CELLS runs the one-wavefront-per-env kernels
 * (ARC's own regime, 30 x 30: everything in this header applies).  A handle with H * W > ARCLE_MAX_CELLS (H, W <= ARCLE_MAX_SIDE) runs one
 * WORKGROUP per env (arcle_amd/csrc/arcle_big.hip): same layout (PS = H*W rounded up to 128), same entry points, same results bit for
 * bit against the reference's algorithm — with these differences:
 *   served      arcle_reset / _reset_from_table(_aug) / _reset_sampled (task augmentation included), arcle_step_mask / _bbox / _point / _bbox5 /
 *               _bits (+ arcle_pack_mask_bits; rows of plane_stride / 8 bytes), arcle_step_many, arcle_rollout_*
 *               (= n_steps step launches: the state does not fit a wavefront's registers), arcle_transition_rows (three launches over
 *               library-owned scratch envs, allocated on first use: not inside a stream capture), arcle_flatten_obs / _get_state_rows /
 *               _set_state_rows, arcle_pack_obs, planes, status; step flags AUTORESET, ELIDE_SELECTED, TRUNCATE, RESAMPLE, DENSE (the pair is
 *               computed from the planes every step: no cache), CONTINUE_RULE, RESET_ON_SUBMIT, FLAT_OBS (tail and completion signal
 *               included), PACK_OBS; ROWS_INCREMENTAL is accepted and rewrites the rows in full (identical bytes)
 *   accounting  arcle_enable_accounting counts every 16-byte access the threads issue (planes, table entries, mask chunks, rows in) + the
 *               env's scalars as `issued`; the `bytes` figure is the same without the row padding (x H*W / plane_stride)
 *   no-ops      arcle_set_dispatch_order, arcle_hint_next_ops, arcle_autotune (returns 0 candidates: one launch plan), arcle_launch_info
 *               reports {0, 0, waves per workgroup, 0}
 * Action arrays and row buffers may be device or pinned host memory as everywhere else. */

/* Creates a handle. `bufs` are caller-owned device buffers (e.g. torch tensors' data_ptr);
 * if bufs == NULL the library allocates all planes itself (hipMalloc) and frees them in
 * arcle_destroy. Replaces AbstractARCEnv.__init__ state allocation (base.py:37-66). */
int arcle_create(const arcle_config* cfg, const arcle_buffers* bufs, arcle_env** out);
int arcle_destroy(arcle_env* env);
/* Fills `out` with the device pointers the handle uses. */
int arcle_get_buffers(const arcle_env* env, arcle_buffers* out);

/* Installs the operation table (host array of descriptors). Replaces the list returned by
 * create_operations() (o2arcenv.py:76-113, arcenv.py:26-41,110-138). */
int arcle_set_op_table(arcle_env* env, const uint32_t* descs, int32_t n_ops);

/* 1 if the installed op table keeps the invariant documented at ARCLE_STEP_ELIDE_SELECTED, else 0. */
int arcle_can_elide_selected(const arcle_env* env);

/* Re-initialises envs from PL_INPUT / REC_INPUT_DIM (init_state: base.py:155-166,
 * o2arcenv.py:16-34; counters as in reset base.py:73-79). `mask` is a device uint8[n_envs]
 * (non-zero = reset) or NULL for all envs. The task (input, answer, dims) must have been
 * written into PL_INPUT, PL_ANSWER, REC_INPUT_DIM, REC_ANSWER_DIM by the host layer. */
int arcle_reset(arcle_env* env, const uint8_t* mask, void* stream);

/* Device task table: n_tasks (input, answer) pairs, already zero-padded to the plane stride PS = H*W rounded up
 * (arcle_config.plane_stride): in_planes / ans_planes int8 [n_tasks][PS], in_dims / ans_dims int8 [n_tasks][2] (device pointers, owned
 * by the caller, must stay alive).  It is the packed form of what Loader.parse yields (loader.py:89-113), one entry
 * per (task, pair).  Alignment (ABI 5): the planes 16 bytes; the dims arrays 4 bytes, their allocation covering 2 * n_tasks rounded up
 * to a multiple of 4 bytes (an entry's two dims are read with one aligned dword load) — misaligned arrays are rejected. */
int arcle_set_task_table(arcle_env* env, const int8_t* in_planes, const int8_t* in_dims, const int8_t* ans_planes,
                         const int8_t* ans_dims, int32_t n_tasks);
/* reset() with the task choice made by the caller (base.py:95-108): for every env with mask[env] != 0 (mask NULL =
 * all) copies table entry task_idx[env] (device int32[n_envs]) into PL_INPUT / PL_ANSWER / REC_*_DIM and runs
 * init_state.  No host loop, no host->device copy of grids. */
int arcle_reset_from_table(arcle_env* env, const int32_t* task_idx, const uint8_t* mask, void* stream);

/* One step() of every env. Replaces O2ARCv2Env.step (o2arcenv.py:130-147).
 *   op      device int32[n_envs]       action['operation']
 *   reward  device int32[n_envs] out   0/1 (o2arcenv.py:121-128)
 *   term    device uint8[n_envs] out   bool(state['terminated'][0])
 * selection ingress, three forms:
 *   _mask : sel  device int8 [n_envs][H*W]   action['selection'] as given
 *   _bbox : bbox device int32[n_envs][4]     (x1,y1,x2,y2) as BBoxWrapper.action (bbox.py:22-30)
 *   _point: xy   device int32[n_envs][2]     (x,y)         as PointWrapper.action (bbox.py:43-49)
 */
int arcle_step_mask(arcle_env* env, const int8_t* sel, const int32_t* op, int32_t* reward,
                    uint8_t* term, uint32_t flags, void* stream);
int arcle_step_bbox(arcle_env* env, const int32_t* bbox, const int32_t* op, int32_t* reward,
                    uint8_t* term, uint32_t flags, void* stream);
int arcle_step_point(arcle_env* env, const int32_t* xy, const int32_t* op, int32_t* reward,
                     uint8_t* term, uint32_t flags, void* stream);

/* Two more forms of the action, same step():
 *   _bbox5: act5 device int32[n_envs][5]   the BBoxWrapper action as ONE record per env (x1, y1, x2, y2, operation) — exactly the
 *           5-tuple `BBoxWrapper.action` receives (bbox.py:22-30, examples/example_bbox.py:13-15); no separate op array, so a
 *           host-resident policy moves its actions with one 20-byte-per-env copy (or none: act5 may be pinned host memory)
 *   _bits : bits device uint8[n_envs][128] boolean selection masks, bit-packed: bit (f & 7) of byte (f >> 3) of row e = cell f
 *           (row-major, f = row * W + col) of env e is selected; rows are ARCLE_MAX_CELLS / 8 = 128 bytes apart (handles of more than
 *           ARCLE_MAX_CELLS cells: plane_stride / 8 bytes — arcle_mask_bits_stride() says which), 2-byte aligned.
 *           `create_action_space` accepts boolean masks (base.py:134-138); packed they are 1/8 of the int8 traffic and need no
 *           byte -> bit reduction in the kernel.  arcle_pack_mask_bits converts int8 [n_envs][H*W] masks (truthy = non-zero). */
enum arcle_ingress { ARCLE_INGRESS_MASK = 0, ARCLE_INGRESS_BBOX = 1, ARCLE_INGRESS_POINT = 2, ARCLE_INGRESS_BBOX5 = 3, ARCLE_INGRESS_BITS = 4 };
int arcle_step_bbox5(arcle_env* env, const int32_t* act5, int32_t* reward, uint8_t* term, uint32_t flags, void* stream);
int arcle_step_bits(arcle_env* env, const uint8_t* bits, const int32_t* op, int32_t* reward, uint8_t* term, uint32_t flags,
                    void* stream);
int arcle_pack_mask_bits(arcle_env* env, const int8_t* sel, uint8_t* bits, void* stream);
int arcle_mask_bits_stride(const arcle_env* env); /* bytes between the envs' rows of a bit-packed mask array */

/* n_steps consecutive step() LAUNCHES enqueued by ONE call (the loop `for t in range(n): env.step(actions[t])` of a caller that
 * holds the actions of the next n steps; every step is a full step(): state observable in between on the stream, all step flags
 * valid).  `ingress`: arcle_ingress; sel: the form's payload [n_steps][n_envs][...]; op int32 [n_steps][n_envs] (NULL for BBOX5);
 * reward int32 [n_steps][n_envs], term uint8 [n_steps][n_envs] out.  Per-handle outputs (truncated, dense, flat / packed rows) hold
 * the LAST step's values afterwards.  Capturable into a hipGraph like the single-step calls (no synchronisation).
 * Host-resident actions: with ARCLE_INGRESS_BBOX5 `sel` may be PINNED HOST memory (the records of a policy that runs on the CPU).
 * For the standard 30 x 30 batch stepped with ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED the library then pipelines the PCIe
 * traffic: eight extra workgroups at the front of launch t copy step t+1's records into a device staging buffer while launch t
 * runs, and step t+1 reads them from HBM (only step 0 reads across PCIe itself).  The staging buffer (2 x 20 B per env) is allocated
 * by the first such call made outside a stream capture. */
int arcle_step_many(arcle_env* env, int ingress, int32_t n_steps, const void* sel, const int32_t* op, int32_t* reward,
                    uint8_t* term, uint32_t flags, void* stream);
/* Dispatch order (ABI 5).  A launch of one wave per env ends with its last Move / Rotate / Flip wave (they run ~1.2 us longer than the
 * other operations' waves, and the hardware starts the waves of a launch over ~2 us), so the library hands the object operations to
 * the waves that start first.  Since ABI 5 every launch does that for ITSELF, from the operations it is about to execute: the standard
 * 30 x 30 batch (plane stride 1024) stepped through arcle_step_bbox / _bbox5 / _point / _mask / _bits — and arcle_step_many over them — with
 * the flag set ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED; the tuple forms also with ARCLE_STEP_ELIDE_SELECTED alone (no auto-reset);
 * bbox / bbox5 also with ARCLE_STEP_AUTORESET | _ELIDE_SELECTED | _PACK_OBS or the research env's set with incremental FilterO2ARC rows; n_envs a multiple of 256 in [2304, 10240], actions in DEVICE memory, an op table with object operations, no byte
 * accounting.  Waves of a group of 32 consecutive envs read the group's 32 actions and permute the group among their 32 dispatch slots
 * (arcle_step_kernel, GROUPED).  Scheduling only: every env is stepped exactly once whatever the operations are; results, outputs and
 * their layout are those of the plain launch.  No tables, no extra memory, nothing to allocate before a stream capture.
 * arcle_set_dispatch_order(env, 0) turns it off for the handle (default on).
 * arcle_hint_next_ops: ABI 4's one-shot hint of the NEXT step's operations (the table form of ordered dispatch needed them a step
 * ahead).  Still accepted and validated — stride 1 for op arrays, 5 for the op field of BBoxWrapper records — and ignored. */
int arcle_set_dispatch_order(arcle_env* env, int enable);
int arcle_hint_next_ops(arcle_env* env, const int32_t* next_op, int32_t stride);
/* How a step launch of this handle runs, chosen by batch size from tables measured on one MI355X with a cache-resident action stream:
 * the cache policy of the speculative grid request (0 none; 'A' spec + write-through stores, small batches; 'B' + non-temporal stores,
 * 'H' non-temporal request, 'J' both: state beyond the 256 MiB Infinity Cache), the workgroup size (4 or 8 waves), and whether the
 * launch orders itself.  arcle_launch_info reports the plan a launch with (ingress, flags) and device-resident actions would take:
 * out4 = {orders itself, policy (0 or the letter), waves per workgroup, 1 if arcle_autotune chose it}.
 * arcle_autotune replaces the tables for THIS handle: it times every plan the launch can take — on this handle's batch size, this box and
 * the caller's own action arrays where they live (n_batches consecutive action batches laid out as for arcle_step_many: sel
 * [n_batches][n_envs][...], op [n_batches][n_envs]; a representative stretch of the policy's output — ONE repeated batch is not: the
 * state degenerates under it; ingress BBOX, BBOX5, POINT, MASK or BITS; flags within ARCLE_STEP_AUTORESET | _ELIDE_SELECTED | _PACK_OBS)
 * — 48 .. 96 warm-up launches (the caches' steady state: fewer mis-rank the non-temporal policies) and 24 .. 64 timed ones each, walking
 * the batches in order — and keeps the fastest for later launches with the same ingress and flags.  The env state is saved first and restored before every candidate and at the end (a temporary copy of the state
 * in device memory; reward / terminated of the timed launches go to scratch): the handle is left exactly as it was found.  Synchronises
 * the stream; not inside a stream capture.  report (may be NULL): int32 [report_rows][4] = {orders itself, policy, waves per workgroup,
 * ns per launch} per candidate timed.  Returns the number of candidates timed (>= 0) or a negative ARCLE_ERR_* code. */
int arcle_launch_info(arcle_env* env, int ingress, uint32_t flags, int32_t* out4);
int arcle_autotune(arcle_env* env, int ingress, int32_t n_batches, const void* sel, const int32_t* op, uint32_t flags, int32_t* report,
                   int32_t report_rows, void* stream);

/* n_steps consecutive step()s of every env in ONE launch — a rollout / trace replay for callers that already hold
 * the whole action sequence (the loop `for a in trace: env.step(a)`, e.g. tests/o2arc_check.py:139-199 of the
 * reference).  Semantically identical to n_steps calls of arcle_step_bbox/_point; the env state is kept on chip
 * between the steps, so only the final state is observable afterwards.
 *   bbox / xy  device int32[n_steps][n_envs][4 | 2]      op      device int32[n_steps][n_envs]
 *   reward     device int32[n_steps][n_envs] out         term    device uint8[n_steps][n_envs] out
 * flags: ARCLE_STEP_AUTORESET | _ELIDE_SELECTED; mask ingress also _CONTINUE_RULE | _RESET_ON_SUBMIT; and (ABI 5) ARCLE_STEP_PACK_OBS:
 * the launch then also writes the packed observation row of EVERY step — the buffer installed with arcle_set_packed_output must hold
 * uint8 [n_steps][n_envs][arcle_packed_obs_size()] — so an action-chunk caller keeps Gym's "observation after every step". */
int arcle_rollout_bbox(arcle_env* env, int32_t n_steps, const int32_t* bbox, const int32_t* op, int32_t* reward,
                       uint8_t* term, uint32_t flags, void* stream);
int arcle_rollout_point(arcle_env* env, int32_t n_steps, const int32_t* xy, const int32_t* op, int32_t* reward,
                        uint8_t* term, uint32_t flags, void* stream);
/* the same with full selection masks: sel device int8 [n_steps][n_envs][H*W] (the O2ARC trace replayer's form,
 * tests/o2arc_check.py:139-199 of the reference) */
int arcle_rollout_mask(arcle_env* env, int32_t n_steps, const int8_t* sel, const int32_t* op, int32_t* reward,
                       uint8_t* term, uint32_t flags, void* stream);

/* Device-side task choice.  pair_off / pair_cnt: device int32 [n_problems] — first task-table entry and number of
 * entries (pairs) of every problem that has at least one (Loader.pick's candidates for the current adaptation mode);
 * episode: device int32 [n_envs] episodes started so far (the RNG stream position; the library increments it);
 * cur_task: device int32 [n_envs] or NULL, receives the table index each env currently runs; env_base: global id of
 * this handle's env 0 (multi-GPU shards), aug_flags: ARCLE_AUG_*.  The draw is a pure function of
 * (seed, env_base + env, episode[env]) — see arcle::draw_task. */
int arcle_set_sampler(arcle_env* env, const int32_t* pair_off, const int32_t* pair_cnt, int32_t n_problems, uint64_t seed,
                      int64_t env_base, int32_t* episode, int32_t* cur_task, uint32_t aug_flags);
/* reset() of the masked envs (mask NULL = all) onto device-drawn tasks (arcle_set_sampler). */
int arcle_reset_sampled(arcle_env* env, const uint8_t* mask, void* stream);
/* arcle_reset_from_table with an explicit augmentation per env: aug_k device uint8 [n_envs] (np.rot90 count) and/or
 * aug_perm device uint8 [n_envs][16] (perm[c] for colour c < 10); NULL = none. */
int arcle_reset_from_table_aug(arcle_env* env, const int32_t* task_idx, const uint8_t* mask, const uint8_t* aug_k,
                               const uint8_t* aug_perm, void* stream);
/* Installs the output of ARCLE_STEP_DENSE: dense_out device int32 [n_envs][2]; NULL removes it.  (0, 0) is written for a step that
 * executed no action (the auto-reset step of an env, a skipped step): "no dense term".  The library keeps the pair of every env's
 * current grid in a cache of its own and recomputes it only when a step stored the grid plane; a step or rollout launched WITHOUT
 * ARCLE_STEP_DENSE on such a handle drops the whole cache first (stream-ordered), so the pairs never describe a grid that moved. */
int arcle_set_dense_output(arcle_env* env, int32_t* dense_out);
/* For code that edits state planes BEHIND the library's back (plain copies into the buffers of arcle_get_buffers, host-applied
 * op slots): forget everything derived from the state (the dense-pair cache).  The library's own writers (reset kernels,
 * arcle_set_state_rows, auto-reset) do this themselves.  Asynchronous on `stream`. */
int arcle_invalidate(arcle_env* env, void* stream);

/* Installs the output of ARCLE_STEP_TRUNCATE: trunc_out device uint8[n_envs], step_limit = max_episode_steps.
 * trunc_out == NULL removes it. */
int arcle_set_truncation(arcle_env* env, uint8_t* trunc_out, int32_t step_limit);

/* Flattened observation — what the reference's policies consume.  One row per env holding the state dict in Gymnasium
 * FlattenObservation order (keys sorted, nested object_states in place; agents/models/GPTPolicy.py:17-35
 * `unflatten_vec`):  clip, clip_dim, grid, grid_dim, input, input_dim, active, background, object, object_dim,
 * object_pos, object_sel, rotation_parity, selected, terminated, trials_remain = 7*H*W + 14 bytes for O2ARCv2Env (6314 at
 * 30x30; env kinds without some keys omit them), or — filtered != 0 — the FilterO2ARC subset the reference's training
 * script flattens (agents/env.py:109-126, agents/train.py:61-68): active, clip, clip_dim, grid, grid_dim, object,
 * object_dim, object_pos, trials_remain = 3*H*W + 10 bytes (2710).
 * arcle_flat_obs_size() is that logical row length; `out` is a device buffer int8 [n_envs][out_stride], 16-byte aligned,
 * out_stride a multiple of 16 >= the length (the tail of every row is zero).  Rows are written with aligned, fully
 * coalesced 16-byte stores. */
int arcle_flat_obs_size(const arcle_env* env, int filtered);
int arcle_flatten_obs(arcle_env* env, int8_t* out, int32_t out_stride, int filtered, void* stream);
/* Destination of ARCLE_STEP_FLAT_OBS: every step call carrying that flag also writes the observation rows of the state it
 * produced (same stream, one ABI call per step, no host round trip).  out == NULL removes it. */
int arcle_set_flat_output(arcle_env* env, int8_t* out, int32_t out_stride, int filtered);

/* The same with tail != 0: the LAST 16 bytes of every row's stride then carry the step outputs of the env —
 *   int32 reward | int32 action_steps | int32 submit_count | uint8 terminated | uint8 truncated | uint8 status | 0
 * (status: the ARCLE_ST_* bits THIS env raised in THIS step) — so that one copy of the row (or none, when `out` is pinned host
 * memory) returns everything step() returns.  out_stride >= arcle_flat_obs_size() rounded up to 16, plus 16. */
int arcle_set_flat_output_ex(arcle_env* env, int8_t* out, int32_t out_stride, int filtered, int tail);
/* Completion signal in the row tail (ABI 5): with seq in 1..255 the LAST byte of every tail written from now on — by steps carrying
 * ARCLE_STEP_FLAT_OBS and by arcle_transition_rows with a tail — holds `seq`, and that word is stored last, behind a release at
 * system scope: a host that finds `seq` in the tail of a row in PINNED memory also finds the whole row and the other tail words.  It
 * can therefore poll that byte instead of synchronising the stream (a blocking synchronise wakes up tens of microseconds late): what
 * the single-env classes do — the caller changes seq from launch to launch.  0 (default) writes the tail as one plain 16-byte store. */
int arcle_set_flat_seq(arcle_env* env, int32_t seq);

/* ---- state rows at the boundary -------------------------------------------------------------------------------------------
 * A "state row" is the full (unfiltered) flattened observation of one env: the reference's state dict (base.py:155-166,
 * o2arcenv.py:16-34) as arcle_flat_obs_size(env, 0) bytes in FlattenObservation order.
 *   arcle_get_state_rows   = arcle_flatten_obs(.., filtered = 0): resident state -> rows (16-byte aligned, stride multiple of 16)
 *   arcle_set_state_rows   rows -> resident state of env e for every e with mask[e] != 0 (mask NULL = all); any row alignment /
 *                          stride >= the row length.  The inverse of the above: get + set is a checkpoint / restore of the state
 *                          dict (the task side — answer, answer_dim — and the counters are not part of a row and stay).
 *   arcle_transition_rows  the reference's `transition(state, action)` (o2arcenv.py:149-151; README.md:55
 *                          `env.transition(deepcopy(state), action)`) for a BATCH of (state, action) pairs: row r of rows_in + action r
 *                          -> row r of rows_out; the handle's resident envs are not touched, so any number of hypothetical states
 *                          (planning / search) can be expanded per launch.  Submit and the reward compare with the answer of resident
 *                          env src_env[r] (device int32[n_rows]; NULL = env r, then n_rows <= n_envs).  ingress: ARCLE_INGRESS_MASK /
 *                          _BBOX / _POINT with the matching `sel` array [n_rows][...]; op int32[n_rows]; reward int32[n_rows],
 *                          term uint8[n_rows] out; tail as arcle_set_flat_output_ex (action_steps = 1, submit_count = 1 iff the
 *                          Submit counted, base.py:174-175).  flags: ARCLE_STEP_RESET_ON_SUBMIT | _DENSE | _CONTINUE_RULE.
 *                          rows_out may equal rows_in when the strides agree: IN PLACE, only the planes the op changed (and the scalars)
 *                          are rewritten — about half the time of the out-of-place form. */
int arcle_get_state_rows(arcle_env* env, int8_t* rows, int32_t stride, void* stream);
int arcle_set_state_rows(arcle_env* env, const int8_t* rows, int32_t stride, const uint8_t* mask, void* stream);
int arcle_transition_rows(arcle_env* env, int32_t n_rows, const int8_t* rows_in, int32_t in_stride, int ingress, const void* sel,
                          const int32_t* op, const int32_t* src_env, int8_t* rows_out, int32_t out_stride, int tail,
                          int32_t* reward, uint8_t* term, uint32_t flags, void* stream);
/* One state plane as a dense [n_envs][H*W] int8 array (device or pinned host memory), a strided copy on the stream: the
 * get_state()/set_state() of single keys of the reference's state dict. */
int arcle_get_plane(arcle_env* env, int plane, int8_t* dst, void* stream);
int arcle_set_plane(arcle_env* env, int plane, const int8_t* src, void* stream);

/* Packed minimal observation for a central learner (what the multi-GPU gather moves): one row per env,
 *   grid (H*W bytes) | grid_dim (2) | reward int32 little-endian (4) | terminated (1) | zero padding
 * of arcle_packed_obs_size() bytes (H*W + 7 rounded up to 16; 912 for 30x30).  reward / term are the arrays the last step
 * wrote; out is a device buffer uint8 [n_envs][arcle_packed_obs_size()], 16-byte aligned. */
int arcle_packed_obs_size(const arcle_env* env);
int arcle_pack_obs(arcle_env* env, const int32_t* reward, const uint8_t* term, uint8_t* out, void* stream);
/* destination of ARCLE_STEP_PACK_OBS: uint8 [n_envs][arcle_packed_obs_size()], 16-byte aligned (NULL uninstalls it) */
int arcle_set_packed_output(arcle_env* env, uint8_t* out);

/* Reads and (optionally) clears the sticky device status word (ARCLE_ST_*): one atomic exchange on the device, so a
 * bit raised by a kernel on another stream is never lost between the read and the clear.  Synchronises the stream.
 * Ordering of the setters (arcle_set_op_table, _task_table, _sampler, _truncation, _dense_output, _flat_output, _packed_output):
 * every launch takes its parameters — table pointers included — BY VALUE at the moment it is enqueued (or captured into a
 * hipGraph).  A setter therefore changes what LATER launches see and never what is already in flight or captured; the library's
 * own op-table copy is versioned (a new device buffer per arcle_set_op_table, the old ones are freed in arcle_destroy), and
 * caller-owned arrays (task table, sampler arrays, output buffers) must simply stay alive until the launches / graphs that were
 * given them have finished.  No setter synchronises. */
int arcle_get_status(arcle_env* env, uint32_t* status, int clear, void* stream);

/* Algorithmic HBM bytes (SURVEY.md §8d accounting) moved by all step launches (and the observation rows written by
 * arcle_flatten_obs / ARCLE_STEP_FLAT_OBS: planes + record read once, row written once) since the
 * last call with clear != 0; accumulated on device by the step kernel only when the handle
 * was created with accounting enabled via arcle_enable_accounting(env, 1). */
int arcle_enable_accounting(arcle_env* env, int on);
int arcle_get_accounting(arcle_env* env, uint64_t* bytes, uint64_t* steps, int clear, void* stream);
/* ... and next to the algorithmic figure `issued`: the bytes of every global-memory access the step kernel actually issued for
 * those steps (whole 16-byte lanes of every plane access incl. row padding, record / counters / action / outputs, observation rows,
 * task-table reads) — elided writes are not in it, re-reads are. */
int arcle_get_accounting_ex(arcle_env* env, uint64_t* bytes, uint64_t* issued, uint64_t* steps, int clear, void* stream);

const char* arcle_last_error(const arcle_env* env);
int arcle_abi_version(void);

/* (Diagnostic builds compiled with -DARCLE_TRACE_WAVES additionally export a per-wave timestamp dump used by tools/wavetrace.py; it does
 * not exist in the shipped library and is therefore not declared here.) */

#ifdef __cplusplus
}
#endif
#endif /* ARCLE_HIP_H */
