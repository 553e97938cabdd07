// arcle_big_params.h — the kernel-argument block of the workgroup-per-env kernels (arcle_big.h) and the host-side entry points of
// arcle_big.hip; shared with arcle_hip.hip, which routes handles of more than ARCLE_MAX_CELLS cells here.
#pragma once
#include <stdint.h>

#include "../../include/arcle_hip.h"

#ifndef ARCLE_BIG_HD
#define ARCLE_BIG_HD
#endif

namespace arcle_big {

enum { ING_MASK = 0, ING_BBOX = 1, ING_POINT = 2, ING_BBOX5 = 3, ING_BITS = 4 };  // (= enum arcle_ingress)
enum { MAX_SIDE = 127, MAX_PS = (127 * 127 + 127) & ~127, MIN_THREADS = 16, FILL_INNER = 8 };

struct BigParams {
  int8_t* plane[ARCLE_N_PLANES];
  int8_t* rec;
  int32_t* cnt;
  const int32_t* op;
  const void* sel;  // int8 [N][P] | int32 [N][4] | int32 [N][2] | int32 [N][5]
  int32_t* reward;
  uint8_t* term;
  int32_t n_envs, H, W, P, PS;
  int32_t n_ops, max_trial, ingress;
  uint32_t flags;
  int32_t step_limit;
  uint32_t* status;
  const uint32_t* d_ops;  // ARCLE_MAX_OPS + 1 entries, slot n_ops empty
  uint8_t* trunc;
  int8_t* flat_out;
  int32_t flat_stride, flat_filter, flat_tail, flat_seq;
  uint8_t* pack_out;
  // reset kernels
  const uint8_t* rmask;
  const int32_t* task_idx;
  const int8_t *tbl_in, *tbl_ans, *tbl_in_dim, *tbl_ans_dim;
  int32_t n_tasks;
  uint64_t seed;
  int64_t env_base;
  int32_t* episode;
  int32_t* cur_task;
  const int32_t *pair_off, *pair_cnt;
  int32_t n_problems;
  // state rows in
  const int8_t* rows_in;
  int32_t rows_in_stride;
  // arcle_transition_rows: the launch works on SCRATCH envs (one per row); the task side of row r — answer plane, answer_dim — is that of
  // resident env src_env[r] (NULL = env r)
  int32_t n_resident;
  const int32_t* src_env;
  const int8_t* res_answer;  // the handle's own answer plane / records
  const int8_t* res_rec;
  // task augmentation at reset (agents/env.py:31-42): ARCLE_AUG_* flags of the device draw, or explicit per-env arrays
  uint32_t aug_flags;
  const uint8_t* aug_k;     // uint8 [n_envs]: np.rot90 count
  const uint8_t* aug_perm;  // uint8 [n_envs][16]: perm[c] for colour c < 10
  uint32_t* acct;  // arcle_enable_accounting: uint32 [2][n_envs] — bytes without the row padding / bytes of every access issued, per env
  int32_t* dense;  // ARCLE_STEP_DENSE: int32 [n_envs][2] = (cells of the grid that match the answer inside the common rectangle, total cells)
  uint32_t w_magic;  // 2^21 / W + 1: cell index -> row by a multiply and a shift (arcle_big.h div_w); filled in by the launchers (with_magic)
};
ARCLE_BIG_HD inline BigParams with_magic(const BigParams& p) {
  BigParams q = p;
  q.w_magic = (1u << 21) / (uint32_t)(p.W > 0 ? p.W : 1) + 1u;
  return q;
}

// bytes of LDS one workgroup needs: four staging planes + the reduction block + two row boards of 128 x 128 bits + a row's scalars and layout
// (the flood fill's two row boards, 16 bytes per plane row each, live in the B and C tiles — idle during a fill — whenever a tile holds them:
// 16 * H <= PS, i.e. always for W >= 16; only narrow planes get 4 KB of their own)
ARCLE_BIG_HD inline bool boards_in_tiles(int PS, int H) { return 16 * H <= PS; }
enum { LDS_GUARD = 16, LEAN_MAX_THREADS = 512 };  // bytes in front of the first tile (arcle_big.h shifted16 reads whole 5-word windows)
ARCLE_BIG_HD inline int lds_bytes(int PS, int H) { return LDS_GUARD + 4 * PS + 64 + 256 + (boards_in_tiles(PS, H) ? 0 : 2 * 128 * 16); }  // 65 360 at 127 x 127

ARCLE_BIG_HD inline int flat_len(int P, bool o2, bool clip, int filtered) {
  if (filtered) return 3 * P + 10;
  return 2 * P + 6 + (clip ? P + 2 : 0) + (o2 ? 4 * P + 6 : 0);
}
ARCLE_BIG_HD inline int packed_stride(int P) { return (P + 7 + 15) & ~15; }

// host side (arcle_big.hip): one workgroup of 256 threads per env on `stream`; return a hipError_t as int (0 = success)
int launch_step(const BigParams& p, void* stream);
int launch_reset(const BigParams& p, int mode, void* stream);     // 0 arcle_reset, 1 arcle_reset_from_table, 2 arcle_reset_sampled
int launch_rows(const BigParams& p, int mode, void* stream);      // 0 flat rows of the resident state, 1 packed rows
int launch_set_rows(const BigParams& p, void* stream);
int workgroup_threads(int PS);  // threads per workgroup the reset / row launches (and the generic step kernel) use for a plane stride
int step_threads(const BigParams& p);  // ... and a step launch with these parameters (flags, ingress, accounting)

}  // namespace arcle_big
