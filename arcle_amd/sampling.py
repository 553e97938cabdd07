"""Host mirror of the device-side task draw (`arcle::draw_task`, arcle_amd/csrc/arcle_wave.h).

A draw is a pure function of (seed, GLOBAL env id, episode number): which problem, which of its pairs, and — when
augmentation is on — the np.rot90 count and the colour permutation (agents/env.py:31-42 of the reference).  Because
the key is the global env id, a batch sharded over any number of GPUs walks exactly the same task sequence per env
(SURVEY.md §8e).  One rule on top of the draw: on a non-square max_grid_size a drawn quarter turn (k odd) that does not fit the
H x W plane is dropped by the device, i.e. the task is loaded with k & 2 (square grids — the reference's 30 x 30 — always fit).  The kernels are the product; this module only lets host code predict / log what they will draw.
"""
M64 = 0xFFFFFFFFFFFFFFFF
GOLD = 0x9E3779B97F4A7C15
AUG_PERMUTE, AUG_ROT90 = 1, 2


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def _mulhi32(r, n):
    return ((r & 0xFFFFFFFF) * int(n)) >> 32


def draw_task(seed, gid, episode, pair_cnt, aug_flags=0):
    """-> (problem index into pair_off/pair_cnt, pair index, rot90 count, permutation list of the colours 0..9).
    Two splitmix64 outputs: z0 gives the problem (high word) and the pair (low word) by multiply-shift range reduction, z1 the quarter
    turns (two low bits) and — its high word read as a 32-bit fraction, consumed digit by digit — the Fisher-Yates swaps."""
    n = len(pair_cnt)
    z0 = mix64((seed + gid * GOLD + episode * 0xD1B54A32D192ED03) & M64)
    problem = _mulhi32(z0 >> 32, n)
    sub = _mulhi32(z0, int(pair_cnt[problem]))
    z1 = mix64((z0 + GOLD) & M64)
    k = int(z1 & 3) if aug_flags & AUG_ROT90 else 0
    perm = list(range(10))
    if aug_flags & AUG_PERMUTE:
        r = z1 >> 32
        for i in range(9, 0, -1):
            j = _mulhi32(r, i + 1)
            r = (r * (i + 1)) & 0xFFFFFFFF
            perm[i], perm[j] = perm[j], perm[i]
    return int(problem), int(sub), k, perm


def draw_aug_batch(seed, gids, episodes, aug_flags):
    """The augmentation part of `draw_task` for many envs at once (NumPy uint64, same stream positions): -> (k uint8 [n],
    perm uint8 [n, 10]).  Used by ARCVecEnv.reset when the caller names the tasks (prob_index / subprob_index) but the env was built
    with `augment=`: env i then gets the rot90 count and colour permutation the device would have drawn for (seed, gid, episode)."""
    import numpy as np
    g = np.asarray(gids, np.uint64)
    e = np.asarray(episodes, np.uint64)

    def mix(z):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
    with np.errstate(over="ignore"):
        G = np.uint64(GOLD)
        z0 = mix(np.uint64(seed & M64) + g * G + e * np.uint64(0xD1B54A32D192ED03))  # problem / pair draw
        z1 = mix(z0 + G)                                                             # augmentation draw
        k = (z1 & np.uint64(3)).astype(np.uint8) if aug_flags & AUG_ROT90 else np.zeros(len(g), np.uint8)
        perm = np.tile(np.arange(10, dtype=np.uint8), (len(g), 1))
        if aug_flags & AUG_PERMUTE:
            rows = np.arange(len(g))
            r = z1 >> np.uint64(32)
            for i in range(9, 0, -1):
                j = ((r * np.uint64(i + 1)) >> np.uint64(32)).astype(np.int64)
                r = (r * np.uint64(i + 1)) & np.uint64(0xFFFFFFFF)
                a, b = perm[rows, i].copy(), perm[rows, j].copy()
                perm[rows, i], perm[rows, j] = b, a
    return k, perm
