"""Host mirror of the device-side task draw (`arcle::draw_task`, arcle_amd/csrc/arcle_wave.h).

A draw is a pure function of (seed, GLOBAL env id, episode number): which problem, which of its pairs, and — when
augmentation is on — the np.rot90 count and the colour permutation (agents/env.py:31-42 of the reference).  Because
the key is the global env id, a batch sharded over any number of GPUs walks exactly the same task sequence per env
(SURVEY.md §8e).  The kernels are the product; this module only lets host code predict / log what they will draw.
"""
M64 = 0xFFFFFFFFFFFFFFFF
GOLD = 0x9E3779B97F4A7C15
AUG_PERMUTE, AUG_ROT90 = 1, 2


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def draw_task(seed, gid, episode, pair_cnt, aug_flags=0):
    """-> (problem index into pair_off/pair_cnt, pair index, rot90 count, permutation list of the colours 0..9)."""
    n = len(pair_cnt)
    z = mix64((seed + gid * GOLD + episode * 0xD1B54A32D192ED03) & M64)
    problem = (z >> 32) % n
    z = mix64((z + GOLD) & M64)
    sub = (z >> 32) % int(pair_cnt[problem])
    z = mix64((z + GOLD) & M64)
    k = int(z & 3) if aug_flags & AUG_ROT90 else 0
    perm = list(range(10))
    if aug_flags & AUG_PERMUTE:
        for i in range(9, 0, -1):
            z = mix64((z + GOLD) & M64)
            j = (z >> 32) % (i + 1)
            perm[i], perm[j] = perm[j], perm[i]
    return int(problem), int(sub), k, perm
