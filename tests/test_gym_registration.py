"""With a `gymnasium` on the path the package registers the reference's four env ids (arcle/envs/__init__.py:7-25) — plus the same
under the ARCLE-AMD namespace — and builds its spaces from gymnasium.spaces.  gymnasium itself is not installed in this image, so the
test runs a child interpreter with the build-owned stand-in of oracle/stubs on its path (test infrastructure; constructing an env
needs no GPU — the device batch is only created by reset / step)."""
import os
import subprocess
import sys

import backends as B

CHILD = r"""
import gymnasium as gym
import arcle_amd
from arcle_amd import spaces
from arcle_amd.envs import AbstractARCEnv
from arcle_amd.loaders import SyntheticLoader
from arcle_amd.wrappers import BBoxWrapper, PointWrapper
assert spaces.HAVE_GYMNASIUM and spaces.Box is gym.spaces.Box and spaces.Env is gym.Env
from gymnasium.envs.registration import registry
want = {"RawARCEnv-v0": "RawARCEnv", "ARCEnv-v0": "ARCEnv", "O2ARCEnv-v2": "O2ARCv2Env", "O2ARCv2Env-v0": "O2ARCv2Env"}
for ns in ("ARCLE", "ARCLE-AMD"):
    for suffix, cls in want.items():
        assert f"{ns}/{suffix}" in registry, (ns, suffix)
        env = gym.make(f"{ns}/{suffix}", data_loader=SyntheticLoader(n_tasks=3, seed=1, max_size=(5, 5)), max_grid_size=(5, 5))
        assert isinstance(env, AbstractARCEnv) and type(env).__module__.startswith("arcle_amd.envs") and type(env).__name__ == cls
        assert isinstance(env.observation_space, gym.spaces.Dict) and isinstance(env.action_space, gym.spaces.Dict)
        assert env.action_space["operation"].n == len(env.operations) == {"RawARCEnv": 12, "ARCEnv": 27, "O2ARCv2Env": 35}[cls]
env = BBoxWrapper(gym.make("ARCLE/O2ARCv2Env-v0", data_loader=SyntheticLoader(n_tasks=3, seed=1, max_size=(5, 5)), max_grid_size=(5, 5)))
a = env.action_space.sample()          # examples/example_bbox.py:12-13
assert len(a) == 5 and env.action(a)["selection"].shape == (5, 5)
assert len(PointWrapper(env.unwrapped).action_space.sample()) == 3
print("registered:", len(registry))
"""


def test_reference_gym_ids_resolve_to_arcle_amd():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(B.ROOT, "oracle", "stubs"), B.ROOT]))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "registered: 8" in out.stdout
