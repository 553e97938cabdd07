from .base import AbstractARCEnv
from .arcenv import RawARCEnv, ARCEnv
from .o2arcenv import O2ARCv2Env
from .o2arcenv import O2ARCv2Env as O2ARCEnv
from .vec import ARCVecEnv

from .. import spaces as _spaces

if _spaces.HAVE_GYMNASIUM:
    from gymnasium.envs.registration import register, registry
    # the reference's ids (arcle/envs/__init__.py:7-25): `gym.make('ARCLE/O2ARCv2Env-v0', ...)` keeps working when this
    # package replaces the reference; the same entry points also under the ARCLE-AMD namespace (both packages installed)
    for _ns in ("ARCLE", "ARCLE-AMD"):
        for _id, _ep in ((f"{_ns}/RawARCEnv-v0", "arcle_amd.envs.arcenv:RawARCEnv"),
                         (f"{_ns}/ARCEnv-v0", "arcle_amd.envs.arcenv:ARCEnv"),
                         (f"{_ns}/O2ARCEnv-v2", "arcle_amd.envs:O2ARCEnv"),
                         (f"{_ns}/O2ARCv2Env-v0", "arcle_amd.envs.o2arcenv:O2ARCv2Env")):
            if _id not in registry:
                register(id=_id, entry_point=_ep)
