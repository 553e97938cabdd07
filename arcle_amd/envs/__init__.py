from .base import AbstractARCEnv
from .arcenv import RawARCEnv, ARCEnv
from .o2arcenv import O2ARCv2Env
from .o2arcenv import O2ARCv2Env as O2ARCEnv
from .vec import ARCVecEnv

from .. import spaces as _spaces

if _spaces.HAVE_GYMNASIUM:  # pragma: no cover - gymnasium is optional
    from gymnasium.envs.registration import register, registry
    # the reference's ids (arcle/envs/__init__.py:7-25) under the ARCLE-AMD namespace
    for _id, _ep in (("ARCLE-AMD/RawARCEnv-v0", "arcle_amd.envs.arcenv:RawARCEnv"),
                     ("ARCLE-AMD/ARCEnv-v0", "arcle_amd.envs.arcenv:ARCEnv"),
                     ("ARCLE-AMD/O2ARCEnv-v2", "arcle_amd.envs:O2ARCEnv"),
                     ("ARCLE-AMD/O2ARCv2Env-v0", "arcle_amd.envs.o2arcenv:O2ARCv2Env")):
        if _id not in registry:
            register(id=_id, entry_point=_ep)
