"""O2ARCv2Env — mirror of /root/reference/arcle/envs/o2arcenv.py (the north-star env)."""
import numpy as np

from .. import actions as A
from .. import spaces
from ..loaders import ARCLoader, Loader
from .base import AbstractARCEnv


class O2ARCv2Env(AbstractARCEnv):
    """35-op table (o2arcenv.py:88-113); state adds selected, clip, clip_dim, object_states{7} (:16-34)."""
    KIND = "o2arc"

    def __init__(self, data_loader: Loader = None, max_grid_size=(30, 30), colors=10, max_trial=-1, render_mode=None,
                 render_size=None, device=None):
        super().__init__(data_loader if data_loader is not None else ARCLoader(), max_grid_size, colors, max_trial,
                         render_mode, render_size, device)

    @staticmethod
    def default_operations():
        R = A.reset_sel
        ops = [R(A.gen_color(i)) for i in range(10)]                        # :91
        ops += [R(A.gen_flood_fill(i)) for i in range(10)]                  # :92
        ops += [A.gen_move(i) for i in range(4)]                            # :95
        ops += [A.gen_rotate(1), A.gen_rotate(3), A.gen_flip("H"), A.gen_flip("V")]  # :96-99
        ops += [R(A.gen_copy("I")), R(A.gen_copy("O")), R(A.gen_paste(paste_blank=True))]  # :102-104
        ops += [R(A.copy_from_input), R(A.reset_grid), R(A.resize_grid)]    # :107-109
        ops += [A.submit]                                                   # :112
        return ops

    def create_operations(self):
        return self.default_operations()

    def create_state_space(self):  # :36-66
        old = super().create_state_space()
        S, H, W, C = spaces, self.H, self.W, self.colors
        new = {
            "selected": S.Box(0, 1, (H, W), dtype=np.int8),
            "clip": S.Box(0, C, (H, W), dtype=np.int8),
            "clip_dim": S.Box(low=np.array([0, 0]), high=np.array([H, W]), dtype=np.int8),
            "object_states": S.Dict({
                "active": S.MultiBinary(1),
                "object": S.Box(0, C, (H, W), dtype=np.int8),
                "object_sel": S.Box(0, 1, (H, W), dtype=np.int8),
                "object_dim": S.Box(low=np.array([0, 0]), high=np.array([H, W]), dtype=np.int8),
                "object_pos": S.Box(low=np.array([-128, -128]), high=np.array([127, 127]), dtype=np.int8),
                "background": S.Box(0, C, (H, W), dtype=np.int8),
                "rotation_parity": S.MultiBinary(1),
            })}
        new.update(old.spaces)
        return S.Dict(new)

    def init_info(self):  # :115-119
        info = super().init_info()
        info["steps"] = 0
        info["submit_count"] = 0
        return info


O2ARCEnv = O2ARCv2Env  # alias, envs/__init__.py:4
