"""RawARCEnv / ARCEnv — mirrors of /root/reference/arcle/envs/arcenv.py."""
import numpy as np

from .. import actions as A
from .. import spaces
from ..loaders import ARCLoader, Loader
from .base import AbstractARCEnv


class RawARCEnv(AbstractARCEnv):
    """12 ops: Color0-9, ResizeToAnswer, Submit (arcenv.py:26-41); base state only."""
    KIND = "raw"
    _RECORD_ACTION_FIRST = True  # arcenv.py:62-64

    def __init__(self, data_loader: Loader = None, max_grid_size=(30, 30), colors=10, max_trial=-1, render_mode=None,
                 render_size=None, device=None):
        super().__init__(data_loader if data_loader is not None else ARCLoader(), max_grid_size, colors, max_trial,
                         render_mode, render_size, device)

    @staticmethod
    def default_operations():
        return [A.gen_color(i) for i in range(10)] + [A.resize_to_answer, A.submit]

    def create_operations(self):
        return self.default_operations()

    def init_info(self):
        info = super().init_info()
        info["steps"] = 0
        return info

    def _step_flags(self):
        """reset_on_submit in RawARCEnv.step (arcenv.py:62-76): `state = self.current_state` is bound BEFORE the op runs, so the reward
        and `terminated` the step returns are those of the state that was submitted, while the observation is the re-initialised
        one (the other classes read all three from the new state, o2arcenv.py:138-147).  The kernel therefore steps without
        RESET_ON_SUBMIT and AbstractARCEnv.step re-initialises the state afterwards."""
        return 0


class ARCEnv(AbstractARCEnv):
    """The 27 ops arcenv.py:123-137 installs (Color, FloodFill, CopyI/O, Paste, CopyFromInput, ResetGrid,
    ResizeGrid, Submit) — the reference class itself cannot be constructed because its table keeps 8 `None`
    slots (arcenv.py:120 -> base.py:66 AttributeError); here the table is those 27 ops and
    Submit (index 26) is the rewarded last op.  State = base + clip, clip_dim (arcenv.py:81-89)."""
    KIND = "arc"

    def __init__(self, data_loader: Loader = None, max_grid_size=(30, 30), colors=10, max_trial=3, render_mode=None,
                 render_size=None, device=None):
        super().__init__(data_loader if data_loader is not None else ARCLoader(), max_grid_size, colors, max_trial,
                         render_mode, render_size, device)

    @staticmethod
    def default_operations():
        ops = [A.gen_color(i) for i in range(10)] + [A.gen_flood_fill(i) for i in range(10)]
        ops += [A.gen_copy("I"), A.gen_copy("O"), A.gen_paste(True)]
        ops += [A.copy_from_input, A.reset_grid, A.resize_grid, A.submit]
        return ops

    def create_operations(self):
        return self.default_operations()

    def create_state_space(self):
        old = super().create_state_space()
        new = {"clip": spaces.Box(0, self.colors, (self.H, self.W), dtype=np.int8),
               "clip_dim": spaces.Box(low=np.array([0, 0]), high=np.array([self.H, self.W]), dtype=np.int8)}
        new.update(old.spaces)
        return spaces.Dict(new)

    def init_info(self):
        info = super().init_info()
        info["steps"] = 0
        info["submit_count"] = 0
        return info
