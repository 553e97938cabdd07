"""BBoxWrapper / PointWrapper — the action-wrapper plugin points of /root/reference/arcle/wrappers/bbox.py.

For a single env they keep the reference's interface (tuple in, dict action out).  The mask arithmetic of
`action()` is also fused into the step kernel: the batched path (`ARCVecEnv.step_bbox/step_point`) ships the
raw tuples to the device (16 / 8 bytes per env instead of an HxW mask)."""
import numpy as np

from . import spaces


class BBoxWrapper(spaces.ActionWrapper):
    def __init__(self, env):
        super().__init__(env)
        e = env.unwrapped  # do not rely on Wrapper.__getattr__ forwarding
        self.H, self.W, self.operations = e.H, e.W, e.operations
        self.action_space = spaces.Tuple((spaces.Discrete(self.H), spaces.Discrete(self.W), spaces.Discrete(self.H),
                                          spaces.Discrete(self.W), spaces.Discrete(len(self.operations))))

    def action(self, action):
        x1, y1, x2, y2, op = action  # bbox.py:24
        selection = np.zeros((self.H, self.W), dtype=np.int8)
        x1, x2 = min(x1, x2), max(x1, x2)
        y1, y2 = min(y1, y2), max(y1, y2)
        selection[x1:x2 + 1, y1:y2 + 1] = 1
        return {"selection": selection, "operation": op}


class PointWrapper(spaces.ActionWrapper):
    def __init__(self, env):
        super().__init__(env)
        e = env.unwrapped
        self.H, self.W, self.operations = e.H, e.W, e.operations
        self.action_space = spaces.Tuple((spaces.Discrete(self.H), spaces.Discrete(self.W),
                                          spaces.Discrete(len(self.operations))))

    def action(self, action):
        x, y, op = action  # bbox.py:45
        selection = np.zeros((self.H, self.W), dtype=np.int8)
        selection[x, y] = 1
        return {"selection": selection, "operation": op}
