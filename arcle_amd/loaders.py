"""Task loaders — the `Loader` plugin point of the reference (/root/reference/arcle/loaders/loader.py),
kept in Python as north_star asks.  `Loader.pick` feeds `EnvBatch.set_tasks`, which packs the tasks into
the device planes."""
import glob
import json
import os
from abc import ABCMeta, abstractmethod

import numpy as np


class Loader(metaclass=ABCMeta):
    """Abstract ARC-like problem loader (loader.py:8-57): __init__ calls get_path() then parse();
    parse() returns a list of (train_inputs, train_outputs, test_inputs, test_outputs, description)."""
    _pathlist = []

    def __init__(self, rng: np.random.Generator = None, **kwargs):
        self.rng = rng
        self._pathlist = self.get_path(**kwargs)
        self.data = self.parse(**kwargs)

    @abstractmethod
    def get_path(self, **kwargs):
        pass

    @abstractmethod
    def parse(self, **kwargs):
        pass

    def pick(self, data_index=None, **kwargs):
        """loader.py:41-57 — random (global np.random or self.rng) or indexed task."""
        assert self.data is not None and len(self.data) > 0, "Dataset wasn't loaded properly"
        sel, max_index = data_index, len(self.data)
        if data_index is None:
            sel = np.random.randint(0, max_index) if self.rng is None else self.rng.integers(0, max_index)
        assert 0 <= sel < max_index, f"Problem indices should be in [0, {max_index})."
        return self.data[sel]


def _parse_arc_json(fp, null_to_zero=False):
    txt = fp.read()
    if null_to_zero:
        txt = txt.replace("null", '"0"')  # MiniARC quirk, loader.py:139
    problem = json.loads(txt)
    ti = [np.array(d["input"], dtype=np.int8) for d in problem["train"]]
    to = [np.array(d["output"], dtype=np.int8) for d in problem["train"]]
    ei = [np.array(d["input"], dtype=np.int8) for d in problem["test"]]
    eo = [np.array(d["output"], dtype=np.int8) for d in problem["test"]]
    return ti, to, ei, eo


class ARCLoader(Loader):
    """Original ARC (loader.py:60-113).  `root` defaults to <package>/arcs/ARC/data like the reference's
    submodule layout; the dataset itself is not vendored (drop it there or pass root=...)."""

    def __init__(self, train=True, root=None):
        self._root = root
        super().__init__(train=train)

    def get_path(self, **kwargs):
        base = self._root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "arcs", "ARC", "data")
        self.train = kwargs["train"]
        pathlist = glob.glob(os.path.join(base, "training" if self.train else "evaluation", "*.json"))
        pathlist.sort()
        return pathlist

    def parse(self, **kwargs):
        dat = []
        for p in self._pathlist:
            with open(p) as fp:
                ti, to, ei, eo = _parse_arc_json(fp)
                dat.append((ti, to, ei, eo, {"id": os.path.basename(fp.name).split(".")[0]}))
        return dat


class MiniARCLoader(Loader):
    """Mini-ARC (loader.py:116-157)."""

    def __init__(self, root=None):
        self._root = root
        super().__init__()

    def get_path(self, **kwargs):
        base = self._root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "arcs", "Mini-ARC", "data", "MiniARC")
        pathlist = glob.glob(os.path.join(base, "*.json"))
        pathlist.sort(key=lambda fn: fn.split("_")[-1])
        return pathlist

    def parse(self, **kwargs):
        dat = []
        for p in self._pathlist:
            with open(p) as fp:
                ti, to, ei, eo = _parse_arc_json(fp, null_to_zero=True)
                fns = os.path.basename(fp.name).split("_")
                desc = {"id": fns[-1].split(".")[-2], "description": " ".join(fns[0:-1]).strip()}
                dat.append((ti, to, ei, eo, desc))
        return dat


class SyntheticLoader(Loader):
    """Deterministic ARC-shaped random tasks (there is no dataset in the image).
    Task t has `pairs` demo pairs and one test pair; grids are (h,w) in [min_size,max_size]^2 with colours
    0..colors-1; the answer equals the input with probability `p_same`, else it is an unrelated grid."""

    def __init__(self, n_tasks=64, max_size=(30, 30), min_size=(1, 1), colors=10, pairs=2, p_same=0.5, seed=0):
        self._cfg = (n_tasks, tuple(max_size), tuple(min_size), colors, pairs, p_same, seed)
        super().__init__()

    def get_path(self, **kwargs):
        return [""]

    def parse(self, **kwargs):
        n_tasks, (H, W), (h0, w0), colors, pairs, p_same, seed = self._cfg
        g = np.random.default_rng(seed)

        def grid():
            h, w = g.integers(h0, H + 1), g.integers(w0, W + 1)
            a = g.integers(0, colors, (h, w)).astype(np.int8)
            if g.random() < 0.5:
                a *= (g.random((h, w)) < 0.5)
            return a

        dat = []
        for t in range(n_tasks):
            ins = [grid() for _ in range(pairs + 1)]
            outs = [a.copy() if g.random() < p_same else grid() for a in ins]
            dat.append((ins[:pairs], outs[:pairs], ins[pairs:], outs[pairs:], {"id": f"synthetic{t:05d}"}))
        return dat
