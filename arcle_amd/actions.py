"""Operation descriptors — the host-side mirror of the reference's `arcle.actions` package.

The reference builds its op table out of Python closures `op(state, action) -> None`
(/root/reference/arcle/actions/{color,object,critical}.py) and `AbstractARCEnv.create_operations`
returns the list (base.py:140-142).  Here every generator returns an `Operation`: it carries the same
`__name__` (so `op_names` come out identical, base.py:66) plus the uint32 descriptor the HIP step kernel
dispatches on (include/arcle_hip.h).  Tables may be re-ordered, truncated, wrapped (`reset_sel`,
`keep_sel`) or have slots swapped exactly like user code does with the reference
(agents/env.py:23-28, agents/wrapper.py:53-57).  An arbitrary Python callable `op(state, action)` in the table
cannot run inside the kernel: its slot becomes a device no-op (`OP_HOST`: the step is counted, nothing else
happens) and the env applies the callable on the host to a fetched state (envs/base.py, envs/vec.py) — the
SURVEY.md §8(b) "custom ops" contract; it is the slow path by construction.
"""

# op kinds / flags — include/arcle_hip.h
(OP_NONE, OP_COLOR, OP_FLOODFILL, OP_MOVE, OP_ROTATE, OP_FLIP, OP_COPY, OP_PASTE, OP_COPY_FROM_INPUT,
 OP_RESET_GRID, OP_RESIZE_GRID, OP_CROP_GRID, OP_RESIZE_TO_ANSWER, OP_SUBMIT, OP_HOST) = range(15)
OPF_RESET_SEL, OPF_KEEP_SEL = 1, 2


class Operation:
    """One slot of an env's operation table."""

    __slots__ = ("kind", "arg", "flags", "__name__")

    def __init__(self, kind, arg=0, flags=0, name=""):
        self.kind, self.arg, self.flags, self.__name__ = kind, arg, flags, name

    @property
    def desc(self):
        return self.kind | (self.arg << 8) | (self.flags << 16)

    def __call__(self, state, action):
        raise TypeError(
            f"{self.__name__} is a device operation descriptor; apply it with env.transition(state, action) "
            "or env.step(action) — the HIP kernel executes it, there is no host implementation")

    def __repr__(self):
        return f"<Operation {self.__name__} desc=0x{self.desc:06x}>"


def _check(cond, msg="invalid argument"):
    if not cond:
        raise AssertionError(msg)  # the reference asserts as well (object.py:175,226,261,289)


def gen_color(color):  # color.py:62-77
    return Operation(OP_COLOR, int(color) & 0xFF, 0, f"Color{color}")


def gen_flood_fill(color):  # color.py:79-103
    return Operation(OP_FLOODFILL, int(color) & 0xFF, 0, f"FloodFill{color}")


def gen_move(d=0):  # object.py:218-243
    _check(0 <= d < 4)
    return Operation(OP_MOVE, d, 0, f"Move_{'UDRL'[d]}")


def gen_rotate(k=1):  # object.py:167-216
    _check(0 < k < 4)
    return Operation(OP_ROTATE, k, 0, f"Rotate_{90 * k}")


def gen_flip(axis="H"):  # object.py:245-279
    _check(axis in ("H", "V", "D0", "D1"), "Invalid Axis")
    return Operation(OP_FLIP, ("H", "V", "D0", "D1").index(axis), 0, f"Flip_{axis}")


def gen_copy(source="I"):  # object.py:281-314
    _check(source in ("I", "O"), "Invalid Source grid")
    return Operation(OP_COPY, 0 if source == "I" else 1, 0, f"Copy_{source}")


def gen_paste(paste_blank=False):  # object.py:316-349
    return Operation(OP_PASTE, 1 if paste_blank else 0, 0, "Paste")


reset_grid = Operation(OP_RESET_GRID, 0, 0, "reset_grid")            # critical.py:8-17
copy_from_input = Operation(OP_COPY_FROM_INPUT, 0, 0, "copy_from_input")  # critical.py:19-29
resize_grid = Operation(OP_RESIZE_GRID, 0, 0, "resize_grid")         # critical.py:31-46
crop_grid = Operation(OP_CROP_GRID, 0, 0, "crop_grid")               # critical.py:48-66
resize_to_answer = Operation(OP_RESIZE_TO_ANSWER, 0, 0, "resize_to_answer")  # arcenv.py:31-35
submit = Operation(OP_SUBMIT, 0, 0, "submit")                        # base.py:172-183


def reset_sel(op):  # object.py:10-26 (functools.wraps keeps the name)
    return Operation(op.kind, op.arg, op.flags | OPF_RESET_SEL, op.__name__)


def keep_sel(op):  # object.py:28-41
    return Operation(op.kind, op.arg, op.flags | OPF_KEEP_SEL, op.__name__)


def is_submit(op):
    """`self.submit` bound method, as the reference's tables use it (o2arcenv.py:112)."""
    return getattr(op, "__self__", None) is not None and getattr(op, "__name__", "") == "submit"


def host_slots(operations):
    """Indices of table slots holding arbitrary Python callables (applied on the host)."""
    return [i for i, op in enumerate(operations) if not isinstance(op, Operation) and not is_submit(op)]


def table_descs(operations):
    """List[Operation | callable] -> list of uint32 descriptors for the HIP step kernel."""
    out = []
    for i, op in enumerate(operations):
        if is_submit(op):
            op = submit
        if isinstance(op, Operation):
            out.append(op.desc)
        elif callable(op):
            out.append(OP_HOST)  # device no-op slot; the env applies the callable on the host
        else:
            raise TypeError(f"operation table slot {i} ({op!r}) is neither an arcle_amd Operation nor a callable")
    return out
