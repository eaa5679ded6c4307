"""Layout contract of one block: index math, iteration ranges and per-step stencil extents.

Mirrors ref src/domain_ranges.jl (DomainRange, StepsRanges), src/blocking/blocking.jl:19-216 (BlockSize
and its index functions) and src/parameters.jl:988-1025 (compute_steps_ranges). Indices on the Python
side are 1-BASED like the reference (so these functions can be checked against the reference's own
index tests, ref test/blocking.jl:110-183); ``DomainRange.to_c()`` gives the 0-based ``armon_range``.
"""
from dataclasses import dataclass
from enum import IntEnum

from ._lib import Range


class Axis(IntEnum):      # ref src/blocking/blocking.jl (Axis.X = 1, Axis.Y = 2)
    X = 1
    Y = 2


class Side(IntEnum):      # ref Side.Left = 1, Right, Bottom, Top
    Left = 1
    Right = 2
    Bottom = 3
    Top = 4


def first_sides():
    return (Side.Left, Side.Bottom)


def first_side(axis):
    return Side.Left if axis == Axis.X else Side.Bottom


def last_side(axis):
    return Side.Right if axis == Axis.X else Side.Top


def sides_along(axis):
    return (first_side(axis), last_side(axis))


def axis_of(side):
    return Axis.X if side in (Side.Left, Side.Right) else Axis.Y


def opposite_of(side):
    return {Side.Left: Side.Right, Side.Right: Side.Left, Side.Bottom: Side.Top, Side.Top: Side.Bottom}[side]


@dataclass(frozen=True)
class StepRange:
    """Julia ``first:step:last`` (1-based, inclusive)."""
    first: int
    step: int
    last: int

    def __len__(self):
        if self.step > 0:
            return max(0, (self.last - self.first) // self.step + 1)
        return max(0, (self.first - self.last) // (-self.step) + 1)

    def __iter__(self):
        return iter(range(self.first, self.last + (1 if self.step > 0 else -1), self.step))

    @property
    def stop(self):
        n = len(self)
        return self.first + (n - 1) * self.step if n else self.first - self.step

    # ref src/domain_ranges.jl:7-14
    def shift(self, n=1):
        return StepRange(self.first + self.step * n, self.step, self.last + self.step * n)

    def expand(self, n=1):
        return StepRange(self.first, self.step, self.last + self.step * n)

    def prepend(self, n=1):
        return StepRange(self.first - self.step * n, self.step, self.last)

    def inflate(self, n=1):
        return StepRange(self.first - self.step * n, self.step, self.last + self.step * n)


@dataclass(frozen=True)
class DomainRange:
    """ref src/domain_ranges.jl:39-61: cell index = col[j] + row[i] - 1."""
    col: StepRange
    row: StepRange

    def __len__(self):
        return len(self.col) * len(self.row)

    @property
    def size(self):
        return (len(self.row), len(self.col))

    def first(self):
        return self.col.first + self.row.first - 1

    def last(self):
        return self.col.stop + self.row.stop - 1

    def __contains__(self, x):
        if not (self.first() <= x <= self.last()):
            return False
        ix = x - self.col.first + 1
        idc = ix // self.col.step
        ix -= idc * self.col.step
        return self.row.first <= ix <= self.row.stop

    def __iter__(self):
        for j in self.col:
            for i in self.row:
                yield j + i - 1

    def _along(self, axis, fname, n):
        if axis == Axis.X:
            return DomainRange(self.col, getattr(self.row, fname)(n))
        return DomainRange(getattr(self.col, fname)(n), self.row)

    def shift_dir(self, axis, n=1):
        return self._along(axis, "shift", n)

    def prepend_dir(self, axis, n=1):
        return self._along(axis, "prepend", n)

    def expand_dir(self, axis, n=1):
        return self._along(axis, "expand", n)

    def inflate_dir(self, axis, n=1):
        return self._along(axis, "inflate", n)

    def to_c(self):
        assert self.row.step == 1
        return Range(self.col.first - 1, self.col.step, len(self.col), self.row.first - 1, len(self.row))


@dataclass(frozen=True)
class BlockSize:
    """ref StaticBSize/DynamicBSize (src/blocking/blocking.jl:19-66): ``size`` includes the ghosts."""
    size: tuple
    ghosts: int

    @property
    def real_size(self):
        return (self.size[0] - 2 * self.ghosts, self.size[1] - 2 * self.ghosts)

    @property
    def n_cells(self):
        return self.size[0] * self.size[1]

    # ref src/blocking/blocking.jl:71-85
    def domain_range(self, bottom_left=(0, 0), top_right=(0, 0)):
        g, row = self.ghosts, self.size[0]
        block_start = row * (g - 1) + g

        def block_idx(I):
            return block_start + I[1] * row + I[0]

        first_I = (bottom_left[0] + 1, bottom_left[1] + 1)
        last_I = (top_right[0] + self.real_size[0], top_right[1] + self.real_size[1])
        col = StepRange(block_idx(first_I), row, block_idx(last_I))
        rowr = StepRange(1, 1, last_I[0] - first_I[0] + 1)
        return DomainRange(col, rowr)

    # ref :99-104
    def position(self, i):
        iy, ix = divmod(i - 1, self.size[0])
        return (ix - self.ghosts + 1, iy - self.ghosts + 1)

    # ref :129-131
    def lin_position(self, I):
        return (I[1] + self.ghosts - 1) * self.size[0] + (I[0] + self.ghosts)

    # ref :216
    def is_ghost(self, i, o=0):
        I = self.position(i)
        rs = self.real_size
        return not all(1 - o <= I[d] <= rs[d] + o for d in range(2))

    # ref :141-165
    def border_domain(self, side, single_strip=True):
        rs = self.real_size
        if side == Side.Left:
            bl, tr = (0, 0), (1 - rs[0], 0)
        elif side == Side.Right:
            bl, tr = (rs[0] - 1, 0), (0, 0)
        elif side == Side.Bottom:
            bl, tr = (0, 0), (0, 1 - rs[1])
        else:
            bl, tr = (0, rs[1] - 1), (0, 0)
        domain = self.domain_range(bl, tr)
        if single_strip:
            return domain
        if side in first_sides():
            return domain.expand_dir(axis_of(side), self.ghosts - 1)
        return domain.prepend_dir(axis_of(side), self.ghosts - 1)

    # ref :178-187
    def ghost_domain(self, side, single_strip=True):
        domain = self.border_domain(side)
        domain = domain.shift_dir(axis_of(side), -self.ghosts if side in first_sides() else self.ghosts)
        if single_strip:
            return domain
        if side in first_sides():
            return domain.expand_dir(axis_of(side), self.ghosts - 1)
        return domain.prepend_dir(axis_of(side), self.ghosts - 1)

    # ref :197-210
    def stride_along(self, axis):
        return 1 if axis == Axis.X else self.size[0]

    def size_along(self, axis_or_side):
        ax = axis_or_side if isinstance(axis_or_side, Axis) else axis_of(axis_or_side)
        return self.size[int(ax) - 1]

    def real_size_along(self, axis_or_side):
        ax = axis_or_side if isinstance(axis_or_side, Axis) else axis_of(axis_or_side)
        return self.real_size[int(ax) - 1]

    def real_face_size(self, axis_or_side):
        rs = self.real_size
        return (rs[0] * rs[1]) // self.real_size_along(axis_or_side)


@dataclass(frozen=True)
class StepsRanges:
    """ref src/domain_ranges.jl:96-105: corner offsets (bottom_left, top_right) per solver step."""
    direction: Axis
    real_domain: tuple
    full_domain: tuple
    EOS: tuple
    fluxes: tuple
    cell_update: tuple
    advection: tuple
    projection: tuple


def compute_steps_ranges(axis, ghosts, projection_stencil_width):
    """ref src/parameters.jl:992-1025."""
    extra = projection_stencil_width
    real = ((0, 0), (0, 0))
    full = ((-ghosts, -ghosts), (ghosts, ghosts))
    if axis == Axis.X:
        fl = ((-extra, 0), (extra + 1, 0))
        cu = ((-extra, 0), (extra, 0))
        ad = ((0, 0), (1, 0))
    else:
        fl = ((0, -extra), (0, extra + 1))
        cu = ((0, -extra), (0, extra))
        ad = ((0, 0), (0, 1))
    return StepsRanges(axis, real, full, real, fl, cu, ad, real)
