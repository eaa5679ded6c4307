"""Property tests of the host-side index math (no GPU): tile partition, halo ranges of the library against index sets
built the slow way, the Y run-length model's contract. hypothesis draws the shapes."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import armon_amd
from armon_amd import _lib
from armon_amd.parameters import ArmonParameters


def _tile(N, P, cc, nghost=4):
    p = ArmonParameters.__new__(ArmonParameters)
    p.N, p.nghost, p.proc_dims, p.cart_coords, p.projection_scheme = N, nghost, P, cc, "euler_2nd"
    p._init_indexing()
    return p


@settings(max_examples=200, deadline=None)
@given(st.integers(8, 400), st.integers(8, 400), st.integers(1, 6), st.integers(1, 6))
def test_tiles_cover_the_grid_exactly_once(nx, ny, px, py):
    """ref src/parameters.jl:673-697: N ÷ P cells per tile, the remainder on the last one of each axis; the origins
    tile the global grid without gap or overlap."""
    if nx // px < 4 or ny // py < 4:
        with pytest.raises(armon_amd.SolverException):
            for cx in range(px):
                for cy in range(py):
                    _tile((nx, ny), (px, py), (cx, cy))
        return
    seen = np.zeros((ny, nx), dtype=np.int32)
    for cx in range(px):
        for cy in range(py):
            p = _tile((nx, ny), (px, py), (cx, cy))
            ox, oy = p.N_origin[0] - 1, p.N_origin[1] - 1
            assert p.N[0] == nx // px + (nx % px if cx == px - 1 else 0)
            assert p.N[1] == ny // py + (ny % py if cy == py - 1 else 0)
            seen[oy:oy + p.N[1], ox:ox + p.N[0]] += 1
    assert (seen == 1).all()


def _cells(r):
    """Linear cell indices of an armon_range, row by row."""
    return [r.col_start + j * r.col_step + r.row_start + k for j in range(r.col_len) for k in range(r.row_len)]


@settings(max_examples=150, deadline=None)
@given(st.integers(1, 40), st.integers(1, 40), st.integers(1, 5), st.integers(0, 3))
def test_halo_ranges_are_the_ghost_layers_and_their_mirror_images(nx, ny, g, side):
    """armon_hip_halo_ranges: `ghost` = exactly the g layers outside the side (real extent along the side, no corners),
    `border` = the g real layers next to it, both with face × g cells — against index sets built cell by cell."""
    if (nx if side < 2 else ny) < g:
        return                                   # a tile thinner than its ghost layers cannot be exchanged (checked elsewhere)
    L = armon_amd.lib()
    b, gh, face = _lib.Range(), _lib.Range(), C.c_int64()
    assert L.armon_hip_halo_ranges(nx, ny, g, side, C.byref(b), C.byref(gh), C.byref(face)) == 0
    row = nx + 2 * g
    idx = lambda ix, iy: (iy + g) * row + ix + g             # 0-based real coordinates, ghosts negative
    if side == 0:      # Left
        want_b = {idx(ix, iy) for iy in range(ny) for ix in range(g)}
        want_g = {idx(ix, iy) for iy in range(ny) for ix in range(-g, 0)}
    elif side == 1:    # Right
        want_b = {idx(ix, iy) for iy in range(ny) for ix in range(nx - g, nx)}
        want_g = {idx(ix, iy) for iy in range(ny) for ix in range(nx, nx + g)}
    elif side == 2:    # Bottom
        want_b = {idx(ix, iy) for iy in range(g) for ix in range(nx)}
        want_g = {idx(ix, iy) for iy in range(-g, 0) for ix in range(nx)}
    else:              # Top
        want_b = {idx(ix, iy) for iy in range(ny - g, ny) for ix in range(nx)}
        want_g = {idx(ix, iy) for iy in range(ny, ny + g) for ix in range(nx)}
    got_b, got_g = _cells(b), _cells(gh)
    assert len(got_b) == len(set(got_b)) == face.value * g and set(got_b) == want_b
    assert len(got_g) == len(set(got_g)) == face.value * g and set(got_g) == want_g
    assert face.value == (ny if side < 2 else nx)
    assert max(got_g) < (nx + 2 * g) * (ny + 2 * g) and min(got_g) >= 0


@settings(max_examples=100, deadline=None)
@given(st.integers(4, 300), st.integers(4, 300), st.integers(1, 5), st.integers(1, 5))
def test_neighbour_faces_have_the_same_size(nx, ny, px, py):
    """What a tile sends across a side is what its neighbour expects there (the native exchange checks it at run time:
    'tiles disagree on the size of their common face')."""
    if nx // px < 4 or ny // py < 4:
        return
    tiles = {(cx, cy): _tile((nx, ny), (px, py), (cx, cy)) for cx in range(px) for cy in range(py)}
    for (cx, cy), p in tiles.items():
        if cx + 1 < px:
            assert p.N[1] == tiles[(cx + 1, cy)].N[1]          # left/right neighbours share their height
        if cy + 1 < py:
            assert p.N[0] == tiles[(cx, cy + 1)].N[0]          # bottom/top neighbours share their width
