"""Text output / input in the reference's file format, and the per-step comparison against dumps.

Mirrors ref src/io.jl: ``write_blocks_to_file`` (:3-27, one line ``x, y, ρ, u, v, p`` per cell in ascending
(x, y) order, ``%#{p+7}.{p}e`` fields joined by ", ", a blank line between rows for gnuplot's pm3d),
``read_data_from_file`` (:30-43), ``build_file_path`` (:44-58, ``_{cx}×{cy}`` suffix per sub-domain),
``write/read_time_step_file`` (:84-105) and ``step_checkpoint`` / ``compare_with_file`` (:113-227) which dump
or diff every sub-step of the staged solver cycle (``compare`` / ``is_ref`` options). The golden files of
ref test/reference_data use the same cell format after a ``dt, cycles`` header line
(ref test/reference_data/reference_functions.jl:37-50).
"""
import os

import numpy as np

SAVED_VARS = ("x", "y", "rho", "u", "v", "p")      # ref src/blocking/blocks.jl:49


def _fmt(precision):
    return f"%#{precision + 7}.{precision}e"


def build_file_path(params, file_name):
    path = os.path.join(params.output_dir, file_name)
    if params.is_root and not os.path.isdir(params.output_dir):
        os.makedirs(params.output_dir, exist_ok=True)
    if params.use_MPI:
        path += "_" + "×".join(str(c) for c in params.cart_coords)
    return path


def _rows(params, ghosts):
    """(first, last) 0-based array indices of the rows / columns written (ghosts only where the tile
    touches the global boundary, like the reference's ``global_ghosts``)."""
    g = params.nghost
    nx, ny = params.N
    if not ghosts:
        return (g, g + nx), (g, g + ny)
    P, cc = params.proc_dims, params.cart_coords
    x0 = 0 if cc[0] == 0 else g
    x1 = nx + 2 * g if cc[0] == P[0] - 1 else g + nx
    y0 = 0 if cc[1] == 0 else g
    y1 = ny + 2 * g if cc[1] == P[1] - 1 else g + ny
    return (x0, x1), (y0, y1)


def write_blocks_to_file(params, host, file, vars=SAVED_VARS, global_ghosts=False, for_3D=True):
    """``host`` = dict name → flat ghosted numpy array (``BlockGrid.device_to_host()``)."""
    fmt = ", ".join([_fmt(params.output_precision)] * len(vars)) + "\n"
    sx = params.N[0] + 2 * params.nghost
    (x0, x1), (y0, y1) = _rows(params, global_ghosts)
    cols = [np.asarray(host[v]).reshape(-1, sx) for v in vars]
    for j in range(y0, y1):
        if j != y0 and for_3D:
            file.write("\n")                      # separate rows to use pm3d plotting with gnuplot
        rows = [c[j, x0:x1] for c in cols]
        file.write("".join(fmt % vals for vals in zip(*rows)))


def read_data_from_file(params, host, file, vars=SAVED_VARS, global_ghosts=False):
    sx = params.N[0] + 2 * params.nghost
    (x0, x1), (y0, y1) = _rows(params, global_ghosts)
    vals = np.array([[float(t) for t in line.split(",")] for line in file if line.strip()], dtype=np.float64)
    expected = (x1 - x0) * (y1 - y0)
    if vals.shape != (expected, len(vars)):
        raise ValueError(f"expected {expected} lines of {len(vars)} values, found an array of shape {vals.shape}")
    vals = vals.reshape(y1 - y0, x1 - x0, len(vars))
    for k, v in enumerate(vars):
        np.asarray(host[v]).reshape(-1, sx)[y0:y1, x0:x1] = vals[:, :, k]


def write_sub_domain_file(params, grid, file_name, no_msg=False, vars=SAVED_VARS):
    path = build_file_path(params, file_name)
    host = grid.device_to_host(vars)
    with open(path, "w") as f:
        write_blocks_to_file(params, host, f, vars=vars, global_ghosts=params.write_ghosts)
    if not no_msg and params.is_root and params.silent < 2:
        print(f"\nWrote to files {path}_*x*")
    return path


def write_slices_files(params, grid, file_name, vars=SAVED_VARS):
    """``write_slices=true`` (ref src/parameters.jl:229-232, call at src/solver.jl:508): "all saved_vars() to 3 output
    files, one for the middle X row, another for the middle Y column, and another for the diagonal; with write_ghosts the
    ghost cells too". The reference calls ``write_slices_files`` but does not define it anywhere in src/, so the file
    names are this backend's: ``<file>_X`` (the row j = Ny÷2), ``<file>_Y`` (the column i = Nx÷2), ``<file>_diag`` (cells
    (k, k), k < min(Nx, Ny)), each in the cell format of ``write_blocks_to_file``. Single block only (a tile writes the
    slices of its own sub-domain, with the sub-domain suffix of ``build_file_path``)."""
    fmt = ", ".join([_fmt(params.output_precision)] * len(vars)) + "\n"
    sx = params.N[0] + 2 * params.nghost
    g = params.nghost
    (x0, x1), (y0, y1) = _rows(params, params.write_ghosts)
    host = grid.device_to_host(vars)
    cols = [np.asarray(host[v]).reshape(-1, sx) for v in vars]
    jm, im = g + params.N[1] // 2, g + params.N[0] // 2
    n_diag = min(x1 - x0, y1 - y0)
    picks = {"X": [(jm, i) for i in range(x0, x1)], "Y": [(j, im) for j in range(y0, y1)],
             "diag": [(y0 + k, x0 + k) for k in range(n_diag)]}
    paths = []
    for tag, cells in picks.items():
        path = build_file_path(params, f"{file_name}_{tag}")
        with open(path, "w") as f:
            f.write("".join(fmt % tuple(c[j, i] for c in cols) for j, i in cells))
        paths.append(path)
    if params.is_root and params.silent < 2:
        print(f"\nWrote slices to {', '.join(paths)}")
    return paths


def prepare_animation_dir(params):
    """ref src/solver.jl:436-442: the root empties (or creates) the ``anim`` directory before the run."""
    if params.animation_step == 0 or not params.is_root:
        return
    d = os.path.join(params.output_dir, "anim")
    if os.path.isdir(d):
        for name in os.listdir(d):
            os.remove(os.path.join(d, name))
    else:
        os.makedirs(d)


def write_animation_frame(params, grid):
    """ref src/solver.jl:373-378, called after ``next_cycle!``: every ``animation_step`` cycles the saved_vars go to
    ``anim/<output_file>_<frame:03d>`` as with write_output."""
    cycle = grid.global_dt.cycle
    if params.animation_step == 0 or (cycle - 1) % params.animation_step != 0:
        return None
    frame = (cycle - 1) // params.animation_step
    return write_sub_domain_file(params, grid, os.path.join("anim", params.output_file) + f"_{frame:03d}", no_msg=True)


def read_sub_domain_file(params, file_name, vars=SAVED_VARS):
    """Returns dict name → flat ghosted array (cells not in the file are NaN)."""
    n = params.block_size.n_cells
    host = {v: np.full(n, np.nan) for v in vars}
    with open(build_file_path(params, file_name)) as f:
        read_data_from_file(params, host, f, vars=vars, global_ghosts=params.write_ghosts)
    return host


def write_time_step_file(params, dt, file_name):
    with open(build_file_path(params, file_name), "w") as f:
        f.write(_fmt(params.output_precision) % dt + "\n")


def read_time_step_file(params, file_name):
    with open(build_file_path(params, file_name)) as f:
        return float(f.read().strip())


def read_reference_file(path, N):
    """A golden file of ref test/reference_data: header ``dt, cycles`` then the saved_vars of the real cells.
    Returns (dt, cycles, dict name → (Ny, Nx) array)."""
    with open(path) as f:
        head = f.readline().split(",")
        vals = np.array([[float(t) for t in line.split(",")] for line in f if line.strip()])
    vals = vals.reshape(N[1], N[0], len(SAVED_VARS))
    return float(head[0]), int(head[1]), {v: vals[:, :, k] for k, v in enumerate(SAVED_VARS)}


def write_reference_file(params, grid, dt, cycles, path):
    """ref test/reference_data/reference_functions.jl:37-43"""
    host = grid.device_to_host(SAVED_VARS)
    with open(path, "w") as f:
        f.write("%#.15g, %d\n" % (dt, cycles))
        write_blocks_to_file(params, host, f)


# ---- comparison (ref src/io.jl:113-227) -------------------------------------------------------------------
def compare_host(params, ref, ours, label, vars=SAVED_VARS, verbose=True):
    """True when ``ours`` differs from ``ref`` beyond ``comparison_tolerance`` (relative), real cells only
    unless write_ghosts."""
    sx = params.N[0] + 2 * params.nghost
    (x0, x1), (y0, y1) = _rows(params, params.write_ghosts)
    different = False
    for v in vars:
        a = np.asarray(ref[v]).reshape(-1, sx)[y0:y1, x0:x1]
        b = np.asarray(ours[v]).reshape(-1, sx)[y0:y1, x0:x1]
        mask = ~np.isclose(a, b, rtol=params.comparison_tolerance, atol=0.0, equal_nan=True)
        n = int(mask.sum())
        if n:
            if verbose:
                if not different:
                    print(f"At {label}:")
                print(f"  {n} differences found in {v}")
                for (j, i) in np.argwhere(mask)[:20]:
                    print(f"   - ({i + x0 - params.nghost + 1:3d},{j + y0 - params.nghost + 1:3d}): "
                          f"{a[j, i]:12.5g} ≢ {b[j, i]:12.5g} ({a[j, i] - b[j, i]:12.5g})")
            different = True
    return different


def step_checkpoint(params, grid, step_label, axis_letter):
    """ref src/io.jl:185-227: dump (is_ref) or diff (compare) the state after one sub-step. Returns True when
    a difference was found (the caller stops the cycle, like the reference's ``@checkpoint``)."""
    if not params.compare:
        return False
    gdt = grid.global_dt
    name = f"{params.output_file}_{gdt.cycle:03d}_{step_label}_{axis_letter}"
    if params.is_ref:
        if step_label == "time_step":
            write_time_step_file(params, gdt.current_dt, name)
        else:
            write_sub_domain_file(params, grid, name, no_msg=True)
        return False
    if step_label == "time_step":
        ref_dt = read_time_step_file(params, name)
        different = not np.isclose(ref_dt, gdt.current_dt, rtol=params.comparison_tolerance, atol=0.0)
        if different:
            print(f"Time step difference: ref Δt = {ref_dt:.18f}, Δt = {gdt.current_dt:.18f}, "
                  f"diff = {ref_dt - gdt.current_dt:.18f}")
    else:
        ref = read_sub_domain_file(params, name)
        different = compare_host(params, ref, grid.device_to_host(SAVED_VARS), step_label)
    if params.use_MPI:
        from .halo_exchange import allreduce_sum
        different = allreduce_sum(params, (float(different),))[0] > 0
    if different:
        write_sub_domain_file(params, grid, name + "_diff", no_msg=True)
        print(f"Difference file written to {name}_diff")
    return different
