"""The reference-side binding (integration/ArmonHIPNative.jl) cannot be executed here (no `julia`), so what can be
checked statically is: every `ccall(fn(:armon_hip_…[, T]), ret, (argtypes…), args…)` in it is parsed and compared with
the C ABI as bound by armon.jl_amd/_lib.py (SIGNATURES = include/armon_hip.h, itself checked against the library's
exports by test_host_logic.py) — symbol exists, arity, the C class and width of every argument and of the return type,
number of values actually passed — for the fp64 symbol and, where the call is generic in T, the `_f32` one; and the
Julia mirrors of the C structs (CRange, CBlockData, SweepDesc, HaloDesc) against the ctypes structures field by field.
"""
import ctypes as C
import os
import re

import pytest

from armon_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = open(os.path.join(ROOT, "integration", "ArmonHIPNative.jl"), encoding="utf-8").read()
CODE = "\n".join(line.split("#", 1)[0] if not line.lstrip().startswith('"') else line for line in JL.splitlines())


def split_top(s):
    """Split on top-level commas (parentheses, braces and brackets nest)."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def matching(s, i):
    """Index of the parenthesis closing the one at s[i]."""
    depth = 0
    for k in range(i, len(s)):
        if s[k] in "({[":
            depth += 1
        elif s[k] in ")}]":
            depth -= 1
            if depth == 0:
                return k
    raise ValueError("unbalanced")


def ccalls():
    found = []
    for m in re.finditer(r"ccall\(\s*fn\(:(armon_hip_\w+)(\s*,\s*T)?\)", CODE):
        start = CODE.index("(", m.start())
        body = CODE[start + 1:matching(CODE, start)]
        parts = split_top(body)
        ret, argtypes, args = parts[1], parts[2], parts[3:]
        assert argtypes.startswith("(") and argtypes.endswith(")"), (m.group(1), argtypes)
        types = split_top(argtypes[1:-1])
        found.append((m.group(1), bool(m.group(2)), ret, types, args))
    return found


def jl_kind(t, real_width):
    t = t.strip()
    if t in ("Cint", "Int32"):
        return ("int", 4)
    if t == "Int64":
        return ("int", 8)
    if t == "Csize_t":
        return ("int", 8)
    if t == "Float64":
        return ("float", 8)
    if t == "T":
        return ("float", real_width)
    if t == "CRange":
        return ("struct", C.sizeof(_lib.Range))
    if t == "SweepDesc{T}":
        return ("struct", C.sizeof(_lib.SweepDesc))
    if t == "Cstring" or t.startswith("Ptr{") or t.startswith("Ref{"):
        return ("ptr", 8)
    raise AssertionError(f"unmapped Julia type {t!r}")


def c_kind(t):
    if t is None:
        return ("void", 0)
    if t in (C.c_int, C.c_int32):
        return ("int", 4)
    if t in (C.c_int64, C.c_size_t):
        return ("int", 8)
    if t is C.c_double:
        return ("float", 8)
    if t is C.c_float:
        return ("float", 4)
    if t in (C.c_void_p, C.c_char_p) or isinstance(t, type(C.POINTER(C.c_int))) and issubclass(t, C._Pointer):
        return ("ptr", 8)
    if isinstance(t, type) and issubclass(t, C.Structure):
        return ("struct", C.sizeof(t))
    raise AssertionError(f"unmapped ctypes type {t!r}")


CALLS = ccalls()


def test_the_binding_calls_the_abi_it_claims():
    names = {c[0] for c in CALLS}
    assert len(CALLS) >= 35
    # the 1:1 kernels, the hooks, the fused sweep and the multi-GPU entry points are all bound
    for must in ("perfect_gas_EOS", "bizarrium_EOS", "acoustic", "acoustic_GAD", "cell_update", "advection_first_order",
                 "advection_second_order", "euler_projection", "boundary_conditions", "pack_to_array", "unpack_from_array",
                 "dtCFL", "conservation_vars", "init_test", "sweep", "tune_placement", "init", "sync", "malloc", "free",
                 "memcpy", "memcpy_async", "malloc_host", "event_record", "event_sync", "device_memory_info",
                 "mgpu_init", "mgpu_init_rank", "mgpu_unique_id", "halo_exchange_start", "halo_exchange_finish",
                 "dt_allreduce", "mgpu_cycle"):
        assert "armon_hip_" + must in names, must


@pytest.mark.parametrize("call", CALLS, ids=[f"{c[0]}{'<T>' if c[1] else ''}#{i}" for i, c in enumerate(CALLS)])
def test_ccall_matches_the_header(call):
    name, generic, ret, types, args = call
    variants = [(name, 8)] + ([(name + "_f32", 4)] if generic else [])
    for sym, width in variants:
        assert sym in _lib.SIGNATURES, f"{sym} is not declared in include/armon_hip.h"
        res, argtypes = _lib.SIGNATURES[sym]
        assert len(types) == len(argtypes), f"{sym}: {len(types)} argument types in the ccall, {len(argtypes)} in the header"
        assert len(args) == len(types), f"{sym}: {len(args)} values passed for {len(types)} declared types"
        assert jl_kind(ret, width) == c_kind(res), f"{sym}: return type {ret}"
        for k, (jt, ct) in enumerate(zip(types, argtypes)):
            assert jl_kind(jt, width) == c_kind(ct), f"{sym}: argument {k} is {jt} in the ccall, {ct} in the header"


def jl_struct(name):
    m = re.search(r"struct\s+" + name + r"(\{T\})?\s*[;\n](.*?)\bend\b", CODE, re.S)
    assert m, name
    fields = []
    for f in re.split(r"[;\n]", m.group(2)):
        f = f.strip()
        if "::" in f:
            n, t = f.split("::", 1)
            fields.append((n.strip(), t.strip()))
    return fields


def field_kind(t, width):
    m = re.fullmatch(r"NTuple\{(\d+),\s*(.+)\}", t)
    if m:
        return ("array", int(m.group(1))) + jl_kind(m.group(2), width)
    return jl_kind(t, width)


def ct_field_kind(t):
    if isinstance(t, type) and issubclass(t, C.Array):
        return ("array", t._length_) + c_kind(t._type_)
    return c_kind(t)


@pytest.mark.parametrize("jl,ct,same_names", [("CRange", _lib.Range, True), ("SweepDesc", _lib.SweepDesc, True),
                                              ("HaloDesc", _lib.HaloDesc, True), ("CBlockData", _lib.BlockDataPtrs, False),
                                              ("CyclePlan", _lib.CyclePlan, True), ("TileCycle", _lib.TileCycle, True)])
def test_struct_mirrors(jl, ct, same_names):
    fields = jl_struct(jl)
    assert len(fields) == len(ct._fields_), (jl, len(fields), len(ct._fields_))
    for (jn, jt), (cn, ctt) in zip(fields, ct._fields_):
        if same_names:
            assert jn == cn, (jl, jn, cn)
        assert field_kind(jt, 8) == ct_field_kind(ctt), (jl, jn, jt, ctt)


def test_device_array_type_and_fused_override_are_what_the_reference_needs():
    # ref src/blocking/block_grid.jl:52-53 applies {T, 1} to device_array_type(dev): element type AND rank parameters
    assert re.search(r"mutable struct HIPVector\{T,\s*N\}\s*<:\s*AbstractArray\{T,\s*N\}", CODE)
    assert re.search(r"HIPVector\{T,\s*N\}\(::UndefInitializer,\s*dims::NTuple\{N,\s*Integer\}", CODE)
    assert "device_array_type(::HIPNative) = HIPVector" in CODE
    # the fused path is bound as a solver_cycle method on this device (precedent: ext/ArmonKokkos.jl:212-258)
    assert re.search(r"function solver_cycle\(p::HP\{T\},\s*grid::BlockGrid\)", CODE)
    body = CODE[CODE.index("function solver_cycle(p::HP{T}"):]
    for needle in ("fused_sweep!", "contribute_to_dt!", "emit_p = last && will_end", "armon_hip_memcpy_async",
                   "armon_hip_event_sync", "invoke(solver_cycle"):
        assert needle in body, needle
    assert "v.ptr, a.ptr = a.ptr, v.ptr" in CODE        # ping-pong swap of the allocations behind BlockData's vectors


def _strip_julia(src):
    """Julia source without comments, string / char literals and docstrings (contents replaced by blanks)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith('"""', i):
            j = src.index('"""', i + 3) + 3
            out.append(" " * (j - i))
            i = j
        elif c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""' + " " * (j - i - 1))
            i = j + 1
        elif c == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src[i + 3] == "'")):
            j = i + (3 if src[i + 2] == "'" else 4)
            out.append(" " * (j - i))
            i = j
        elif c == "#":
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        else:
            out.append(c)
            i += 1
    return "".join(out)


def test_julia_file_is_structurally_sound():
    """The file cannot be executed here: at least every block opener has its `end`, brackets pair up, and the functions
    the overrides call exist under the names they are called by."""
    src = _strip_julia(JL)
    # brackets
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for k, ch in enumerate(src):
        if ch in "([{":
            stack.append((ch, k))
        elif ch in ")]}":
            assert stack and stack[-1][0] == pairs[ch], f"unbalanced {ch!r} at line {src.count(chr(10), 0, k) + 1}"
            stack.pop()
    assert not stack, f"unclosed {stack[-1][0]!r} at line {src.count(chr(10), 0, stack[-1][1]) + 1}"
    # blocks: openers at the start of a statement or after `=`/`(`/`,` (begin, do, let, try), `end` as a word outside []
    depth_sq, opens, ends = 0, 0, 0
    tokens = re.finditer(r"[\[\]]|\b(function|if|for|while|begin|do|struct|module|let|try|macro|quote|end)\b", src)
    prev_word_at = {}
    for m in tokens:
        t = m.group(0)
        if t == "[":
            depth_sq += 1
        elif t == "]":
            depth_sq -= 1
        elif t == "end":
            if depth_sq == 0:
                ends += 1
        elif t == "struct":
            opens += 1          # `mutable struct` is one opener (mutable is not in the list)
        elif t in ("if", "for", "while"):
            # a block opener only at the start of a statement; inside (...) or [...] it is a generator / ternary-free filter
            line_start = src.rfind("\n", 0, m.start()) + 1
            before = src[line_start:m.start()]
            if before.strip() == "" or before.rstrip().endswith(("=", "&&", "||", "begin")):
                opens += 1
        else:
            opens += 1
    assert opens == ends, f"{opens} block openers but {ends} `end`s"
    # names used by the overrides are defined in the file
    defined = set(re.findall(r"^\s*(?:function\s+)?([A-Za-z_][\w!]*)\s*(?:\{[^}]*\})?\(", src, re.M))
    for name in ("fused_sweep!", "fused_sweep_mpi!", "swap_state!", "rank_group", "sweep_lag", "fused_state", "unique_id",
                 "halo_exchange_start!", "halo_exchange_finish!", "halo_exchange_finish_edge!", "edge_context", "edge_dt",
                 "edge_join!", "dt_allreduce!", "TileGroup", "HaloDesc", "check", "fn", "sweep_desc", "native_cycle!"):
        assert name in defined, f"{name} is called but never defined"


# ---- names the binding takes from the reference (fixture made from /root/reference by tests/golden/make_julia_names.py) ----
import json

NAMES = json.load(open(os.path.join(ROOT, "tests", "golden", "julia_names.json"), encoding="utf-8"))
SRC = _strip_julia(JL)
IDENT = r"[^\W\d][\w!]*"


def _known(name):
    return name in NAMES["functions"] or name in NAMES["types"]


def test_every_imported_name_exists_in_the_reference():
    """`import Armon: a, b, …` fails at load time for a name Armon does not define: every imported name, and every
    qualified `Armon.X` the file uses, must be a function, type or enum module of the reference's sources."""
    imported = []
    for m in re.finditer(r"^import Armon:(.*(?:\n[ \t]+.*)*)", SRC, re.M):
        imported += [n.strip() for n in m.group(1).replace("\n", " ").split(",") if n.strip()]
    assert len(imported) >= 40
    missing = [n for n in imported if not _known(n)]
    assert not missing, f"imported from Armon but not defined there: {missing}"
    qualified = sorted(set(re.findall(r"\bArmon\.(" + IDENT + r")", SRC)))
    assert len(qualified) >= 15
    missing = [n for n in qualified if not _known(n)]
    assert not missing, f"Armon.X used but not defined in the reference: {missing}"


# which reference struct a variable name of the binding stands for (by the file's own conventions)
VAR_TYPES = {"p": "ArmonParameters", "params": "ArmonParameters", "state": "SolverState", "gdt": "GlobalTimeStep",
             "d": "BlockData", "blk": "LocalTaskBlock", "grid": "BlockGrid"}
SECOND_LEVEL = {("state", "steps_ranges"): "StepsRanges", ("blk", "state"): "SolverState"}


def test_every_field_the_binding_reads_exists_in_the_reference_struct():
    """state.riemann_scheme, p.cart_coords, gdt.current_dt, d.uˢ, blk.size …: a misspelt field is a runtime error in Julia."""
    seen = 0
    for m in re.finditer(r"(?<![\w.])(" + "|".join(VAR_TYPES) + r")\.(" + IDENT + r")(?:\.(" + IDENT + r"))?", SRC):
        var, field, sub = m.groups()
        line = SRC.count("\n", 0, m.start()) + 1
        assert field in NAMES["fields"][VAR_TYPES[var]], f"line {line}: {var}.{field} — {VAR_TYPES[var]} has no such field"
        seen += 1
        if sub and (var, field) in SECOND_LEVEL:
            t = SECOND_LEVEL[(var, field)]
            assert sub in NAMES["fields"][t], f"line {line}: {var}.{field}.{sub} — {t} has no such field"
    assert seen >= 60
    # fields written by the binding must be fields too (p.backend_options = …, state.dt = …)
    for var, field in re.findall(r"(?<![\w.])(params|state|gdt)\.(" + IDENT + r")\s*=(?!=)", SRC):
        assert field in NAMES["fields"][VAR_TYPES[var]], (var, field)


def _method_defs(name):
    """positional parameter lists of the methods `name(...)` the binding defines (long and one-line form)"""
    out = []
    for m in re.finditer(r"(?:^|\n)\s*(?:function\s+)?(?:Armon\.)?" + re.escape(name) + r"\(", SRC):
        start = m.end() - 1
        params = SRC[start + 1:matching(SRC, start)]
        rest = SRC[matching(SRC, start) + 1:matching(SRC, start) + 120]
        if not (m.group(0).lstrip().startswith("function") or re.match(r"\s*(?:where\s+[^=\n]+?)?\s*=(?!=)", rest)):
            continue                                     # a call, not a definition
        pos = split_top_sep(params, ";")[0] if params.strip() else ""
        out.append([a for a in split_top(pos)] if pos.strip() else [])
    return out


def split_top_sep(s, sep):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


@pytest.mark.parametrize("kernel", ["perfect_gas_EOS!", "bizarrium_EOS!", "acoustic!", "acoustic_GAD!", "cell_update!",
                                    "advection_first_order!", "advection_second_order!", "euler_projection!",
                                    "boundary_conditions!", "pack_to_array!", "unpack_from_array!", "init_test"])
def test_kernel_overrides_have_the_generated_main_signature(kernel):
    """Each kernel method of the binding replaces the MAIN function `@generic_kernel` generates —
    K(params, [data,] range, rest...; kwargs), ref src/generic_kernel.jl:825-846 — so it must take exactly those positional
    parameters (the reference's call sites pass them positionally) and accept the keyword arguments (`; kw...`)."""
    sigs = NAMES["kernel_main_signatures"][kernel]
    defs = _method_defs(kernel)
    assert len(defs) == 1, (kernel, defs)
    mine = defs[0]
    want = sigs[0]
    assert len(mine) == len(want), f"{kernel}: binding takes {mine}, the generated main function takes {want}"
    assert re.match(r"p::HP\{T\}", mine[0].strip())
    if "data" in want:
        assert re.match(r"d::BlockData", mine[1].strip()), mine
    assert re.search(r"::DomainRange", mine[want.index("range")]), mine
    # the remaining parameters keep the reference's order: compare the names where the binding keeps them
    rename = {"test_case": "test", "lim": "lim"}
    for a, b in zip(mine[want.index("range") + 1:], want[want.index("range") + 1:]):
        a = a.split("::")[0].strip()
        assert a in (rename.get(b, b), b.rstrip("_")) or a.replace("advection_", "a") == b.replace("advection_", "a"), (kernel, a, b)
    # keyword arguments of the generated function (no_threading, …) must be accepted
    src = SRC[SRC.index(kernel + "(p::HP{T}"):]
    assert re.match(re.escape(kernel) + r"\([^;]*;\s*kw\.\.\.\)", src.replace("\n", " ")), kernel


@pytest.mark.parametrize("fname,arity", [("create_device", 1), ("init_backend", 2), ("device_array_type", 1),
                                         ("host_array_type", 1), ("device_memory_info", 1), ("print_device_info", 3),
                                         ("dtCFL_kernel", 4), ("conservation_vars", 2), ("solver_cycle", 2)])
def test_hook_overrides_match_a_reference_method(fname, arity):
    """The backend hooks and whole-step overrides extend functions of the reference: same name, and a positional arity the
    reference itself defines for that function (ref src/parameters.jl:751-802,921-951; src/reductions.jl:65,271;
    src/solver.jl:288)."""
    assert fname in NAMES["functions"], fname
    defs = _method_defs(fname)
    assert defs, f"{fname} is not defined by the binding"
    for d in defs:
        assert len(d) == arity, (fname, d)
    assert str(arity) in NAMES["functions"][fname], (fname, arity, NAMES["functions"][fname])
