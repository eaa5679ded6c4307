#!/usr/bin/env python3
"""Extract from the reference's Julia sources the NAMES the binding integration/ArmonHIPNative.jl depends on, as a small
data fixture (tests/golden/julia_names.json) for tests/test_julia_binding.py — run in the build container, where
/root/reference exists (the GPU box has no reference; the test reads the fixture only).

What is extracted (names and counts only, no source text):
 * functions : every function / macro-free method name defined in src/**/*.jl with the sorted set of its positional arities
               (long form `function f(args)` and one-line form `f(args) = ...`);
 * types     : struct / mutable struct / abstract type / @enumx names;
 * fields    : field names of the structs the binding reaches into (ArmonParameters, SolverState, GlobalTimeStep, BlockData,
               LocalTaskBlock, BlockGrid, StepsRanges, BlockSize types);
 * kernels   : for every `@generic_kernel function K(args)`, the positional parameter names of the generated MAIN function
               K(params, [data,] range, rest...) — the kernel's own arguments minus those named like a BlockData field (taken
               from `data`) and those named like an ArmonParameters field (taken from `params`), as
               src/generic_kernel.jl:825-846 builds it (pack_struct_fields, :485-498).

    python tests/golden/make_julia_names.py [/root/reference]
"""
import glob
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "julia_names.json")
IDENT = r"[^\W\d][\w!]*"        # Julia identifier (unicode letters, digits, _, !)


def strip_comments(text):
    text = re.sub(r"#=.*?=#", "", text, flags=re.S)
    out = []
    for line in text.splitlines():
        # a '#' outside a string literal starts a comment (good enough: the sources have no '#' inside strings on
        # definition lines)
        q, cut = False, len(line)
        for i, ch in enumerate(line):
            if ch == '"':
                q = not q
            elif ch == "#" and not q:
                cut = i
                break
        out.append(line[:cut])
    return "\n".join(out)


def matching(s, i):
    depth = 0
    for k in range(i, len(s)):
        if s[k] in "([{":
            depth += 1
        elif s[k] in ")]}":
            depth -= 1
            if depth == 0:
                return k
    raise ValueError("unbalanced")


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [p.strip() for p in out if p.strip()]


def arg_name(arg):
    """name of one positional parameter: `x`, `x::T`, `x=1`, `@Const(x::T)`, `::Type` (None), `(; a, b)::T`..."""
    arg = arg.strip()
    m = re.match(r"@\w+\((.*)\)$", arg)
    if m:
        arg = m.group(1)
    arg = split_top(arg, "=")[0] if "=" in arg and not arg.startswith("(") else arg
    arg = arg.split("::")[0].strip()
    arg = arg.rstrip(".")              # varargs `xs...`
    return arg if re.fullmatch(IDENT, arg) else None


def signature(params):
    """(positional parameter strings, has_varargs) of a parameter list without its parentheses"""
    pos = split_top(params, ";")[0] if params.strip() and not params.strip().startswith(";") else ""
    items = split_top(pos) if pos else []
    return items, any(i.rstrip().endswith("...") for i in items)


files = sorted(glob.glob(os.path.join(REF, "src", "**", "*.jl"), recursive=True))
functions, types, fields, kernels = {}, set(), {}, {}
WANT_FIELDS = {"ArmonParameters", "SolverState", "GlobalTimeStep", "BlockData", "LocalTaskBlock", "BlockGrid", "StepsRanges",
               "StaticBSize", "DynamicBSize", "RemoteTaskBlock", "SolverStats"}

for path in files:
    code = strip_comments(open(path, encoding="utf-8").read())
    # types
    for m in re.finditer(r"^\s*(?:mutable\s+)?struct\s+(" + IDENT + r")", code, flags=re.M):
        types.add(m.group(1))
    for m in re.finditer(r"^\s*abstract\s+type\s+(" + IDENT + r")", code, flags=re.M):
        types.add(m.group(1))
    for m in re.finditer(r"@enumx\s+(" + IDENT + r")", code):
        types.add(m.group(1))
    # struct fields (one-line `struct X end` definitions have none and must not open a block)
    code_blocks = re.sub(r"^(?:mutable\s+)?struct[^\n]*\bend[ \t]*$", "", code, flags=re.M)
    for m in re.finditer(r"^(?:mutable\s+)?struct\s+(" + IDENT + r")[^\n]*\n(.*?)^end", code_blocks, flags=re.M | re.S):
        name, body = m.group(1), m.group(2)
        if name not in WANT_FIELDS:
            continue
        names = []
        header = m.group(0).split("\n", 1)[0]
        depth = header.count("{") - header.count("}")      # a type-parameter list that continues on the next lines
        for line in body.splitlines():
            if depth > 0:
                depth += line.count("{") - line.count("}")
                continue
            if re.match(r"\s*function\b", line) or re.match(r"\s{4}" + IDENT + r"\(", line):
                break                                   # inner constructors follow the fields
            fm = re.match(r"\s{4}(?:const\s+)?(" + IDENT + r")\s*(?:::|$)", line)
            if fm and fm.group(1) not in ("end", "function"):
                names.append(fm.group(1))
        fields[name] = names
    # long-form functions (also behind @generic_kernel / @kernel_function / @inline / @fast ...)
    for m in re.finditer(r"(@generic_kernel\s+)?function\s+((?:" + IDENT + r"\.)*" + IDENT + r")\s*(\{[^}]*\})?\(", code):
        name = m.group(2).split(".")[-1]
        start = m.end() - 1
        params = code[start + 1:matching(code, start)]
        items, varargs = signature(params)
        functions.setdefault(name, set()).add(("%d+" % (len(items) - 1)) if varargs else str(len(items)))
        if m.group(1):
            kernels.setdefault(name, []).append([arg_name(a) for a in items])
    # one-line methods at top level: `name(args) = ...` / `name(args) where {T} = ...`
    for m in re.finditer(r"^(?:@" + IDENT + r"\s+)*((?:" + IDENT + r"\.)*" + IDENT + r")\(", code, flags=re.M):
        start = m.end() - 1
        try:
            end = matching(code, start)
        except ValueError:
            continue
        rest = code[end + 1:end + 200]
        if not re.match(r"\s*(?:::[^=\n]+?)?\s*(?:where\s+[^=\n]+?)?\s*=(?!=)", rest):
            continue
        name = m.group(1).split(".")[-1]
        items, varargs = signature(code[start + 1:end])
        functions.setdefault(name, set()).add(("%d+" % (len(items) - 1)) if varargs else str(len(items)))

block_fields = set(fields["BlockData"])
param_fields = set(fields["ArmonParameters"])
kernel_main = {}
for name, defs in kernels.items():
    sigs = []
    for args in defs:
        rest = [a for a in args if a not in block_fields]
        has_data = len(rest) != len(args)
        rest = [a for a in rest if a not in param_fields]
        sigs.append(["params"] + (["data"] if has_data else []) + ["range"] + rest)
    kernel_main[name] = sigs

fixture = {
    "source": "Keluaa/Armon.jl as vendored under /root/reference (names only; made by tests/golden/make_julia_names.py)",
    "functions": {k: sorted(v) for k, v in sorted(functions.items())},
    "types": sorted(types),
    "fields": {k: fields[k] for k in sorted(fields)},
    "kernel_main_signatures": {k: kernel_main[k] for k in sorted(kernel_main)},
}
with open(OUT, "w", encoding="utf-8") as f:
    json.dump(fixture, f, ensure_ascii=False, indent=1, sort_keys=True)
print(f"{OUT}: {len(functions)} functions, {len(types)} types, fields of {sorted(fields)}, {len(kernel_main)} kernels")
