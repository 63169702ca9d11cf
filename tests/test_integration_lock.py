"""INTEGRATION.md's cgo shim cannot be compiled here (no Go toolchain): this keeps its text in lock-step with the headers.
Every `C.gsdf_*` call in the Go blocks must name a function include/gsdf_hip.h declares, with that declaration's number of
arguments; every `C.gsdf_*` type must be a type of the headers; every `C.GSDF_*` constant one of their enumerators or macros;
and every field the Go code reads or writes through a C struct value must be a member of that struct."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _headers():
    return _strip_comments("\n".join(open(os.path.join(ROOT, "include", h)).read() for h in ("gsdf_program.h", "gsdf_hip.h")))


def _split_args(s):
    """Top-level comma split of an argument list (parentheses, brackets and braces nest)."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _call_args(text, at):
    """text[at] is the '(' of a call: returns the argument list text."""
    depth, i = 0, at
    while True:
        depth += (text[i] == "(") - (text[i] == ")")
        if depth == 0:
            return text[at + 1:i]
        i += 1


def header_functions(h):
    """name -> number of parameters, for every function the headers declare."""
    out = {}
    for m in re.finditer(r"\b(gsdf_hip_[a-z0-9_]+)\s*\(", h):
        args = _call_args(h, m.end() - 1).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    return out


def header_types(h):
    t = set(re.findall(r"\btypedef\s+struct\s+(gsdf_[a-z0-9_]+)\s+\1\s*;", h))
    t |= set(re.findall(r"\}\s*(gsdf_[a-z0-9_]+)\s*;", h))
    t |= set(re.findall(r"\bstruct\s+(gsdf_[a-z0-9_]+)\b", h))
    return t


def header_constants(h):
    c = set(re.findall(r"#\s*define\s+(GSDF_[A-Z0-9_]+)", h))
    for body in re.findall(r"\benum\b[^{;]*\{(.*?)\}", h, flags=re.S):
        c |= set(re.findall(r"\b(GSDF_[A-Z0-9_]+)\b", body))
    return c


def struct_fields(h, name):
    m = re.search(r"struct\s+%s\s*\{(.*?)\}" % name, h, flags=re.S) or re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*%s\s*;" % name, h, flags=re.S)
    assert m, name
    f = set()
    for decl in m.group(1).split(";"):
        for part in decl.split(","):
            w = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*$", part.strip())
            if w:
                f.add(w[0])
    return f


def go_blocks():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```go\n(.*?)```", md, flags=re.S)
    assert len(blocks) >= 8
    # Go comments and the cgo preamble (C code inside /* */ in front of import "C") are not Go text
    return [re.sub(r"//[^\n]*", " ", re.sub(r"/\*.*?\*/", " ", b, flags=re.S)) for b in blocks]


def test_every_c_reference_of_the_go_shim_exists_in_the_headers():
    h = _headers()
    funcs, types, consts = header_functions(h), header_types(h), header_constants(h)
    assert len(funcs) >= 60 and "gsdf_hip_eval3" in funcs and "gsdf_mesh_opts" in types and "GSDF_ERR_EMPTY_BUFFERS" in consts
    seen_calls, seen_types, seen_consts = set(), set(), set()
    for go in go_blocks():
        for m in re.finditer(r"\bC\.(gsdf_[a-z0-9_]+)\b", go):
            name = m.group(1)
            rest = go[m.end():]
            if name.startswith("gsdf_hip_"):
                assert name in funcs, f"INTEGRATION.md calls C.{name}: not declared in include/gsdf_hip.h"
                if rest.lstrip().startswith("("):
                    args = _call_args(go, m.end() + (len(rest) - len(rest.lstrip())))
                    n = 0 if not args.strip() else len(_split_args(args))
                    assert n == funcs[name], f"C.{name}: {n} arguments in INTEGRATION.md, {funcs[name]} parameters in the header"
                    seen_calls.add(name)
            else:
                assert name in types, f"INTEGRATION.md uses the type C.{name}: not in the headers"
                seen_types.add(name)
        for m in re.finditer(r"\bC\.(GSDF_[A-Z0-9_]+)\b", go):
            assert m.group(1) in consts, f"INTEGRATION.md uses C.{m.group(1)}: not an enumerator or macro of the headers"
            seen_consts.add(m.group(1))
    # the shim's core calls are all there (a block deleted from the document would otherwise pass vacuously)
    for need in ("gsdf_hip_init", "gsdf_hip_program_create", "gsdf_hip_program_destroy", "gsdf_hip_eval3", "gsdf_hip_evaluations", "gsdf_hip_last_error",
                 "gsdf_hip_mesh_octree", "gsdf_hip_mesh_read", "gsdf_hip_mesh_destroy", "gsdf_hip_mesh_stats_get", "gsdf_hip_mesh_gatherv"):
        assert need in seen_calls, need
    assert {"gsdf_tree", "gsdf_node", "gsdf_program", "gsdf_mesh", "gsdf_mesh_opts", "gsdf_mesh_stats"} <= seen_types
    assert {"GSDF_ERR_EMPTY_BUFFERS", "GSDF_ERR_LENGTH_MISMATCH"} <= seen_consts


def test_struct_members_the_go_shim_touches_exist():
    """`var st C.gsdf_mesh_stats ... st.evals`, `opts := C.gsdf_mesh_opts{...}; opts.field = ...`, `C.gsdf_node{op: ...}`."""
    h = _headers()
    checked = 0
    for go in go_blocks():
        # variables declared with a C struct type in this block
        var_type = {}
        for m in re.finditer(r"\bvar\s+([A-Za-z_][A-Za-z0-9_]*)\s+C\.(gsdf_[a-z0-9_]+)\b", go):
            var_type[m.group(1)] = m.group(2)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*:?=\s*C\.(gsdf_[a-z0-9_]+)\s*\{", go):
            var_type[m.group(1)] = m.group(2)
        for v, t in var_type.items():
            if not re.search(r"struct\s+%s\s*\{" % t, h) and not re.search(r"\}\s*%s\s*;" % t, h):
                continue  # opaque handle
            fields = struct_fields(h, t)
            for m in re.finditer(r"\b%s\.([A-Za-z_][A-Za-z0-9_]*)\b" % re.escape(v), go):
                assert m.group(1) in fields, f"INTEGRATION.md: {v}.{m.group(1)} -- {t} has no such member ({sorted(fields)})"
                checked += 1
        # composite literals: C.gsdf_x{field: value, ...}
        for m in re.finditer(r"\bC\.(gsdf_[a-z0-9_]+)\s*\{", go):
            t = m.group(1)
            if not re.search(r"struct\s+%s\s*\{" % t, h):
                continue
            depth, i = 0, m.end() - 1
            while True:
                depth += (go[i] == "{") - (go[i] == "}")
                if depth == 0:
                    break
                i += 1
            fields = struct_fields(h, t)
            for part in _split_args(go[m.end():i]):
                k = re.match(r"\s*([A-Za-z_][A-Za-z0-9_]*)\s*:", part)
                if k:
                    assert k.group(1) in fields, f"INTEGRATION.md: C.{t}{{{k.group(1)}: ...}} -- no such member"
                    checked += 1
    assert checked >= 10, checked


def test_status_codes_are_what_the_shim_assumes():
    """The shim maps two status codes onto the reference's two error values (gleval/gpu.go:83-87) and treats 0 as success."""
    h = _headers()
    m = dict(re.findall(r"\b(GSDF_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", h))
    assert m.get("GSDF_OK") == "0" and "GSDF_ERR_EMPTY_BUFFERS" in m and "GSDF_ERR_LENGTH_MISMATCH" in m
    assert len(set(m.values())) == len(m)   # distinct
