"""The Go binding (integration/go/**.go: cgo shim, pure-Go tree types, one AppendHIPNodes per node type, renderers, the
RenderShader3D switch) cannot be compiled here (no Go toolchain): this keeps it in lock-step with the headers AND with the
reference's node structs. Every `C.gsdf_*` call must name a function include/gsdf_hip.h declares, with that declaration's number
of arguments; every `C.gsdf_*` type must be a type of the headers; every `C.GSDF_*` constant one of their enumerators or macros;
every field read or written through a C struct value must be a member of that struct; the pure-Go HIPOp constants and HIPNode
must be enum gsdf_op and gsdf_node; every shader node struct of the reference (tests/golden/reference_node_structs.json, checked
against /root/reference where that exists) must have an AppendHIPNodes in its own package that names only fields the struct has,
leaves no parameter field out, and uses the op constant whose header comment lists those fields."""
import glob
import json
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


GO_DIR = os.path.join(ROOT, "integration", "go")


def go_files():
    fs = sorted(glob.glob(os.path.join(GO_DIR, "**", "*.go"), recursive=True))
    assert len(fs) >= 8, fs
    return fs


def _go_code(src):
    """Go text without comments, the cgo preamble (C code inside /* */ in front of import "C"), string and rune literals."""
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', src)
    return re.sub(r"`[^`]*`", '""', src)


def go_blocks():
    """The cgo files (and whatever Go blocks INTEGRATION.md still quotes), comment-free."""
    out = [_go_code(open(f).read()) for f in go_files()]
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out += [_go_code(b) for b in re.findall(r"```go\n(.*?)```", md, flags=re.S)]
    return out


def test_go_files_are_well_formed_and_tagged():
    """What can be checked of Go syntax without a compiler: a package clause, balanced brackets outside comments and strings, build
    tags on every cgo file (and its no-cgo twin), `import "C"` directly behind the preamble."""
    for f in go_files():
        src = open(f).read()
        code = _go_code(src)
        rel = os.path.relpath(f, GO_DIR)
        assert re.search(r"^package [a-z]+$", code, re.M), rel
        assert re.search(r"^package (\w+)$", code, re.M).group(1) == {"gsdf": "gsdf", "threads": "threads"}.get(os.path.basename(os.path.dirname(f)), os.path.basename(os.path.dirname(f))), rel
        for a, b in ("()", "[]", "{}"):
            assert code.count(a) == code.count(b), (rel, a, code.count(a), code.count(b))
        depth = 0
        for ch in code:
            depth += (ch == "{") - (ch == "}")
            assert depth >= 0, rel
        # Go refuses unused imports: every imported package is named somewhere in the code
        for imp in re.findall(r'^\s*"([\w./-]+)"\s*$', re.sub(r"/\*.*?\*/", " ", src, flags=re.S), re.M):
            pkg = imp.split("/")[-1]
            assert re.search(r"\b%s\." % re.escape(pkg), code), f"{rel}: import {imp} is never used"
        if 'import "C"' in src:
            assert src.startswith("//go:build cgo && hip\n\npackage "), rel
            assert re.search(r"\*/\nimport \"C\"\n", src), rel + ': import "C" must follow the preamble comment directly'
            assert '#include "gsdf_hip.h"' in src, rel
        else:
            assert "C." not in re.sub(r"[A-Za-z0-9_]C\.", "", code) or not re.search(r"\bC\.[a-zA-Z_]", code), rel + " uses C.* without importing it"
    tags = {os.path.relpath(f, GO_DIR): open(f).read().split("\n")[0] for f in go_files()}
    assert tags["gleval/gpu_hip.go"] == "//go:build cgo && hip" and tags["gleval/gpu_hip_nocgo.go"] == "//go:build !cgo || !hip"
    # the pure-Go pieces carry no tag: the flattener must build (and be testable) without cgo
    for f in ("gleval/hiptree.go", "gsdf/hip_flatten.go", "forge/threads/hip_flatten.go", "glbuild/hip_unwrap.go"):
        assert not tags[f].startswith("//go:build"), f


def _enum_gsdf_op():
    h = _headers()
    body = re.search(r"enum\s+gsdf_op\s*\{(.*?)\}", h, flags=re.S).group(1)
    return re.findall(r"\b(GSDF_[A-Z0-9_]+)\b", body)


def test_pure_go_op_constants_and_node_layout_are_the_headers():
    src = open(os.path.join(GO_DIR, "gleval", "hiptree.go")).read()
    block = re.search(r"const \(\n\tHIPOpInvalid HIPOp = iota\n(.*?)\n\)", src, flags=re.S).group(1)
    names = ["HIPOpInvalid"] + [m.group(1) for m in re.finditer(r"^\t(HIP[A-Za-z0-9]+)\b", block, re.M)]
    enum = _enum_gsdf_op()
    assert len(names) == len(enum) == 55   # 53 node kinds + invalid + count (the header pins GSDF_OP_COUNT == 54)
    for i, (g, c) in enumerate(zip(names, enum)):
        assert g[3:].upper() == c[5:].replace("_", ""), (i, g, c)
    # the parameter comments of the Go constants are the header's (p0=..., aux = ...): same field <-> slot table on both sides
    hraw = open(os.path.join(ROOT, "include", "gsdf_program.h")).read()
    hcom = dict(re.findall(r"\b(GSDF_[A-Z0-9_]+),\s*/\*\s*(.*?)\s*\*/", hraw))
    gcom = dict(re.findall(r"^\t(HIP[A-Za-z0-9]+)\s*//\s*(.*?)\s*$", block, re.M))
    slots = lambda t: sorted(set(re.findall(r"p\d(?:\.\.\d|,\d)?|\baux\b", t)))
    for g, c in zip(names, enum):
        if g in gcom or slots(hcom.get(c, "")):
            assert slots(gcom.get(g, "")) == slots(hcom.get(c, "")), (g, gcom.get(g), hcom.get(c))
    # HIPNode == gsdf_node, field for field
    node = re.search(r"type HIPNode struct \{(.*?)\n\}", src, flags=re.S).group(1)
    gf = re.findall(r"^\t(\w+)\s+(\S+)", re.sub(r"//[^\n]*", "", node), re.M)
    assert gf == [("Op", "HIPOp"), ("NChild", "uint16"), ("LinkOff", "uint32"), ("AuxOff", "uint32"), ("AuxLen", "uint32"), ("P", "[HIPNodeNParam]float32")]
    assert "type HIPOp uint16" in src and "const HIPNodeNParam = 8" in src
    h = _headers()
    cn = re.search(r"struct\s+gsdf_node\s*\{(.*?)\}", h, flags=re.S).group(1)
    assert re.findall(r"(\w+)\s+(\w+)(?:\[\w+\])?\s*;", cn) == [("uint16_t", "op"), ("uint16_t", "nchild"), ("uint32_t", "link_off"), ("uint32_t", "aux_off"), ("uint32_t", "aux_len"), ("float", "p")]
    assert re.search(r"#\s*define\s+GSDF_NODE_NPARAM\s+8\b", h)
    # and the cgo file refuses to build if the two ever differ in size or count
    cg = open(os.path.join(GO_DIR, "gleval", "gpu_hip.go")).read()
    assert "unsafe.Sizeof(C.gsdf_node{}) - unsafe.Sizeof(HIPNode{})" in cg and "unsafe.Sizeof(HIPNode{}) - unsafe.Sizeof(C.gsdf_node{})" in cg
    assert "uint(C.GSDF_OP_COUNT) - uint(HIPOpCount)" in cg and "uint(HIPOpCount) - uint(C.GSDF_OP_COUNT)" in cg


# parameter fields of a node struct that the evaluators never read (so the flattener must NOT need them), with the reason
UNUSED_FIELDS = {
    ("transform", "t"): "the CPU evaluator multiplies by tInv (cpu_evaluators.go:495-497); t only feeds Bounds and GLSL",
    ("transform", "hash"): "GLSL name suffix",
    ("rotation2D", "t"): "the evaluator uses tInv (cpu_evaluators.go:1186-1203)",
    ("lines2D", "hash"): "GLSL name suffix",
    ("lines2Dssbo", "bufname"): "GLSL buffer name",
    ("polySSBO", "bufname"): "GLSL buffer name",
    ("translateMulti2D", "bufname"): "GLSL buffer name",
}


def _node_structs():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_node_structs.json")))


def test_node_struct_fixture_is_the_references():
    """Build container only: the committed declarations are what /root/reference declares today."""
    import importlib.util
    import pytest
    if not os.path.isdir("/root/reference"):
        pytest.skip("no reference tree on this box")
    spec = importlib.util.spec_from_file_location("make_node_structs", os.path.join(ROOT, "tests", "golden", "make_node_structs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert json.loads(json.dumps(mod.structs())) == _node_structs()
    # ... and the seam the shim replaces still looks as the shim assumes (gleval/gpu.go:35-103, gleval/gleval.go:47-48, glbuild.go:1366)
    gpu = open("/root/reference/gleval/gpu.go").read()
    assert "func NewComputeGPUSDF3(" in gpu and "func (sdf *SDF3Compute) Evaluate(pos []ms3.Vec, dist []float32, userData any) error" in gpu and "Evaluations() uint64" in gpu
    gl = open("/root/reference/gleval/gleval.go").read()
    assert "errEmptyBuffers" in gl and "errMismatchBufferLength" in gl
    assert "func unwraproot(s Shader) Shader" in open("/root/reference/glbuild/glbuild.go").read()
    aux = open("/root/reference/gsdfaux/gsdfaux.go").read()
    for name in ("stopwatch()", "percentUint64(", "renderWithFlatMC", "cfg.STLOutput", "cfg.Resolution"):
        assert name in aux, name


def test_every_reference_node_has_a_flattener_that_copies_its_fields():
    structs = _node_structs()
    nodes = {k: v for k, v in structs.items() if v["node"]}
    assert len(nodes) == 55
    code = {"gsdf": _go_code(open(os.path.join(GO_DIR, "gsdf", "hip_flatten.go")).read()),
            "threads": _go_code(open(os.path.join(GO_DIR, "forge", "threads", "hip_flatten.go")).read())}
    enum = _enum_gsdf_op()
    hraw = open(os.path.join(ROOT, "include", "gsdf_program.h")).read()
    hcom = dict(re.findall(r"\b(GSDF_[A-Z0-9_]+),\s*/\*\s*(.*?)\s*\*/", hraw))
    go_ops = {c[5:].replace("_", ""): c for c in enum}
    methods = {}
    for pkg, src in code.items():
        for m in re.finditer(r"^func \((\w+) \*(\w+)\) AppendHIPNodes\(f \*gleval\.HIPFlattener\) \(uint32, error\) \{\n(.*?)^\}", src, re.S | re.M):
            assert m.group(2) not in methods, m.group(2)
            methods[m.group(2)] = (pkg, m.group(1), m.group(3))
    used_ops = {}
    for name, st in nodes.items():
        own = name in methods
        if not own:
            # only the two SSBO variants may rely on a promoted method (they embed the node they are evaluated as, primitives2d.go:146,536)
            assert name in ("lines2Dssbo", "polySSBO") and st["embedded"][0] in methods, f"{name} ({st['file']}:{st['line']}) has no AppendHIPNodes"
            continue
        pkg, recv, body = methods[name]
        assert pkg == st["package"], (name, pkg, st["package"])
        fields = {f for f, _ in st["fields"]}
        ftypes = dict(st["fields"])
        for e in st["embedded"]:
            fields |= {f for f, _ in structs[e]["fields"]}
            ftypes.update(dict(structs[e]["fields"]))
        touched = set(re.findall(r"\b%s\.([A-Za-z_]\w*)" % recv, body))
        assert touched <= fields, f"{name}: AppendHIPNodes names {sorted(touched - fields)}, the struct has {sorted(fields)}"
        missing = {f for f in fields - touched if (name, f) not in UNUSED_FIELDS}
        assert not missing, f"{name}: fields {sorted(missing)} are not copied"
        # children go in as children, not parameters
        for f in touched:
            if "Shader" in ftypes[f]:
                assert re.search(r"Op[23]\(", body), name
        ops = re.findall(r"gleval\.(HIP[A-Za-z0-9]+)", body)
        assert len(ops) == 1, (name, ops)
        c = go_ops[ops[0][3:].upper()]
        used_ops.setdefault(c, []).append(name)
        # the header's slot comment names this struct's fields: p0=r  /  p0..2=dims p3=round  /  aux = ...
        named = set(re.findall(r"=\s*([A-Za-z]\w*)", hcom.get(c, "")))
        named = {n for n in named if n in fields}
        assert named <= touched, (name, c, named, touched)
    assert not (set(methods) - set(nodes)), set(methods) - set(nodes)
    want = [c for c in enum if c not in ("GSDF_OP_INVALID", "GSDF_OP_COUNT")]
    assert sorted(used_ops) == sorted(want), (set(want) - set(used_ops), set(used_ops) - set(want))
    assert all(len(v) == 1 for v in used_ops.values()), {k: v for k, v in used_ops.items() if len(v) > 1}


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
                assert name in funcs, f"integration/go calls C.{name}: not declared in include/gsdf_hip.h"
                if rest.lstrip().startswith("("):
                    args = _call_args(go, m.end() + (len(rest) - len(rest.lstrip())))
                    n = 0 if not args.strip() else len(_split_args(args))
                    assert n == funcs[name], f"C.{name}: {n} arguments in integration/go, {funcs[name]} parameters in the header"
                    seen_calls.add(name)
            else:
                assert name in types, f"integration/go uses the type C.{name}: not in the headers"
                seen_types.add(name)
        for m in re.finditer(r"\bC\.(GSDF_[A-Z0-9_]+)\b", go):
            assert m.group(1) in consts, f"integration/go uses C.{m.group(1)}: not an enumerator or macro of the headers"
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
        for m in re.finditer(r"^\s*([a-z_][A-Za-z0-9_]*)\s+C\.(gsdf_[a-z0-9_]+)\s*$", go, re.M):   # a field of a Go struct: `st C.gsdf_mesh_stats`
            assert var_type.get(m.group(1), m.group(2)) == m.group(2), f"{m.group(1)} names two C struct types in one file"
            var_type[m.group(1)] = m.group(2)
        for v, t in var_type.items():
            if not re.search(r"struct\s+%s\s*\{" % t, h) and not re.search(r"\}\s*%s\s*;" % t, h):
                continue  # opaque handle
            fields = struct_fields(h, t)
            for m in re.finditer(r"\b%s\.([A-Za-z_][A-Za-z0-9_]*)\b" % re.escape(v), go):
                assert m.group(1) in fields, f"integration/go: {v}.{m.group(1)} -- {t} has no such member ({sorted(fields)})"
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
                    assert k.group(1) in fields, f"integration/go: C.{t}{{{k.group(1)}: ...}} -- no such member"
                    checked += 1
    assert checked >= 10, checked


def test_status_codes_are_what_the_shim_assumes():
    """The shim maps two status codes onto the reference's two error values (gleval/gpu.go:83-87) and treats 0 as success."""
    h = _headers()
    m = dict(re.findall(r"\b(GSDF_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", h))
    assert m.get("GSDF_OK") == "0" and "GSDF_ERR_EMPTY_BUFFERS" in m and "GSDF_ERR_LENGTH_MISMATCH" in m
    assert len(set(m.values())) == len(m)   # distinct
