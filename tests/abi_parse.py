"""Parses include/ecgpu.h and the Rust binding's `extern "C"` block into comparable signatures (rustc is not in this
image, so the binding is checked against the header mechanically: name, arity, argument and return types).
Used by tests/test_abi.py and tools/gen_rust_sys.py."""
import re

# C type (normalised: single spaces, '*' attached to the left token with one space before) -> Rust FFI type
C_TO_RUST = {
    "void": None,
    "int": "c_int",
    "size_t": "usize",
    "double": "f64",
    "double *": "*mut f64",
    "int *": "*mut c_int",
    "size_t *": "*mut usize",
    "uint8_t *": "*mut u8",
    "const uint8_t *": "*const u8",
    "void *": "*mut c_void",
    "const void *": "*const c_void",
    "const char *": "*const c_char",
    "const int *": "*const c_int",
    "const size_t *": "*const usize",
    "const void *const *": "*const *const c_void",
    "ecgpu_ctx *": "*mut EcgpuCtx",
    "const ecgpu_ctx *": "*const EcgpuCtx",
    "ecgpu_ctx **": "*mut *mut EcgpuCtx",
    "ecgpu_group *": "*mut EcgpuGroup",
    "const ecgpu_group *": "*const EcgpuGroup",
    "ecgpu_group **": "*mut *mut EcgpuGroup",
}


def _norm_ctype(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", " *", t)            # "uint8_t*x" / "uint8_t * x" -> "uint8_t *"
    t = re.sub(r"\* \*", "**", t)
    t = t.replace("* const", "*const").replace("*const *", "*const *")
    return t.strip()


def parse_header(path):
    """-> [(name, c_return_type, [(c_arg_type, arg_name), ...]), ...] in declaration order."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ecgpu_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = _norm_ctype(ret)
        if "typedef" in ret or not ret:
            continue
        alist = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.*?)(\w+)(\s*\[[^\]]*\])?$", a, flags=re.S)
                ctype, aname, arr = am.group(1), am.group(2), am.group(3)
                ctype = _norm_ctype(ctype + (" *" if arr else ""))
                alist.append((ctype, aname))
        out.append((name, ret, alist))
    return out


def rust_signature(decl):
    """header declaration -> (name, rust_return or None, [rust arg types])"""
    name, ret, args = decl
    return name, C_TO_RUST[ret], [C_TO_RUST[t] for t, _ in args]


def parse_rust_extern(path):
    """-> {name: (rust_return or None, [rust arg types])} for every `pub fn` inside `extern "C" { ... }` blocks."""
    src = open(path).read()
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
        for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", blk.group(1), flags=re.S):
            name, args, ret = m.group(1), m.group(2), m.group(3)
            types = []
            for a in [x for x in args.split(",") if x.strip()]:
                types.append(re.sub(r"\s+", " ", a.split(":", 1)[1].strip()))
            out[name] = (re.sub(r"\s+", " ", ret.strip()) if ret else None, types)
    return out
