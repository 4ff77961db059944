"""integration/rust/src/index/gpu.rs cannot be compiled here (no Rust toolchain): check its
`extern "C"` block against include/vectorlite_amd.h instead -- every function it declares must exist
in the header with the same arity, argument kinds and return type."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUST_TO_C = {
    "u64": "uint64_t", "u32": "uint32_t", "c_int": "int", "f64": "double",
    "*const f64": "const double *", "*mut f64": "double *", "*const f32": "const float *",
    "*const u64": "const uint64_t *", "*mut u64": "uint64_t *",
    "*const vl_index": "const vl_index *", "*mut vl_index": "vl_index *", "*mut *mut vl_index": "vl_index **",
    "*const c_char": "const char *", "": "void",
    "*const vl_vlc_doc": "const vl_vlc_doc *", "*mut vl_vlc_doc": "vl_vlc_doc *", "*mut *mut vl_vlc_doc": "vl_vlc_doc **",
    "*const vl_comm": "const vl_comm *", "*mut vl_comm": "vl_comm *", "*mut *mut vl_comm": "vl_comm **",
    "*const u8": "const uint8_t *", "*mut u8": "uint8_t *", "*const c_int": "const int *", "*mut c_int": "int *",
}


def c_decls():
    text = open(os.path.join(ROOT, "include", "vectorlite_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(vl_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        norm = []
        for a in [x.strip() for x in args.split(",")]:
            if a == "void" or not a:
                continue
            a = re.sub(r"\s+", " ", a)
            a = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a  # drop the parameter name
            a = re.sub(r"\s*\*\s*", " *", a).replace("* *", "**").replace(" * *", " **")
            a = re.sub(r"\*\s+\*", "**", a).strip()
            norm.append(a)
        out[name] = (re.sub(r"\s+", " ", ret), norm)
    return out


def rust_decls():
    text = open(os.path.join(ROOT, "integration", "rust", "src", "index", "gpu.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"fn (vl_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "").strip()
        kinds = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        out[name] = (ret, kinds)
    return out


def canon(c_type):
    return re.sub(r"\s+", " ", c_type.replace("*", " * ")).replace("* *", "**").strip().replace(" * ", " *").replace(" *", " *")


def test_every_rust_extern_matches_the_header():
    c, r = c_decls(), rust_decls()
    assert len(r) >= 15
    for name, (ret, kinds) in r.items():
        assert name in c, f"{name} is not declared in include/vectorlite_amd.h"
        c_ret, c_args = c[name]
        assert RUST_TO_C[ret] == c_ret, (name, ret, c_ret)
        want = [canon(RUST_TO_C[k]) for k in kinds]
        got = [canon(a) for a in c_args]
        assert want == got, (name, want, got)
