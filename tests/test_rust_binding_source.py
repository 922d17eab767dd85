"""The Rust FFI crate ships as SOURCE ONLY (no Rust toolchain here).  What CAN be checked without a
compiler: it declares every entry point of include/mirt.h, and its #[repr(C)] structs list the same
fields in the same order as the (layout-tested) ctypes mirror."""
import re
from pathlib import Path

import weekend_raytracer_wgpu_amd as m
from weekend_raytracer_wgpu_amd import _abi

ROOT = Path(__file__).resolve().parent.parent
RS = (ROOT / "rust" / "mirt-sys" / "src" / "lib.rs").read_text()
HEADER = (ROOT / "include" / "mirt.h").read_text()


def test_every_header_symbol_is_declared():
    declared = set(re.findall(r"\b(mirt_[a-z0-9_]+)\s*\(", HEADER))
    in_rust = set(re.findall(r"pub fn (mirt_[a-z0-9_]+)\s*\(", RS))
    assert declared == in_rust, declared ^ in_rust


def test_struct_fields_match_the_ctypes_mirror():
    for name in ("MirtSphere", "MirtTextureDescriptor", "MirtMaterial", "MirtGpuCamera", "MirtSkyState", "MirtCamera",
                 "MirtSamplingParams", "MirtScene", "MirtParams", "MirtStats", "MirtGridPlan"):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, RS, re.S).group(1)
        rust_fields = re.findall(r"pub (\w+):", body)
        py_fields = [f[0] for f in getattr(_abi, name)._fields_]
        assert rust_fields == py_fields, name


def _header_enum_constants(prefix):
    """name -> value of every enumerator of include/mirt.h that starts with `prefix`."""
    out = {}
    for name, expr in re.findall(r"^\s*(%s[A-Z0-9_]*)\s*=\s*([^,/\n]+)" % prefix, HEADER, re.M):
        out[name] = eval(expr.strip().replace("u", ""))
    return out


def test_every_mode_and_flag_constant_matches_the_header():
    consts = {**_header_enum_constants("MIRT_MODE_"), **_header_enum_constants("MIRT_FLAG_")}
    assert len(consts) == 12, sorted(consts)              # 2 modes + 10 flags: a new one must be added to the crate and here
    for name, value in consts.items():
        expr = re.search(r"pub const %s: u32 = ([^;]+);" % name, RS)
        assert expr, f"{name} missing from the Rust crate"
        assert eval(expr.group(1)) == value == getattr(_abi, name), name
    in_rust = set(re.findall(r"pub const (MIRT_(?:MODE|FLAG)_[A-Z0-9_]+): u32", RS))
    assert in_rust == set(consts), in_rust ^ set(consts)


def test_every_status_code_matches_the_header():
    codes = {**_header_enum_constants("MIRT_OK"), **_header_enum_constants("MIRT_ERR_")}
    assert codes["MIRT_OK"] == 0 and len(codes) == 22 and all(v < 0 for k, v in codes.items() if k != "MIRT_OK")
    in_rust = dict((n, int(v)) for n, v in re.findall(r"pub const (MIRT_(?:OK|ERR_[A-Z_]+)): c_int = (-?\d+);", RS))
    assert in_rust == codes, set(in_rust.items()) ^ set(codes.items())
    for name, value in codes.items():
        if name != "MIRT_OK":
            assert _abi.STATUS[value] == name and getattr(_abi, name) == value


def test_the_per_call_sample_limit_matches_the_header():
    value = eval(re.search(r"#define MIRT_MAX_SPP_PER_CALL \(([^)]+)\)", HEADER).group(1).replace("u", ""))
    assert value == 1 << 24 == _abi.MIRT_MAX_SPP_PER_CALL
    assert eval(re.search(r"pub const MIRT_MAX_SPP_PER_CALL: u32 = ([^;]+);", RS).group(1)) == value


def _rust_type_of(ctype):
    """The Rust spelling of a ctypes field type (pointers: any `*const` / `*mut`)."""
    import ctypes as C
    scalars = {C.c_uint32: "u32", C.c_uint64: "u64", C.c_float: "f32", C.c_double: "f64", C.c_int32: "i32", C.c_uint8: "u8"}
    if ctype in scalars:
        return scalars[ctype]
    if isinstance(ctype, type) and issubclass(ctype, C.Array):
        return "[%s; %d]" % (_rust_type_of(ctype._type_), ctype._length_)
    if isinstance(ctype, type) and issubclass(ctype, C.Structure):
        return ctype.__name__
    if isinstance(ctype, type) and (issubclass(ctype, C._Pointer) or ctype is C.c_void_p):
        return "*"
    raise AssertionError(f"unmapped ctypes type {ctype}")


def test_struct_field_types_match_the_ctypes_mirror():
    """Names and order are checked above; this compares every field's TYPE (u32 / u64 / f32 / f64 / arrays / nested structs /
    pointers) -- with #[repr(C)] that fixes the layout, which the ctypes mirror's layout tests pin against the header."""
    for name in ("MirtSphere", "MirtTextureDescriptor", "MirtMaterial", "MirtGpuCamera", "MirtSkyState", "MirtCamera",
                 "MirtSamplingParams", "MirtScene", "MirtParams", "MirtStats", "MirtGridPlan"):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, RS, re.S).group(1)
        rust = dict(re.findall(r"pub (\w+): ([^,\n]+),", body))
        assert re.search(r"#\[repr\(C\)\]\s*(#\[derive\([^\]]*\)\]\s*)?pub struct %s " % name, RS), f"{name} is not #[repr(C)]"
        for field, ctype in getattr(_abi, name)._fields_:
            want = _rust_type_of(ctype)
            got = rust[field].strip()
            assert (got.startswith("*const ") or got.startswith("*mut ")) if want == "*" else got == want, (name, field, got, want)


def test_function_signatures_have_the_header_arity_and_return_type():
    """Every prototype of the header against its `pub fn`: same number of parameters, and the same kind of return value."""
    protos = re.findall(r"^([A-Za-z_][A-Za-z0-9_ \*]*?)\b(mirt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", HEADER, re.M | re.S)
    assert len(protos) >= 30
    for ret, fn, args in protos:
        args = " ".join(args.split())
        n_c = 0 if args in ("", "void") else args.count(",") + 1
        m_rs = re.search(r"pub fn %s\s*\(([^)]*)\)\s*(->\s*([^;]+))?;" % fn, RS)
        assert m_rs, fn
        rs_args = m_rs.group(1).strip()
        n_rs = 0 if not rs_args else rs_args.count(",") + 1
        assert n_c == n_rs, (fn, args, rs_args)
        ret = ret.strip()
        want = {"int": "c_int", "uint32_t": "u32", "float": "f32", "void": None, "const char*": "*const c_char"}[ret.replace(" *", "*")]
        got = m_rs.group(3).strip() if m_rs.group(3) else None
        assert got == want, (fn, ret, got)
