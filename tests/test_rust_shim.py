"""The Rust side of the boundary, without a Rust toolchain (there is none in this image): rust/hbmpc_sys.rs is generated
from include/hbmpc_hip.h by tools/gen_rust_sys.py; these tests (i) regenerate it and compare, (ii) parse the header and
the .rs INDEPENDENTLY of the generator's emitter and compare every symbol's arity and pointer / integer kinds,
(iii) check that the exported symbols of the built library are exactly the header's functions, and (iv) check every
hbmpc_* call in the adaptor rust/gpu_shares.rs against the binding (exists, right number of arguments)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_sys as G  # noqa: E402

SYS = os.path.join(ROOT, "rust", "hbmpc_sys.rs")
ADAPTOR = os.path.join(ROOT, "rust", "gpu_shares.rs")


def parse_rs(path=SYS):
    """-> {name: ([(arg, rust type)], ret or None)} from the extern "C" block"""
    text = open(path).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block):
        al = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            an, ty = a.split(":", 1)
            al.append((an.strip(), ty.strip()))
        out[name] = (al, ret.strip() if ret else None)
    return out


def rs_kind(ty):
    levels = []
    while ty.startswith("*"):
        c = ty.startswith("*const ")
        levels.append(c)
        ty = ty[7:] if c else ty[5:]
    # outermost first in the text; the generator's kind lists the innermost level first
    return ("ptr", tuple(reversed(levels)), ty) if levels else ("int", ty)


def test_generated_file_is_current():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"])


def test_every_symbol_matches_the_header():
    funcs, enums = G.parse_header()
    rs = parse_rs()
    assert len(funcs) > 100 and {f[0] for f in funcs} == set(rs)
    for name, ret, args in funcs:
        rargs, rret = rs[name]
        assert len(args) == len(rargs), name
        for (an, ct), (rn, rt) in zip(args, rargs):
            assert G.kind(ct) == rs_kind(rt), (name, an, ct, rt)
        if ret == "void":
            assert rret is None, name
        elif ret.startswith("const char"):
            assert rret == "*const c_char", name
        else:
            assert rret == ret and ret in ("ShareErrorCode", "FieldKind"), name
    text = open(SYS).read()
    for ename, items in enums.items():
        for k, v in items:
            assert re.search(rf"pub const {k}: {ename} = {v};", text), (ename, k)
    # the ShareErrorCode values are the reference's (mpc/src/ffi/c_bindings/share/mod.rs:18-37), in its order
    assert [k for k, _ in enums["ShareErrorCode"]][:9] == ["ShareSuccess", "InsufficientShares", "DegreeMismatch", "IdMismatch",
                                                         "InvalidInput", "TypeMismatch", "NoSuitableDomain",
                                                         "PolynomialOperationError", "DecodingError"]
    assert "pub data: [u64; 4]" in text and "#[repr(C)]" in text


def test_library_exports_exactly_the_declared_symbols():
    lib = os.path.join(ROOT, "mpc-protocols_amd", "libhbmpc_hip.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("library not built")
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in nm.splitlines() if " T " in l and l.split()[-1].startswith("hbmpc_")}
    assert exported == set(parse_rs())


def split_top_level(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def test_adaptor_calls_match_the_binding():
    rs = parse_rs()
    text = open(ADAPTOR).read()
    calls = 0
    for m in re.finditer(r"sys::(hbmpc_\w+)\(", text):
        name = m.group(1)
        assert name in rs, name
        depth, i = 1, m.end()
        while depth:
            depth += text[i] in "([{"
            depth -= text[i] in ")]}"
            i += 1
        args = split_top_level(text[m.end():i - 1])
        assert len(args) == len(rs[name][0]), (name, len(args), len(rs[name][0]))
        calls += 1
    assert calls >= 8
    # the pipelines behind the C ABI: every hbmpc_pipe_* entry point is reachable from the adaptor
    for name in rs:
        if name.startswith("hbmpc_pipe_"):
            assert f"sys::{name}(" in text, name
    # the trait surface of mpc/src/common/mod.rs:101-128 and every ShareErrorCode are covered
    assert "impl SecretSharingScheme<Fr> for GpuRobustShare" in text
    for op in ("impl Add for", "impl Sub for", "impl Add<Fr> for", "impl Sub<Fr> for", "impl Mul<Fr> for"):
        assert op in text
    for code in ("InsufficientShares", "DegreeMismatch", "IdMismatch", "TypeMismatch", "InvalidInput", "NoSuitableDomain",
                 "PolynomialOperationError", "DecodingError"):
        assert f"sys::{code} =>" in text, code
