"""The C-ABI library builds, loads without a GPU, and exports every symbol include/h2v.h declares.
No compute call is made here (there is no GPU in this container): the only calls are the probes
that must work anywhere, and the check that the product fails loudly instead of falling back."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "h2v.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(h2v_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import halo2_verifier_amd as h2v
    if not os.path.exists(h2v.lib_path()):
        subprocess.check_call(["make", "-j", "4", "-C", os.path.join(ROOT, "halo2_verifier_amd", "csrc")])
    return ctypes.CDLL(h2v.lib_path())


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from halo2_verifier_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.load_library()   # sets restype/argtypes for every symbol; raises if one is missing


def test_rust_binding_covers_the_header():
    """integration/rust/ffi.rs (uncompiled here: no rustc) must bind every function and mirror every constant of include/h2v.h."""
    rs = open(os.path.join(ROOT, "integration", "rust", "ffi.rs")).read()
    bound = sorted(set(re.findall(r"pub fn (h2v_[a-z0-9_]+)\s*\(", rs)))
    assert bound == declared_functions()
    text = open(HEADER).read()
    for name, val in re.findall(r"#define (H2V_[A-Z0-9_]+) \(?(-?\d+)\)?", text):
        m = re.search(r"pub const %s: \w+ = (-?\d+);" % name, rs)
        assert m and int(m.group(1)) == int(val), name
    for struct in ("h2v_options", "h2v_tuning"):
        fields = re.search(r"typedef struct %s \{([^}]*)\}" % struct, text).group(1)
        c_names = [n.strip() for decl in fields.split(";") if decl.strip() for n in decl.split(None, 1)[1].split(",")]
        assert c_names == re.findall(r"pub (\w+): \w+", re.search(r"pub struct %s \{([^}]*)\}" % struct, rs).group(1)), struct
    # the ctypes mirrors have the same fields in the same order
    from halo2_verifier_amd import verifier
    for struct, cls in (("h2v_options", verifier._Options), ("h2v_tuning", verifier._Tuning)):
        fields = re.search(r"typedef struct %s \{([^}]*)\}" % struct, text).group(1)
        c_names = [n.strip() for decl in fields.split(";") if decl.strip() for n in decl.split(None, 1)[1].split(",")]
        assert c_names == [f[0] for f in cls._fields_], struct


def test_error_codes_mirror_plonk_error():
    text = open(HEADER).read()
    codes = dict(re.findall(r"#define (H2V_ERR_[A-Z_]+) \((-\d+)\)", text))
    # plonk/mod.rs:19-32 declaration order
    order = ["INVALID_INSTANCES", "CONSTRAINT_SYSTEM_FAILURE", "BOUNDS_FAILURE", "OPENING", "TRANSCRIPT", "INSTANCE_TOO_LARGE"]
    assert [int(codes["H2V_ERR_" + n]) for n in order] == [-1, -2, -3, -4, -5, -6]
    from halo2_verifier_amd import PlonkError
    assert [PlonkError[n.title().replace("_", "")].value for n in order] == [-1, -2, -3, -4, -5, -6]


def test_no_cpu_fallback():
    """Without a HIP device the product refuses to create a context: there is no CPU path to fall back to."""
    import halo2_verifier_amd as h2v
    if h2v.device_count() > 0:
        pytest.skip("a GPU is visible")
    srs = open(os.path.join(ROOT, "tests", "golden", "kzg_bn254_8.srs"), "rb").read()
    params = srs[:4] + srs[4:68] + srs[-256:]
    with pytest.raises(h2v.H2VError) as e:
        h2v.Context(h2v.ParamsKZG(params, h2v.SerdeFormat.RawBytes))
    assert e.value.code == -18


def test_product_does_not_link_or_import_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "halo2_verifier_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("liboracle", "oracle_lib", "h2o_", "import oracle", "../oracle", "oracle/"):
                    if needle in text:
                        # comments that say "shares no code with oracle/" are the only allowed mentions
                        lines = [l for l in text.splitlines() if needle in l and not l.strip().startswith(("//", "#", "*", '"""')) and "no code with oracle" not in l]
                        assert not lines, (f, lines)


def test_product_reads_no_environment_variable():
    """The environment of a proof verifier must not select its code path: kernel variants are chosen from the launch shape, and
    forced only through h2v_ctx_set_tuning / h2v_options (debug fields of the C ABI)."""
    pkg = os.path.join(ROOT, "halo2_verifier_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(pkg)):
        if f.endswith((".hip", ".h")):
            for i, line in enumerate(open(os.path.join(pkg, f), errors="replace"), 1):
                if "getenv" in line:
                    hits.append(f"{f}:{i}")
    assert not hits, hits
    out = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(pkg, "build", "libh2v_amd.so")], capture_output=True, text=True).stdout
    assert "getenv" not in out


def test_options_struct_size_guards_the_layout(lib):
    """h2v_ctx_create_ex rejects an h2v_options whose struct_size is not a layout the library knows (a caller built against another
    revision of the header), before it looks at any other argument that needs a device."""
    from halo2_verifier_amd import verifier
    lib.h2v_abi_version.restype = ctypes.c_int
    assert lib.h2v_abi_version() == 3
    lib.h2v_last_error.restype = ctypes.c_char_p
    srs = open(os.path.join(ROOT, "tests", "golden", "kzg_bn254_8.srs"), "rb").read()
    params = srs[:4] + srs[4:68] + srs[-256:]
    for bad in (0, 12, ctypes.sizeof(verifier._Options) + 8):
        o = verifier._Options(bad, 0, 0, 1, 0)
        h = ctypes.c_void_p()
        rc = lib.h2v_ctx_create_ex(params, ctypes.c_size_t(len(params)), 1, None, ctypes.c_size_t(0), 0, 0, ctypes.byref(o), ctypes.byref(h))
        assert rc == -16 and b"struct_size" in lib.h2v_last_error()


def test_cpp_mirror_header_compiles():
    """include/h2v.hpp (the C++ host-side mirror of the reference surface) and its harness are valid C++17 against h2v.h."""
    import subprocess
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "harness.cpp")], check=True)
