"""The C-ABI library: loads without a GPU, exports every symbol include/trx.h
declares, and fails loudly (never falls back to a CPU path)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="trx.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trx_[a-z0-9_]+)\s*\(", text)))


def test_the_boundary_header_carries_no_development_surface():
    """include/trx.h is what INTEGRATION.md binds; diagnostics and the tuning word live in include/trx_dev.h."""
    product, dev = declared_symbols(), declared_symbols("trx_dev.h")
    assert not [n for n in product if n.startswith("trx_debug_") or n == "trx_set_kernel_variant"]
    assert dev and all(n.startswith("trx_debug_") or n == "trx_set_kernel_variant" for n in dev)
    assert not set(product) & set(dev)
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert not [n for n in dev if n in integration]
    # and it is plain C as well
    subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", "-I",
                    os.path.join(ROOT, "include"), "-"], input=b'#include "trx_dev.h"\nint main(void) { return 0; }\n', check=True)


def test_every_declared_symbol_is_exported_and_bound(trx):
    from tray_racing_amd import _lib
    declared = sorted(set(declared_symbols()) | set(declared_symbols("trx_dev.h")))
    assert len(declared) >= 35
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (trx_[a-z0-9_]+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(_lib.SIGNATURES), sorted(set(declared) ^ set(_lib.SIGNATURES))
    lib = trx.load()
    assert lib.trx_abi_version() == 1
    assert [lib.trx_tri_format_bytes(f) for f in (0, 1, 2, 3)] == [24, 36, 36, 0]


def test_struct_sizes_match_the_header(trx):
    from tray_racing_amd import _lib
    assert C.sizeof(_lib.View) == 160 and C.sizeof(_lib.Ray) == 32 and C.sizeof(_lib.Hit) == 8
    assert C.sizeof(_lib.RayHit) == 16 and C.sizeof(_lib.Shard) == 16 and C.sizeof(_lib.Stats) == 64
    assert trx.HIT_DTYPE.itemsize == 8 and trx.RAY_DTYPE.itemsize == 32


def test_product_never_references_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "tray_racing_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("oracle's", "").replace("the oracle", "").replace("an oracle", ""), \
                    os.path.join(dirpath, f)
    from tray_racing_amd import _lib
    needed = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH]).decode()
    assert "liboracle" not in needed
    syms = subprocess.check_output(["nm", "-D", _lib.LIB_PATH]).decode()
    assert "orc_" not in syms


def test_shard_tiles(trx):
    from tray_racing_amd import _lib, dist
    lib = trx.load()
    for w, h in [(1920, 1080), (52, 44), (8, 8), (3, 3)]:
        for world in (1, 2, 3, 8):
            got = [lib.trx_shard_tiles(w, h, _lib.Shard(r, world, 0, 0)) for r in range(world)]
            assert got == [dist.shard_tiles(w, h, r, world) for r in range(world)]
            assert sum(got) == ((w + 7) // 8) * ((h + 7) // 8)


def test_invalid_arguments_are_reported_not_fatal(trx):
    lib = trx.load()
    from tray_racing_amd import _lib
    flat = trx.flat_build(trx.gen_scene("soup", 50, 1)[0])
    h = C.c_void_p()
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib.trx_scene_create(None, 0, None, 0, 1, None, 0, 0, 0, C.byref(h))
    assert rc == _lib.TRX_ERR_INVALID and b"empty node buffer" in lib.trx_last_error()
    rc = lib.trx_scene_create(P(flat.nodes), flat.n_nodes, P(flat.tri_verts), flat.n_tris, 9, None, 0, 0, 0, C.byref(h))
    assert rc == _lib.TRX_ERR_INVALID and b"tri_format" in lib.trx_last_error()
    v = _lib.View()
    rc = lib.trx_view_from_camera((C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0, 0, 0), 90.0, 8.0, 8.0, C.byref(v))
    assert rc == _lib.TRX_ERR_INVALID
    with pytest.raises(trx.TrxError, match="unknown scene"):
        trx.gen_scene("no_such_scene")
    with pytest.raises(trx.TrxError, match="Error while loading"):
        trx.load_meshs("/nonexistent/model.obj")


def test_malformed_nodes_are_rejected_before_any_kernel_runs(trx):
    """Structural validation happens on the host: an out-of-range child or primitive
    index would make a kernel read out of bounds."""
    lib = trx.load()
    from tray_racing_amd import _lib
    flat = trx.flat_build(trx.gen_scene("soup", 400, 1)[0])
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def create(nodes, n_tris=flat.n_tris):
        h = C.c_void_p()
        return lib.trx_scene_create(P(nodes), nodes.shape[0], P(flat.tri_verts), n_tris, 1, None, 0, 0, 0, C.byref(h))

    bad = flat.nodes.copy()
    bad[0, 4] = 10**9                                   # child_base_idx far outside
    assert create(bad) == _lib.TRX_ERR_FORMAT and b"children" in lib.trx_last_error()
    assert create(flat.nodes, n_tris=flat.n_tris - 10) == _lib.TRX_ERR_FORMAT   # primitives out of range
    bad = flat.nodes.copy()
    bad.view(np.uint8).reshape(-1, 80)[0, 15] ^= 0xFF   # imask disagrees with child_meta
    assert create(bad) == _lib.TRX_ERR_FORMAT and b"imask" in lib.trx_last_error()
    bad = flat.nodes.copy()
    meta = bad.view(np.uint8).reshape(-1, 80)[:, 24:32]
    i, s = np.argwhere((meta != 0) & ((meta & 0x18) != 0x18))[0]
    meta[i, s] = 0xA0 | (meta[i, s] & 0x1F)             # 0b101 is not a unary count
    assert create(bad) == _lib.TRX_ERR_FORMAT and b"leaf meta" in lib.trx_last_error()


def test_no_cpu_fallback_without_a_device(trx, has_gpu):
    if has_gpu:
        pytest.skip("a GPU is present")
    flat = trx.flat_build(trx.gen_scene("soup", 50, 1)[0])
    with pytest.raises(trx.TrxError) as e:
        trx.Scene(flat)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def build_c_consumer(name, tmp_path):
    """gcc -std=c11 -pedantic: include/trx.h must be valid C and libtrx.so linkable from C."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    libdir = os.path.join(root, "tray_racing_amd")
    subprocess.check_call(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", name + ".c"), "-o", exe, "-pthread", "-L", libdir, "-ltrx",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_header_is_plain_c_and_the_library_links_from_c(trx, tmp_path):
    import subprocess
    exe = build_c_consumer("host_only", tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert out.stdout.startswith("objects 5 triangles")


def _c_prototypes():
    """name -> number of parameters, for every function include/trx.h declares."""
    text = open(os.path.join(ROOT, "include", "trx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(trx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return protos, text


def _c_struct_fields(text, name):
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S)
    assert m, name
    n = 0
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if decl:
            n += 1   # one declarator per field in this header (arrays count as one field, as in the Rust struct)
    return n


def test_integration_md_matches_the_header():
    """The Rust binding shown in INTEGRATION.md cannot be compiled here (no Rust toolchain), so it is checked
    mechanically: every function of its extern "C" block exists in include/trx.h with the same number of
    parameters, every trx_* type it uses is declared in the document, and every #[repr(C)] struct it declares has
    as many fields as the header's struct."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = "\n".join(re.findall(r"```rust\n(.*?)```", md, flags=re.S))
    protos, header = _c_prototypes()
    ext = re.search(r'extern "C" \{(.*?)\n\}', rust, flags=re.S).group(1)
    ext = re.sub(r"//[^\n]*", "", ext)
    fns = re.findall(r"pub fn (trx_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", ext, flags=re.S)
    assert len(fns) >= 15
    for name, args in fns:
        assert name in protos, "%s is not declared in include/trx.h" % name
        n = 0 if not args.strip() else args.count(":")
        assert n == protos[name], "%s: %d parameters in INTEGRATION.md, %d in trx.h" % (name, n, protos[name])
    structs = dict(re.findall(r"pub struct (trx_[a-z0-9_]+)\s*\{(.*?)\}", rust, flags=re.S))
    used = set(re.findall(r"\b(trx_[a-z0-9_]+)\b", re.sub(r"pub fn trx_[a-z0-9_]+", "", ext)))
    assert used <= set(structs), "types used but not declared in INTEGRATION.md: %s" % sorted(used - set(structs))
    for name, body in structs.items():
        if "_private" in body:
            continue   # opaque handle
        assert body.count(":") - body.count("::") * 2 == _c_struct_fields(header, name), name
    # every call the shim makes through ffi:: is bound
    called = set(re.findall(r"ffi::(trx_[a-z0-9_]+)\s*\(", rust))
    assert called <= {n for n, _ in fns}, sorted(called - {n for n, _ in fns})


def test_multi_gpu_entry_points_validate_their_arguments(trx):
    """trx_comm_* / trx_gather_shards / trx_assemble_frames refuse bad arguments with an error string before they
    touch RCCL or the device (so this runs without a GPU)."""
    import ctypes as C
    lib = trx.load()
    ident = (C.c_ubyte * 128)()
    comm = C.c_void_p()
    assert lib.trx_comm_create(None, 0, 1, 0, C.byref(comm)) == -1 and b"null" in lib.trx_last_error()
    assert lib.trx_comm_create(ident, 3, 2, 0, C.byref(comm)) == -1 and b"rank 3 of 2" in lib.trx_last_error()
    assert lib.trx_comm_create(ident, 0, 0, 0, C.byref(comm)) == -1
    assert lib.trx_comm_unique_id(None) == -1
    assert lib.trx_gather_shards(None, None, 64, None) == -1
    assert lib.trx_gather_shards_root(None, None, 64, 0, None) == -1
    assert lib.trx_assemble_frames(None, 64, 8, 8, 1, 1, None, None) == -1
    assert lib.trx_comm_world_size(None) == 0
    lib.trx_comm_destroy(None)


def test_missing_rccl_is_an_error_code_not_a_crash():
    """RCCL is loaded on first use; a host whose loader cannot find it gets TRX_ERR_NO_DEVICE and the loader's message
    from every trx_comm_* call (the library once called dlerror() twice there and crashed in strlen(NULL)).  The loader's
    answer is cached per process, hence the child process."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "import tray_racing_amd as T\n"
        "lib = T.load()\n"
        "ident = (C.c_ubyte * 128)()\n"
        "rc = lib.trx_comm_unique_id(ident)\n"
        "msg = lib.trx_last_error()\n"
        "comm = C.c_void_p()\n"
        "rc2 = lib.trx_comm_create(ident, 0, 1, 0, C.byref(comm))\n"
        "print(rc, rc2, msg.decode())\n" % ROOT)
    env = dict(os.environ, TRX_RCCL_LIBRARY="/nonexistent/librccl-missing.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-600:]
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) == -2 and int(rc2) == -2, out.stdout
    assert "librccl.so not found" in msg and "librccl-missing.so" in msg


def test_build_device_setter(trx, has_gpu):
    """trx_set_build_device: -1 (host) always works; a device that does not exist is refused, never silently ignored."""
    lib = trx.load()
    assert lib.trx_set_build_device(-1) == 0
    assert lib.trx_set_build_device(97) == -2 and b"no HIP device 97" in lib.trx_last_error()
    if not has_gpu:
        assert lib.trx_set_build_device(0) == -2
    assert lib.trx_set_build_device(-1) == 0
