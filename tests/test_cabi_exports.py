"""CPU-only: the C-ABI library loads, exports every symbol include/lurk_hip.h declares, and fails
loudly (no CPU fallback) when no gfx950 device is usable."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "lurk_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:lurk_hip|mult_pippenger|cuda_pippenger)_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from lurk_beta_amd import _lib

    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lurk_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    for name in _lib.SIGNATURES:
        assert name in declared, f"{name} bound but not declared in the header"


def test_host_side_constants_match_oracle_without_a_gpu():
    from lurk_beta_amd import poseidon_constants
    from oracle import pyref as R

    for f in (1, 2):
        for arity in (3, 8):
            rf, rp, rc, mds = poseidon_constants(f, arity)
            assert (rf, rp) == R.round_numbers(arity)
            assert rc == list(R.round_constants(f, arity))
            assert mds == [x for row in R.mds_matrix(f, arity) for x in row]


def test_host_side_point_helpers_without_a_gpu():
    from lurk_beta_amd import point_sum, point_to_affine
    from oracle import coracle as C

    a, b = C.gen_mul(0, 5), C.gen_mul(0, 7)
    assert point_to_affine(0, point_sum(0, np.stack([a, b]))) == C.jac_to_affine(0, C.gen_mul(0, 12))
    assert point_to_affine(0, point_sum(0, np.zeros((0, 12), dtype=np.uint64))) == (0, 0)


def test_compute_entry_points_fail_loudly_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lurk_beta_amd import LurkHipError, msm, ntt, poseidon_batch

    with pytest.raises(LurkHipError, match="no CPU fallback"):
        poseidon_batch(1, 8, np.zeros((1, 8, 4), dtype=np.uint64))
    with pytest.raises(LurkHipError):
        msm(0, np.zeros((1, 8), dtype=np.uint64), np.zeros((1, 4), dtype=np.uint64))
    with pytest.raises(LurkHipError):
        ntt(1, np.zeros((8, 4), dtype=np.uint64))


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/lurk_hip.h must stay a C header (a cgo / bindgen / cffi consumer compiles it as C): strict C99 and C++14, no warnings."""
    import shutil
    import subprocess

    inc = os.path.join(ROOT, "include")
    for cc, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++14", "cpp")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not available")
        src = tmp_path / f"h.{ext}"
        src.write_text('#include "lurk_hip.h"\nint main(void) { return 0; }\n')
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + inc, "-fsyntax-only", str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_inline_asm_operands_avoid_the_clobbered_scratch_registers(tmp_path):
    """The generated multipliers use fixed scratch registers (declared as clobbers); compile the accumulate kernel to ISA and check
    that no asm operand was allocated to one of them (bench_tools/check_asm_operands.py)."""
    import shutil
    import subprocess
    import sys

    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "lurk_beta_amd", "csrc")
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable", "-c", "msm_acc.hip",
                        "-save-temps=obj", "-o", str(tmp_path / "msm_acc.o")], cwd=csrc, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    isa = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert isa
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "bench_tools", "check_asm_operands.py"), str(tmp_path / isa[0])], capture_output=True, text=True)
    assert chk.returncode == 0, chk.stdout[-800:]


def test_sort_kernels_fit_beside_a_resident_accumulation(tmp_path):
    """Commitments in flight: the persistent accumulation keeps ONE wave on every SIMD (msm_acc.hip) and the next commitment's sort
    runs under it in 1024-thread workgroups - four waves per SIMD.  gfx950 allocates wave64 VGPRs in granules of 8 out of 512 per SIMD:
    a sort kernel whose four waves do not fit beside the accumulate wave waits for an accumulation to END (measured: 988 -> 839
    Mscalar-mul/s at 2^22 when msm_scatter1 grew from 65 to 84 registers).  Checked on the compiler's own resource remarks."""
    import shutil
    import subprocess

    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "lurk_beta_amd", "csrc")

    def usage(src):
        r = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable",
                            "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / (src + ".o"))], cwd=csrc, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-800:]
        out, name = {}, None
        for ln in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r" VGPRs: (\d+)", ln)
            if m and name:
                out[name] = int(m.group(1))
        return out

    granule = lambda v: -(-v // 8) * 8
    acc = usage("msm_acc_persistent.hip")
    persistent = max(v for k, v in acc.items() if "persistent" in k)
    assert persistent <= 256
    sort = usage("msm_sort.hip")
    checked = 0
    for k, v in sort.items():
        if any(x in k for x in ("hist1", "scatter1", "part2")) and ("Li20E" in k or "Li16E" in k or "part2" in k):
            assert 4 * granule(v) + granule(persistent) <= 512, (k, v, persistent)
            checked += 1
    assert checked >= 5
    # the latency-bound tail kernels (finalize, the bit-plane levels) run one 256-thread workgroup's wave per SIMD beside TWO resident
    # accumulations: held to 128 registers (__launch_bounds__(256, 4)) - with 183 every level waited for an accumulation to end
    for src, needle in (("msm_finalize.hip", "msm_finalize_kernel"), ("msm_reduce.hip", "msm_planes29_kernel")):
        tail = {k: v for k, v in usage(src).items() if needle in k}
        assert tail and all(granule(v) + 2 * granule(persistent) <= 512 for v in tail.values()), tail


def test_rust_ffi_matches_the_header():
    """rust/lurk-hip-sys/src/ffi.rs (the sys crate's extern block) is generated from include/lurk_hip.h: the committed file must be
    what the generator emits today, its functions must be exactly the header's (and exported by the built library), and - independently
    of the generator - every function must take as many arguments as the ctypes signature the tests call it with, with pointer /
    integer / by-value kinds agreeing."""
    import ctypes
    import subprocess
    import sys

    from lurk_beta_amd import _lib

    r = subprocess.run([sys.executable, os.path.join(ROOT, "rust", "gen_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rs = open(os.path.join(ROOT, "rust", "lurk-hip-sys", "src", "ffi.rs")).read()
    block = rs[rs.index('extern "C" {'):]
    fns = dict(re.findall(r"pub fn (\w+)\((.*?)\)(?: -> [^;]+)?;", block))
    assert sorted(fns) == _declared_symbols()
    lib = _lib.load()
    for name, params in fns.items():
        assert hasattr(lib, name)
        res, args = _lib.SIGNATURES[name]
        plist = [p.split(":", 1)[1].strip() for p in params.split(", ")] if params.strip() else []
        assert len(plist) == len(args), name
        for rust_t, ct in zip(plist, args):
            is_ptr_rs = rust_t.startswith("*") or rust_t.endswith("_fn")
            is_ptr_ct = ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, "contents")
            assert is_ptr_rs == is_ptr_ct, (name, rust_t, ct)
            if not is_ptr_rs:
                width = {"c_int": 4, "c_uint": 4, "u32": 4, "usize": 8, "u64": 8, "bool": 1}[rust_t]
                assert ctypes.sizeof(ct) == width, (name, rust_t, ct)
    # the #[repr(C)] structs: field count and size against what the header compiles to
    for struct, size in (("lurk_hip_store_node", 32), ("lurk_hip_w2_patch", 24), ("lurk_hip_rust_error", 16)):
        assert f"pub struct {struct} {{" in rs
    assert ctypes.sizeof(_lib.RustError) == 16
    # the safe wrapper only calls functions the extern block declares
    src_dir = os.path.join(ROOT, "rust", "lurk-hip-sys", "src")
    for name in sorted(os.listdir(src_dir)):  # lib.rs and its modules (store.rs: the Poseidon hooks, dump.rs: the LURKDUMP writers)
        if name == "ffi.rs" or not name.endswith(".rs"):
            continue
        text = open(os.path.join(src_dir, name)).read()
        called = set(re.findall(r"\b((?:lurk_hip|mult_pippenger|cuda_pippenger)_[a-z0-9_]+)\s*\(", text))
        assert called <= set(fns), (name, called - set(fns))
        if name in ("lib.rs", "store.rs"):
            assert called, name
        # arity of every call: as many top-level arguments as the extern declaration has parameters
        for m in re.finditer(r"\b((?:lurk_hip|mult_pippenger|cuda_pippenger)_[a-z0-9_]+)\s*\(", text):
            fn, i, depth, args, cur = m.group(1), m.end(), 1, 0, ""
            if text[m.start() - 3:m.start()] == "fn " or "`" in text[max(0, m.start() - 1):m.start()]:
                continue
            while depth and i < len(text):
                ch = text[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                if ch == "," and depth == 1:
                    args += bool(cur.strip())
                    cur = ""
                elif depth:
                    cur += ch
                i += 1
            args += bool(cur.strip())
            want = len([p for p in fns[fn].split(", ") if p.strip()]) if fns[fn].strip() else 0
            assert args == want, (name, fn, args, want)
    store_rs = open(os.path.join(src_dir, "store.rs")).read()
    for hook in ("fn hash_ptrs", "fn hash_compact", "fn hash_commitment", "fn hash3", "fn hash4", "fn hash6", "fn hash8", "pub fn hydrate"):
        assert hook in store_rs, hook


def test_parameter_blocks_have_one_layout_in_c_ctypes_and_rust(tmp_path):
    """lurk_hip_ro_params / lurk_hip_ck_params (round 6) cross the ABI by pointer: their size and every field's offset as the header
    compiles must be what lurk_beta_amd/_lib.py declares to ctypes, and rust/lurk-hip-sys/src/ffi.rs must list the same fields."""
    import ctypes
    import shutil
    import subprocess

    from lurk_beta_amd import _lib

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    rs = open(os.path.join(ROOT, "rust", "lurk-hip-sys", "src", "ffi.rs")).read()
    for cname, st in (("lurk_hip_ro_params", _lib.RoParamsStruct), ("lurk_hip_ck_params", _lib.CkParamsStruct)):
        fields = [f[0] for f in st._fields_]
        prog = '#include <stdio.h>\n#include <stddef.h>\n#include "lurk_hip.h"\nint main(void) {\n' + f'  printf("%zu", sizeof({cname}));\n' + \
            "".join(f'  printf(" %zu", offsetof({cname}, {f}));\n' for f in fields) + "  return 0;\n}\n"
        src, exe = tmp_path / f"{cname}.c", tmp_path / cname
        src.write_text(prog)
        subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        nums = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
        assert nums[0] == ctypes.sizeof(st), cname
        assert nums[1:] == [getattr(st, f).offset for f in fields], cname
        block = rs[rs.index(f"pub struct {cname} {{"):]
        block = block[:block.index("}")]
        assert re.findall(r"pub (\w+):", block) == fields, cname
