"""A GPU call costs minutes; a NameError in a code path only the GPU tests reach costs one.  Every global name a function of the package,
the bench or the entry points loads must exist in its module (or the builtins): checked on the byte code, no GPU needed."""
import builtins
import dis
import importlib
import importlib.util
import os
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODULES = ["lurk_beta_amd." + m[:-3] for m in sorted(os.listdir(os.path.join(ROOT, "lurk_beta_amd"))) if m.endswith(".py") and m != "__init__.py"]
MODULES += ["oracle.coracle", "oracle.pyref", "oracle.spartan_ref", "oracle.spartan_fast", "oracle.keccak_transcript", "oracle.circuit_ref", "oracle.keygen_ref"]
MODULES += ["bench_workloads." + m[:-3] for m in sorted(os.listdir(os.path.join(ROOT, "bench_workloads"))) if m.endswith(".py") and m != "__init__.py"]
FILES = ["bench.py", "__graft_entry__.py"]


def _global_loads(code):
    names = set()
    for ins in dis.get_instructions(code):
        if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME"):
            names.add(ins.argval)
    for c in code.co_consts:
        if isinstance(c, types.CodeType):
            names |= _global_loads(c)
    return names


def _check(mod):
    src = open(mod.__file__).read()
    code = compile(src, mod.__file__, "exec")
    known = set(vars(mod)) | set(vars(builtins)) | {"__class__"}
    # names a module-level function imports locally are bound by STORE_FAST / STORE_NAME inside that function: collect every store too
    def stores(c):
        s = set()
        for ins in dis.get_instructions(c):
            if ins.opname in ("STORE_NAME", "STORE_GLOBAL", "IMPORT_NAME", "STORE_FAST", "STORE_DEREF"):
                s.add(str(ins.argval).split(".")[0])
        for k in c.co_consts:
            if isinstance(k, types.CodeType):
                s |= stores(k)
        return s
    missing = sorted(n for n in _global_loads(code) - known - stores(code))
    assert not missing, f"{mod.__name__}: names loaded but never defined: {missing}"


@pytest.mark.parametrize("name", MODULES)
def test_module_globals_resolve(name):
    _check(importlib.import_module(name))


@pytest.mark.parametrize("fname", FILES)
def test_script_globals_resolve(fname):
    spec = importlib.util.spec_from_file_location("_chk_" + fname.replace(".", "_"), os.path.join(ROOT, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # both files only define functions at import time (main() is guarded)
    _check(mod)


def test_every_run_time_switch_is_documented():
    """INTEGRATION.md section 10 lists the library's and the mirror's LURK_* environment switches: every one the sources read must be
    in that table, and the table must not name one nobody reads."""
    import re

    read = set()
    csrc = os.path.join(ROOT, "lurk_beta_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp", ".cuh")):
            read |= set(re.findall(r'(?:getenv|geti)\("(LURK_[A-Z0-9_]+)"', open(os.path.join(csrc, f)).read()))
    for f in sorted(os.listdir(os.path.join(ROOT, "lurk_beta_amd"))):
        if f.endswith(".py"):
            read |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(LURK_[A-Z0-9_]+)"', open(os.path.join(ROOT, "lurk_beta_amd", f)).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = doc[doc.index("## 10. Run-time switches"):]
    # (constants of the C ABI named in a row's text are not switches)
    listed = set(re.findall(r"`(LURK_[A-Z0-9_]+)`", section)) - {"LURK_MSM_FLAG_AUTO_SLICES", "LURK_MSM_FLAG_SMALL_FORM", "LURK_MSM_SUBMIT_FOLLOW", "LURK_HIP_ERR_OOM"}
    assert read, "no switch found: the patterns above no longer match the sources"
    assert read - listed == set(), f"read by the sources, missing from INTEGRATION.md section 10: {sorted(read - listed)}"
    assert listed - read == set(), f"listed in INTEGRATION.md section 10, read by nobody: {sorted(listed - read)}"
