"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/vllm_rs_amd.h declares (with the seven names the reference imports, src/utils/gptq.rs:3-6), the
ctypes mirror agrees with the C struct layout, and the product path refuses to run without a device."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vllm_rs_amd import _lib  # noqa: E402

REFERENCE_FFI = ["awq_repack", "gemm_half_q_half_alt", "gptq_repack", "marlin_4bit_bf16", "marlin_4bit_f16",
                 "marlin_awq_4bit_bf16", "marlin_awq_4bit_f16"]  # src/utils/gptq.rs:3-6


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) > 90, "header parse found too few prototypes"
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/vllm_rs_amd.h but not exported: {missing}"
    assert not _lib.MISSING, f"bound by _lib.py but not exported: {_lib.MISSING}"


def test_reference_ffi_symbols_present():
    lib = _lib.load()
    names = set(_lib.declared_symbols())
    for n in REFERENCE_FFI:
        assert n in names, f"{n} not declared in the header"
        assert hasattr(lib, n), f"{n} not exported"


def test_exported_symbols_are_plain_c():
    """extern "C", no C++ mangling and no torch types on the boundary."""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split()[-2:-1] == ["T"]}
    for n in _lib.declared_symbols():
        assert n in exported, f"{n} is not a defined text symbol"
    hdr = open(_lib.HEADER_PATH).read()
    assert "torch" not in hdr.lower() and "at::" not in hdr and "std::" not in hdr


def test_product_library_holds_no_experiment_kernels():
    """the kernels that lost their A/B (experiments/: persistent decode step, fused q/k/v + attention launch) are neither
    compiled into nor exported by the library a reference-side caller links"""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    bad = [l.split()[-1] for l in out.splitlines() if re.search(r"decode_step|qkv_attn", l)]
    assert not bad, f"experiment symbols in the product library: {bad[:5]}"
    srcs = os.listdir(os.path.join(ROOT, "vllm_rs_amd", "csrc"))
    assert not [f for f in srcs if f.startswith(("decode_step", "qkv_attn"))]


def test_struct_layout_matches_header(tmp_path):
    """sizeof/offsetof of the config structs as gcc sees the header == the ctypes mirror."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vllm_rs_amd.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(vra_model_config), sizeof(vra_engine_config),'
                   ' offsetof(vra_model_config, rope_theta), offsetof(vra_model_config, dtype),'
                   ' offsetof(vra_engine_config, seed), sizeof(vra_step_meta));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(_lib.ModelConfig), C.sizeof(_lib.EngineConfig), _lib.ModelConfig.rope_theta.offset,
            _lib.ModelConfig.dtype.offset, _lib.EngineConfig.seed.offset, C.sizeof(_lib.StepMeta)]
    assert got == want


def test_header_compiles_as_c_and_cxx(tmp_path):
    for comp, ext in (("gcc", "c"), ("g++", "cpp")):
        f = tmp_path / f"inc.{ext}"
        f.write_text('#include "vllm_rs_amd.h"\nint main(void){return 0;}\n')
        subprocess.run([comp, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(f), "-o", str(tmp_path / f"inc_{ext}.o")], check=True)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under vllm_rs_amd/ may import, link or execute it."""
    pat = re.compile(r"\boracle\b")
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "vllm_rs_amd")):
        if "build" in d:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".cuh")) or f == "Makefile":
                for i, line in enumerate(open(os.path.join(d, f), errors="ignore"), 1):
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    if pat.search(code) and ("import" in code or "include" in code or "dlopen" in code or "-l" in code):
                        bad.append(f"{f}:{i}: {line.strip()}")
    assert not bad, bad
    nm = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "vra_oracle" not in nm


def test_engine_fails_loudly_without_gpu():
    lib = _lib.load()
    if lib.vra_device_count() > 0:
        pytest.skip("a GPU is visible")
    from vllm_rs_amd import engine as E
    with pytest.raises(RuntimeError, match="no HIP device"):
        E.Engine(dict(E.TINYLLAMA))


def test_missing_library_is_an_error(tmp_path):
    code = ("import os,sys; sys.path.insert(0, %r); os.environ['VRA_LIB']=%r\n"
            "from vllm_rs_amd import _lib\n"
            "try:\n    _lib.load()\nexcept (OSError, RuntimeError) as e:\n    print('RAISED'); sys.exit(0)\nsys.exit(1)\n") % (ROOT, str(tmp_path / "nope.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "RAISED" in r.stdout, r.stdout + r.stderr
