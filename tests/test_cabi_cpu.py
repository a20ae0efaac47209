"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/dlc.h declares; host-only entry points work; the
product path fails loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from deeploopcloser_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "deeploopcloser_amd", "csrc"), "-j4"])
    return _lib.load()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "dlc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dlc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_python_binds(lib):
    from deeploopcloser_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_host_only_entry_points(lib):
    from deeploopcloser_amd import _lib
    assert lib.dlc_abi_version() == _lib.DLC_ABI_VERSION
    header = open(os.path.join(ROOT, "include", "dlc.h")).read()
    assert "#define DLC_ABI_VERSION %d\n" % _lib.DLC_ABI_VERSION in header
    assert lib.dlc_status_string(0) == b"ok"
    assert lib.dlc_status_string(_lib.DLC_ERR_WORKSPACE) == b"workspace too small"
    # workspace sizes are pure host arithmetic
    w = lib.dlc_cosine_topk_workspace_bytes(256, 1_000_000, 4096, 20)
    assert w > 256 * (1_000_000 // 16) * 4 and w % 256 == 0
    assert lib.dlc_cosine_topk_workspace_bytes(256, 1000, 4096, 0) == 0
    assert lib.dlc_cosine_topk_workspace_bytes(256, 1000, 4096, _lib.DLC_MAX_K + 1) == 0
    dims = (C.c_int64 * 6)(1681, 2500, 2500, 2500, 2500, 2500)
    assert lib.dlc_sdav_encode_workspace_bytes(300, dims, 5, _lib.DLC_F64) >= 2 * 300 * 2500 * 8
    assert lib.dlc_sdav_similarity_workspace_bytes(20, 30, 2500, 0, 0) > 0
    # the similarity's two forms (include/dlc.h): the int8 arg-min filter at the reference's shape -- quantised panels twice,
    # the arg-mins the product kernel emits, 0.9 GB -- and the fp64 Gram form (transpose + fp64 block, 8.7 GB) for P > 32 or on request
    w_i8 = lib.dlc_sdav_similarity_workspace_bytes(1063, 30, 2500, 0, 0)
    w_f64 = lib.dlc_sdav_similarity_workspace_bytes(1063, 30, 2500, _lib.DLC_SIM_FORCE_F64, 0)
    assert 0.8e9 < w_i8 < 1.0e9 and 8.5e9 < w_f64 < 9.0e9          # r03: the int32 product block (4 GB) and the second panel are gone
    assert lib.dlc_sdav_similarity_workspace_bytes(1063, 30, 2500, _lib.DLC_SIM_NO_HOST_SYNC, 0) == w_i8
    rows = 300 * 40
    assert lib.dlc_sdav_similarity_workspace_bytes(300, 40, 64, 0, 0) >= rows * rows * 8 // 2      # 40 patches: fp64 form
    assert lib.dlc_sdav_similarity_workspace_bytes(1063, 30, 2500, _lib.DLC_SIM_FORCE_F64, 1 << 20) < 6.0e9   # fp64 form: one frame per chunk
    assert lib.dlc_last_error(None) == b"null context"


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_fails_loudly(lib):
    import deeploopcloser_amd as dlc
    ctx = C.c_void_p()
    assert lib.dlc_create(0, C.byref(ctx)) < 0 and not ctx.value
    for make in (lambda: dlc.SDAV(), lambda: dlc.CnnVtl(), lambda: dlc.DistanceCalculator.calculate_distance([1], [2]),
                 lambda: dlc.match_topk([[1.0]], [[1.0]], 1)):
        with pytest.raises(RuntimeError):
            make()


def test_kernel_register_contracts_hold_in_the_object_code(lib):
    """gram_i8_kernel relies on hipcc leaving two things alone: M0 around its LDS-DMA pieces and the fixed accumulation
    registers a0..a191 around its inline-asm MFMAs (an accumulator hipcc moved on its own would carry no wait states behind
    the MFMA that wrote it, docs/LAB.md 9.2).  `make` runs csrc/check_m0.py on the object; so does this test."""
    csrc = os.path.join(ROOT, "deeploopcloser_amd", "csrc")
    obj = os.path.join(csrc, "build", "gram_i8.o")
    if not os.path.exists(obj):
        subprocess.check_call(["make", "-C", csrc, "-j4"])
    out = subprocess.run(["python3", os.path.join(csrc, "check_m0.py"), obj], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "192 zero writes" in out.stdout and "no other M0 access" in out.stdout


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deeploopcloser_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_math_utils_matches_golden(golden):
    from deeploopcloser_amd import MathUtils
    g = golden("mathutils.npz")
    got = [MathUtils.compressed_size(int(v), float(g["compression"])) for v in g["values"]]
    assert got == g["sizes"].tolist()


def _build_c_demo(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.join(ROOT, "deeploopcloser_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", "-L" + lib_dir, "-ldlc_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_header_is_plain_c_and_demo_links(lib, tmp_path):
    """include/dlc.h compiles as C11 (gcc -Wall -Werror) and a C program links against the library;
    without a GPU it stops at dlc_create, loudly."""
    exe = _build_c_demo(tmp_path)
    if not torch.cuda.is_available():
        res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert res.returncode == 1 and "no MI355X visible" in res.stderr


@pytest.mark.gpu
def test_c_demo_runs_on_the_gpu(lib, tmp_path):
    res = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "c_abi_demo ok" in res.stdout, res.stdout + res.stderr
