"""The C-ABI library loads without a GPU and exports every symbol include/gsmvi_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols(headers=("gsmvi_hip.h",)):
    """Every function the given headers declare (gsmvi_hip.h: the boundary; gsmvi_hip_debug.h: diagnostics)."""
    out = set()
    for h in headers:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b(gsmvi_[a-z0-9_]+)\s*\(", txt))
    return sorted(out)


def _nm_dynamic(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())


def _map_symbols(name):
    txt = open(os.path.join(ROOT, "gsm-vi_amd", "csrc", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(re.findall(r"^\s+(gsmvi_[a-z0-9_]+);", txt, flags=re.M))


def test_library_is_built():
    from gsmvi_amd import _lib
    assert os.path.exists(_lib.library_path()), "run __graft_entry__.build() first"


def test_exports_every_declared_symbol():
    from gsmvi_amd import _lib
    lib = _lib.load_library()
    declared = _header_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(_lib.exported_symbols()) == set(declared), "ctypes table out of sync with the header"


def test_product_library_exports_exactly_the_header():
    """-fvisibility=hidden + export list: `nm -D` of the product library is include/gsmvi_hip.h, symbol for symbol -- no
    internal C++ functions, no kernel stubs, no diagnostic entry points."""
    from gsmvi_amd import _lib
    declared = _header_symbols()
    assert _nm_dynamic(_lib.library_path(debug=False)) == declared
    assert _map_symbols("exports.map") == declared


def test_debug_library_adds_only_the_debug_header():
    from gsmvi_amd import _lib
    path = _lib.library_path(debug=True)
    assert os.path.exists(path), "make -C gsm-vi_amd/csrc builds libgsmvi_hip_debug.so beside the product library"
    both = _header_symbols(("gsmvi_hip.h", "gsmvi_hip_debug.h"))
    assert _nm_dynamic(path) == both
    assert _map_symbols("exports_debug.map") == both
    assert set(_lib.exported_symbols(debug=True)) == set(both)
    assert all(n.startswith("gsmvi_debug_") for n in set(both) - set(_header_symbols()))


def test_abi_version_and_status_strings():
    from gsmvi_amd import _lib
    lib = _lib.load_library()
    assert lib.gsmvi_abi_version() == 1
    assert lib.gsmvi_status_string(0) == b"ok"
    assert b"argument" in lib.gsmvi_status_string(1)
    assert lib.gsmvi_workspace_bytes(1024, 32) > 8 * 1024 * 32 * 8
    assert lib.gsmvi_workspace_bytes(0, 1) == 0


def test_workspace_covers_the_b_sized_slabs_of_the_transposed_products():
    """gsmvi_workspace_bytes is a pure function (no GPU): the slab area must hold what the TRANSPOSED panel products leave --
    kc slabs of up to (2B + 8)^2 doubles, kc <= min(8, D / 64) -- not only the (2B + 8) x D slabs of the plain products.  For
    B >> D it did not until round 5 (D = 64, B = 640: a wrong BaM update without a flag; tests/test_gpu_bam.py pins it on the
    GPU, this pins the sizing rule)."""
    from gsmvi_amd import _lib
    lib = _lib.load_library()
    for D in (2, 16, 64, 100, 256, 1024, 4096):
        for B in (1, 8, 32, 128, 300, 640, 1024):
            R = 2 * B + 8
            kct = max(1, min(8, (D + 63) // 64))
            assert lib.gsmvi_workspace_bytes(D, B) >= 8 * (kct * R * R + 8 * R * D), (D, B)
    # monotone in both arguments (the engine regrows by comparing sizes)
    assert lib.gsmvi_workspace_bytes(1024, 64) >= lib.gsmvi_workspace_bytes(1024, 32) >= lib.gsmvi_workspace_bytes(512, 32)


def test_bad_arguments_are_rejected_without_a_gpu():
    from gsmvi_amd import _lib
    lib = _lib.load_library()
    ctx = ctypes.c_void_p()
    assert lib.gsmvi_create(None, 0, 8, 2) == 1
    assert lib.gsmvi_create(ctypes.byref(ctx), 0, -1, 2) == 1
    assert lib.gsmvi_gsm_update_f64(None, None, 4, 2, None, 4, None, 4, None, None, 4, None, None, 4) == 1
    assert b"ctx" in lib.gsmvi_last_error()


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must raise, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import gsmvi_amd
    x = np.zeros((2, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gsmvi_amd.gsm_update(x, x, np.zeros(4), np.eye(4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gsmvi_amd.GSM(4, None, lambda s: s).fit(0, niter=1, verbose=False)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "gsm-vi_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("oracle-backed", "").replace("oracle(test", ""), \
                    f"{f} mentions the oracle"


@pytest.mark.gpu
@pytest.mark.parametrize("D,B", [(64, 8), (1024, 32), (37, 5)])
def test_plain_c_program_drives_the_abi(tmp_path, D, B):
    """tests/abi_c/abi_smoke.c: a C99 program (gcc, no C++, no torch) links libgsmvi_hip.so, passes raw HIP device
    pointers, and reproduces the oracle -- the drop-in boundary is a C ABI, not a Python extension."""
    import subprocess
    import numpy as np
    from oracle import gsm_oracle as orc
    from gsmvi_amd import _lib
    from conftest import rel_err
    libdir = os.path.dirname(_lib.library_path())
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_c", "abi_smoke.c"), "-o", exe, "-L" + libdir, "-lgsmvi_hip",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    st = orc.make_update_state(D, B, 7)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([D, B], dtype=np.int32).tobytes())
        for k in ("samples", "vs", "mu0", "S0"):
            f.write(np.ascontiguousarray(st[k], dtype=np.float64).tobytes())
    p = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    raw = open(tmp_path / "out.bin", "rb").read()
    out = np.frombuffer(raw[:8 * (D + 2 * D * D)], dtype=np.float64)
    info = int(np.frombuffer(raw[8 * (D + 2 * D * D):], dtype=np.int32)[0])
    mu, S, R = out[:D], out[D:D + D * D].reshape(D, D), out[D + D * D:].reshape(D, D)
    mu_o, S_o = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
    assert rel_err(mu, mu_o) < 1e-11 and rel_err(S, S_o) < 1e-11 and info == 0
    assert rel_err(R.T @ R, S_o) < 1e-11 and np.array_equal(np.tril(R, -1), np.zeros_like(R))


@pytest.mark.gpu
@pytest.mark.parametrize("D,B", [(1024, 32), (100, 6), (1024, 128)])      # (1024, 128): BASELINE config 4's shape (world 1 here)
def test_rccl_taking_entry_point_from_plain_c(tmp_path, D, B):
    """tests/abi_c/rccl_sharded.c: a C program owning the ncclComm_t drives gsmvi_gsm_update_sharded_f64,
    gsmvi_gsm_factor_update_sharded_f64 and gsmvi_bam_update_sharded_f64 (local stage -> ncclAllGather on the caller's
    stream -> combined update).  One rank per visible GPU, at most 2 (the GPU box has one; RCCL does not allow two
    ranks on one device)."""
    import subprocess
    import numpy as np
    import torch
    from oracle import gsm_oracle as orc
    from gsmvi_amd import _lib
    from conftest import rel_err
    libdir = os.path.dirname(_lib.library_path())
    exe = str(tmp_path / "rccl_sharded")
    cmd = ["gcc", "-std=gnu99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_c", "rccl_sharded.c"), "-o", exe, "-L" + libdir, "-lgsmvi_hip",
           "-L/opt/rocm/lib", "-lamdhip64", "-lrccl", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    st = orc.make_update_state(D, B, 7)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([D, B], dtype=np.int32).tobytes())
        for k in ("samples", "vs", "mu0", "S0", "Z"):
            f.write(np.ascontiguousarray(st[k], dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(st["L"].T, dtype=np.float64).tobytes())     # F0: S0 = F0^T F0, samples = mu0 + Z F0
    nranks = min(2, torch.cuda.device_count())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    procs = [subprocess.Popen([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(nranks), str(r),
                               str(tmp_path / "nccl.id")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=env) for r in range(nranks)]
    outs = [q.communicate(timeout=300) for q in procs]
    for q, (so, se) in zip(procs, outs):
        assert q.returncode == 0, (q.returncode, se[-2000:])
    mu_o, S_o = orc.gsm_update_faithful(st["samples"], st["vs"], st["mu0"], st["S0"])
    from oracle import bam_oracle as borc
    mu_b, S_b = borc.bam_lowrank_update_exact(st["samples"], st["vs"], st["mu0"], st["S0"], 2.0)
    S_b = 0.5 * (S_b + S_b.T)
    res = []
    n1 = D + D * D
    for r in range(nranks):
        out = np.frombuffer(open(str(tmp_path / "out.bin") + f".{r}", "rb").read(), dtype=np.float64)
        mu, S = out[:D], out[D:n1].reshape(D, D)
        assert rel_err(mu, mu_o) < 1e-11 and rel_err(S, S_o) < 1e-11
        muf, F, fl = out[n1:n1 + D], out[n1 + D:2 * n1].reshape(D, D), out[2 * n1]
        assert fl == 0.0 and rel_err(muf, mu_o) < 1e-10 and rel_err(F.T @ F, S_o) < 1e-10      # factor form = dense update
        mub, Sb = out[2 * n1 + 1:2 * n1 + 1 + D], out[2 * n1 + 1 + D:3 * n1 + 1].reshape(D, D)
        assert rel_err(mub, mu_b) < 1e-7 and rel_err(Sb, S_b) < 1e-7                              # BaM vs the restatement
        o4 = out[3 * n1 + 1:]
        if 2 * B <= D:                                                                             # factor-form BaM = dense BaM
            mubf, Fb, flb = o4[:D], o4[D:n1].reshape(D, D), o4[n1]
            assert flb == 0.0 and rel_err(mubf, mu_b) < 1e-7 and rel_err(Fb.T @ Fb, S_b) < 1e-7
        res.append(out)
    assert all(np.array_equal(res[0], x) for x in res)          # replicas bit-identical


def test_host_code_under_asan_ubsan(tmp_path):
    """the host side of the ABI (validation, workspace arithmetic, error strings, launch
    geometry code paths reachable without a device) built with AddressSanitizer + UBSan (`make -C gsm-vi_amd/csrc asan`)
    and walked by a plain-C driver; any sanitizer report aborts the driver.  (GPU sanitizers are unavailable on the
    pool; the device code of that build is the ordinary one.)"""
    import subprocess
    csrc = os.path.join(ROOT, "gsm-vi_amd", "csrc")
    p = subprocess.run(["make", "-C", csrc, "-j4", "asan"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    libdir = os.path.join(ROOT, "gsm-vi_amd")
    exe = str(tmp_path / "abi_hostcheck")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    cmd = [clang, "-std=c99", "-O1", "-g", "-fsanitize=address,undefined", "-shared-libsan",
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_c", "abi_hostcheck.c"), "-o", exe,
           "-L" + libdir, "-lgsmvi_hip_asan", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               LD_LIBRARY_PATH=os.path.dirname(rt) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "abi_hostcheck ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """Argument validation of the entry points added for the sharded factor path and the draw stream runs before any
    HIP call, so it is checkable on a machine without a GPU."""
    from gsmvi_amd import _lib
    lib = _lib.load_library()
    assert lib.gsmvi_gsm_factor_local_stage_f64(None, None, 8, 2, None, 8, None, 8, None, 8, None, None, 8, None, 24) == 1
    assert b"ctx" in lib.gsmvi_last_error()
    assert lib.gsmvi_gsm_factor_apply_f64(None, None, 8, 2, None, 8, None, 24, None, None, 8, None, None, 8, None, None) == 1
    assert lib.gsmvi_randn_f64(None, None, 1, 0, 16, None, None) == 1
    assert lib.gsmvi_gsm_record_len(5) == 16 and lib.gsmvi_gsm_record_len(4) == 12
    assert lib.gsmvi_gsm_update_sharded_f64(None, None, None, 8, 2, None, 8, None, 8, None, None, 8, None, None, None, 8) == 1
    assert lib.gsmvi_gram_f64(None, None, 8, None, 8, None, 8) == 1
    assert lib.gsmvi_whiten_rows_f64(None, None, 8, 2, None, 8, None, 8, None, None, 8, None) == 1
