"""CPU: AddressSanitizer run of the HOST side of the C-ABI (SURVEY.md section 5 plan): weight packing (BatchNorm folding, LSTM
gate-row permutation, MFMA-fragment order), blob layout and workspace planning are plain host C++ that index into caller
buffers with sizes derived from the dims - exactly what ASan is good at.  gvx_api.hip is compiled with -fsanitize=address for
the host pass only (-fno-gpu-sanitize: GPU ASan is not available on this pool) and linked with the regular kernel objects;
tests/host_asan/driver.cpp feeds it weight tensors that are allocated with EXACTLY the element counts of the state_dict, so
any read past a tensor or write past the blob aborts the run.  No GPU call is made."""
import glob
import os
import subprocess

import numpy as np
import pytest

from genvox_amd import build as gbuild
from genvox_amd.configs import AudioConfig, Tacotron2Config, TextConfig
from genvox_amd.tacotron2 import dims_from_configs
from genvox_amd.weights import state_dict_spec
from tests.golden.cases import SMALL

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _manifest(path, mc, ac, tc):
    d = dims_from_configs(mc, ac, tc)
    items = [(k, int(np.prod(shape)) if len(shape) else 1) for k, (shape, kind, _arg) in state_dict_spec(mc, ac, tc).items()
             if kind != "count"]
    with open(path, "w") as f:
        f.write(" ".join(str(getattr(d, n)) for n, _ in d._fields_) + "\n")
        f.write(f"{len(items)}\n")
        for k, n in items:
            f.write(f"{k} {n}\n")


def test_host_side_under_address_sanitizer(tmp_path):
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not (os.path.exists(gbuild.HIPCC) and os.path.exists(CLANG) and rt):
        pytest.skip("hipcc / clang++ / the ASan runtime are not all present")
    gbuild.build(verbose=False)   # the kernel objects the sanitized library links against
    out = str(tmp_path)
    san = ["-fsanitize=address", "-shared-libsan"]
    api_o = os.path.join(out, "gvx_api.asan.o")
    subprocess.run([gbuild.HIPCC, "-O1", "-g", "-std=c++17", "-fPIC", f"--offload-arch={gbuild.ARCH}", "-ffp-contract=off", *san,
                    "-fno-gpu-sanitize", "-c", os.path.join(gbuild.CSRC, "gvx_api.hip"), "-o", api_o], check=True)
    others = [os.path.join(gbuild.CSRC, s.replace(".hip", ".o")) for s in gbuild.SOURCES if s != "gvx_api.hip"]
    lib = os.path.join(out, "libgenvox_amd_asan.so")
    subprocess.run([gbuild.HIPCC, "-shared", "-fPIC", f"--offload-arch={gbuild.ARCH}", *san, api_o, *others, "-L/opt/rocm/lib", "-lrocfft",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", lib], check=True)
    drv = os.path.join(out, "driver")
    subprocess.run([CLANG, "-O1", "-g", "-std=c++17", *san, "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "tests", "host_asan", "driver.cpp"), "-L", out, "-lgenvox_amd_asan", f"-Wl,-rpath,{out}",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", drv], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(rt[0]) + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    cases = {"full": (Tacotron2Config(), AudioConfig(filter_length=1024), TextConfig(n_tokens=40)),
             "small": (Tacotron2Config(**SMALL["model"]), AudioConfig(filter_length=1024, n_mels=SMALL["n_mels"]), TextConfig(n_tokens=SMALL["n_tokens"]))}
    for name, (mc, ac, tc) in cases.items():
        man = os.path.join(out, name + ".txt")
        _manifest(man, mc, ac, tc)
        r = subprocess.run([drv, man], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok"), f"{name}: rc {r.returncode}\n{r.stdout}\n{r.stderr[-3000:]}"
        assert "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
