"""The in-kernel timing probe of the chain kernels is a build flag of the product source (csrc/chain.hip, -DCHAIN_STAMPS: every
workgroup records s_memrealtime / s_memtime stamps and its CU; read on a GPU box by tools/probes/stamps_chain.py).  This keeps
the probe build compiling for gfx950 (hipcc cross-compiles without a GPU) and the product build free of it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvgformer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_chain_stamps_build_compiles_and_exports_its_reader(tmp_path):
    obj = tmp_path / "chain_stamps.o"
    subprocess.run([HIPCC] + FLAGS + ["-DCHAIN_STAMPS", "-c", os.path.join(CSRC, "chain.hip"), "-o", str(obj)], check=True, timeout=900)
    syms = subprocess.run(["nm", "-g", str(obj)], check=True, capture_output=True, text=True).stdout
    assert "mvg_chain_read_stamps" in syms
    # the product library carries neither the reader nor the stamp buffer
    lib = os.path.join(ROOT, "mvgformer_amd", "libmvgformer_hip.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", lib], check=True, capture_output=True, text=True).stdout
        assert "mvg_chain_read_stamps" not in syms and "chain_stamps" not in syms
