"""The in-kernel timing probes are build flags of the product sources (csrc/chain.hip -DCHAIN_STAMPS, csrc/msda.hip -DGSAMP_STAMPS, csrc/geom.hip -DTRI_STAMPS:
every workgroup records s_memrealtime / s_memtime stamps and its CU; read on a GPU box by tools/probes/stamps_chain.py /
stamps_gsamp.py / stamps_tri.py).  This keeps the probe builds compiling for gfx950 (hipcc cross-compiles without a GPU) and the product build free
of them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvgformer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,flag,reader", [("chain.hip", "-DCHAIN_STAMPS", "mvg_chain_read_stamps"),
                                             ("msda.hip", "-DGSAMP_STAMPS", "mvg_gsamp_read_stamps"),
                                             ("geom.hip", "-DTRI_STAMPS", "mvg_tri_read_stamps")])
def test_stamps_builds_compile_and_export_their_readers(tmp_path, src, flag, reader):
    obj = tmp_path / "stamps.o"
    subprocess.run([HIPCC] + FLAGS + [flag, "-c", os.path.join(CSRC, src), "-o", str(obj)], check=True, timeout=900)
    syms = subprocess.run(["nm", "-g", str(obj)], check=True, capture_output=True, text=True).stdout
    assert reader in syms
    # the product library carries neither the reader nor the stamp buffer
    lib = os.path.join(ROOT, "mvgformer_amd", "libmvgformer_hip.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", lib], check=True, capture_output=True, text=True).stdout
        assert reader not in syms and "_stamps" not in syms
