"""Build-time checks on the gfx950 ISA of the sampling kernels (CPU only: hipcc cross-compiles without a GPU)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="no hipcc")
def test_inline_asm_dpp_reads_keep_their_wait_states(capsys):
    """`dot2c_quad` (csrc/gsamp_dev.h) reads the packed blend weights through a DPP quad broadcast inside inline assembly, where
    hipcc's hazard recogniser cannot see the read: no VALU write of a DPP source register within 2 wait states before any DPP
    instruction of msda.hip, over every path (tools/check_dpp_hazard.py), and the sampler really uses the fused form."""
    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "tools", "check_dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["msda.hip"]) == 0
    line = capsys.readouterr().out.strip().splitlines()[-1]
    n_dpp = int(line.split("kernels,")[1].split("DPP")[0])
    assert n_dpp > 1000, line                 # 64 fused v_dot2c_f32_bf16_dpp per gather batch and kernel instantiation
