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


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="no hipcc")
def test_lds_dma_pieces_of_a_tile_share_one_m0_write(tmp_path):
    """csrc/wreg_gemm.hip issues the four LDS-DMA pieces of a tile from separate asm statements, one per k-step; piece 0 writes M0
    (the LDS base), pieces 1-3 (instruction offsets 1024 / 2048 / 3072) rely on it.  hipcc does not know that: nothing between
    piece 0 and piece 3 of a tile may write M0 in the built ISA."""
    import re
    import subprocess
    out = tmp_path / "wreg.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                    "-ffp-contract=fast", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "mvgformer_amd", "csrc", "wreg_gemm.hip"), "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    lines = [l.split(";")[0].strip() for l in open(out)]
    pieces = [i for i, l in enumerate(lines) if l.startswith("global_load_lds_dwordx4")]
    assert len(pieces) >= 16
    checked = 0
    for i in pieces:
        m = re.search(r"offset:(\d+)", lines[i])
        if not m:
            continue                                   # a piece 0: its statement writes M0 itself
        # back to the previous piece: no write of m0 in between (reads, e.g. as an operand, do not occur either)
        j = i - 1
        while j >= 0 and not lines[j].startswith("global_load_lds_dwordx4"):
            assert not re.search(r"\bm0\b", lines[j]), (lines[j], lines[i])
            j -= 1
        assert j >= 0
        checked += 1
    assert checked >= 6
