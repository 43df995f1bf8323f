"""The in-kernel timing probes (tools/probes/instr_chain_a.py, instr_chain_b.py) patch COPIES of the kernel sources through string
anchors: this keeps the anchors in step with the sources -- each script must still apply and its output must still compile for gfx950
(hipcc cross-compiles without a GPU).  The stamps themselves are read on a GPU box (tools/probes/time_chain_*.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mvgformer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("script,args", [("instr_chain_a.py", []), ("instr_chain_b.py", ["chain.hip", "chain_exp.hip"])])
def test_instrumented_kernel_copies_still_apply_and_compile(tmp_path, script, args):
    work = tmp_path / "mvgformer_amd" / "csrc"
    work.mkdir(parents=True)
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h")):
            shutil.copy(os.path.join(CSRC, f), work / f)
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", script)] + args, cwd=work, check=True)
    src = (work / "chain_exp.hip").read_text()
    assert "__builtin_amdgcn_s_memtime" in src
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize",
                    "-c", "chain_exp.hip", "-o", "chain_exp.o"], cwd=work, check=True, timeout=900)
    assert (work / "chain_exp.o").stat().st_size > 0
