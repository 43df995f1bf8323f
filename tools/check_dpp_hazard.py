"""Checks the gfx950 ISA of a csrc file for the one hazard hipcc cannot see in inline assembly: a DPP instruction reads its
src0 through the cross-lane network, and that register needs 2 wait states behind the VALU instruction that wrote it
(every issued instruction is >= 1 wait state, `s_nop N` is N + 1).  hipcc inserts the s_nops for the DPP instructions it
emits itself; `dot2c_quad` (csrc/gsamp_dev.h) is inline asm.  The check follows textual order plus every branch edge into
a label, two instructions deep.

    python tools/check_dpp_hazard.py [msda.hip ...]       -> exit code 1 and a listing if a hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -S --cuda-device-only".split()
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def parse(path):
    """-> {function: [(label | None, mnemonic, [operands])]} of the device assembly"""
    funcs, cur = {}, None
    for line in open(path):
        line = line.split(";")[0].rstrip()
        if not line:
            continue
        if re.match(r"^_Z\w+:", line):
            cur = funcs.setdefault(line[:-1], [])
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            cur.append((m.group(1), None, []))
            continue
        if line.startswith("\t.") or line.startswith(".") or not line.startswith("\t"):
            if line.strip().startswith(".end_amdhsa_kernel") or line.strip().startswith(".section"):
                cur = None
            continue
        parts = line.strip().split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((None, parts[0], ops))
    return funcs


def written_vgprs(mn, ops):
    if not mn or not mn.startswith("v_") or mn.startswith("v_cmp") or not ops:
        return set()
    return regs(ops[0].split()[0])


def check(funcs):
    bad = []
    for fn, ins in funcs.items():
        labels = {lab: i for i, (lab, mn, _) in enumerate(ins) if lab}
        preds = {}       # label index -> indices of the branches that jump there
        for i, (lab, mn, ops) in enumerate(ins):
            if mn and mn.startswith("s_cbranch") or mn == "s_branch":
                tgt = ops[0] if ops else None
                if tgt in labels:
                    preds.setdefault(labels[tgt], []).append(i)

        def back(i, states):
            """all instruction indices within `states` wait states before instruction i (over every path)"""
            out, work = set(), [(i - 1, states)]
            while work:
                j, left = work.pop()
                while j >= 0 and left > 0:
                    lab, mn, ops = ins[j]
                    if lab:
                        for p in preds.get(j, []):
                            work.append((p, left))
                        j -= 1
                        continue
                    out.add(j)
                    left -= (int(ops[0], 0) + 1) if mn == "s_nop" else 1
                    if mn == "s_branch":
                        break
                    j -= 1
            return out

        for i, (lab, mn, ops) in enumerate(ins):
            if not mn or not mn.endswith("_dpp"):
                continue
            src0 = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            for j in back(i, 2):
                if written_vgprs(ins[j][1], ins[j][2]) & src0:
                    bad.append((fn, i, " ".join([mn] + ops), " ".join([ins[j][1]] + ins[j][2])))
    return bad


def main(files):
    rc = 0
    for f in files:
        src = os.path.join(ROOT, "mvgformer_amd", "csrc", f)
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], check=True, stderr=subprocess.DEVNULL)
            funcs = parse(out)
        n_dpp = sum(1 for ins in funcs.values() for _, mn, _ in ins if mn and mn.endswith("_dpp"))
        bad = check(funcs)
        print("%s: %d kernels, %d DPP instructions, %d hazards" % (f, len(funcs), n_dpp, len(bad)))
        for fn, i, a, b in bad[:20]:
            print("  %s @%d: %s   <- %s" % (fn[:60], i, a, b))
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or ["msda.hip"]))
