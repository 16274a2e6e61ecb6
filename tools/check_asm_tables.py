#!/usr/bin/env python3
"""Build-time guard for the hand-placed LDS table loads of the CM256 walks (the loads are started in one asm statement and
awaited in a later one; the hardware does not interlock VGPRs that wait for LDS data, so NOTHING may touch their destination registers
in between).  Compiles gf_kernels.hip and decim_mfma.hip (the fused Rx kernel carries the Karatsuba walk) to assembly and checks, per kernel,
that no instruction reads or writes a register with an asm-issued ds_read outstanding until an asm-placed s_waitcnt covers it.
usage: python tools/check_asm_tables.py   (exit code 1 and the offending lines on a violation)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdrdaemon_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+))\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(asm_path):
    """Linear scan with a merge at basic-block boundaries: a FORWARD branch hands its outstanding loads to its target label (the
    taken path may have skipped a wait the fall-through path executed), the label continues with the union of both; after a
    merge the issue order of the union is unknown, so a partial `lgkmcnt(n > 0)` then drops nothing.  A BACKWARD branch (a loop)
    with a load outstanding is a violation: the scan has passed its target already and cannot follow it."""
    bad, kernels, issues = [], 0, 0
    name, in_asm, pending = None, False, []  # pending: list of (set of registers, line) in issue order, one entry per ds_read
    merged = False       # pending holds loads of more than one path: their relative order is unknown
    carried, seen = {}, set()  # label -> loads outstanding at forward branches to it; labels passed so far in this kernel
    for ln, line in enumerate(open(asm_path), 1):
        t = line.strip()
        if re.match(r"^_Z\w+:", t):
            name, pending, in_asm, merged, carried, seen = t[:-1], [], False, False, {}, set()
            kernels += 1
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            seen.add(m.group(1))
            extra = carried.pop(m.group(1), [])
            if extra:
                have = {l for _, l in pending}
                add = [e for e in extra if e[1] not in have]
                if add:
                    pending = sorted(pending + add, key=lambda e: e[1])
                    merged = True
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch":
            target = t.split()[-1]
            if pending:
                if target in seen:
                    for r, l in pending:
                        bad.append((name, ln, t + "  <backward branch with a table load outstanding>", l))
                else:
                    carried.setdefault(target, []).extend(pending)
            if op == "s_branch":
                pending, merged = [], False  # (what follows is only reachable through its label)
            continue
        if op in ("s_setpc_b64", "s_swappc_b64") and pending:
            for r, l in pending:
                bad.append((name, ln, t + "  <indirect jump with a table load outstanding>", l))
            pending = []
            continue
        if in_asm and op.startswith("ds_read"):
            dst = t.split(",")[0]
            touched = regs(t.split(",", 1)[1]) if "," in t else set()
            for r, l in pending:
                if r & (touched | regs(dst)):
                    bad.append((name, ln, t, l))
            pending.append((regs(dst), ln))
            issues += 1
            continue
        if in_asm and op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                keep = int(m.group(1))
                if keep == 0:
                    pending, merged = [], False
                elif not merged:
                    pending = pending[max(0, len(pending) - keep):]  # (keep > len: nothing is dropped)
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in t:
            pending, merged = [], False  # (the compiler's own full wait covers them too)
            continue
        if pending:
            touched = regs(t)
            for r, l in pending:
                if r & touched:
                    bad.append((name, ln, t, l))
        if op == "s_endpgm":
            pending, merged = [], False
    return kernels, issues, bad


def main():
    rc = 0
    with tempfile.TemporaryDirectory() as td:
        for src in ("gf_kernels.hip", "decim_mfma.hip"):
            subprocess.check_call([HIPCC, "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form", "--save-temps=obj",
                                   "-c", os.path.join(CSRC, src), "-o", os.path.join(td, src + ".o")], cwd=CSRC, stderr=subprocess.DEVNULL)
            asm = os.path.join(td, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")
            kernels, issues, bad = check(asm)
            print("%s: %d kernels, %d asm-issued LDS table loads, %d violations" % (src, kernels, issues, len(bad)))
            for name, ln, t, l in bad[:20]:
                print("  %s line %d: `%s` touches a register whose ds_read (line %d) is not awaited yet" % (name, ln, t, l))
            rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
