#!/usr/bin/env python3
"""Instruction ledger of the matrix-core decimator's main loop: compiles decim_mfma.hip to assembly, finds the loop that
holds the MFMAs of decim_mfma_kernel<L, PACK16, NG, FR> (one period = 4 x 2^(L-1) steps, a step = 32 raw samples of each of the wave's
8 spans) and counts its instructions by opcode class, per step and per input sample.
usage: python tools/k1m_ledger.py [L=4] [FR=1]      (writes to stdout; profiles/r06_k1m_ledger.txt is its output)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdrdaemon_amd", "csrc")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
FR = int(sys.argv[2]) if len(sys.argv) > 2 else 1

CLASSES = [
    ("MFMA (matrix cores)", lambda op, t: op.startswith("v_mfma")),
    ("v_perm_b32 (limb packs, int16 packs)", lambda op, t: op.startswith("v_perm_b32")),
    ("v_lshl_add_u32 / v_lshl_or / v_add_lshl (Horner recombination of the limb accumulators, addresses)", lambda op, t: op.startswith(("v_lshl_add", "v_lshl_or", "v_add_lshl", "v_lshlrev_b32", "v_lshl_b32"))),
    ("shifts right (v_lshrrev / v_ashrrev / v_bfe: the 19-bit field, >> 13)", lambda op, t: op.startswith(("v_lshrrev", "v_ashrrev", "v_bfe", "v_alignbit"))),
    ("v_xor / v_and / v_or / v_bfi / v_bitop3 (sign flips of the excess-128 limbs, masks)", lambda op, t: op.startswith(("v_xor", "v_and", "v_or", "v_bfi", "v_bitop3", "v_not"))),
    ("DPP / lane moves (I <-> Q exchange, quad hand-over)", lambda op, t: "dpp" in t or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "ds_bpermute", "ds_swizzle", "v_permlane"))),
    ("v_add / v_sub / v_mad / v_mul (addresses, frame-layout bookkeeping)", lambda op, t: op.startswith(("v_add", "v_sub", "v_mad", "v_mul", "v_min", "v_max"))),
    ("v_cmp / v_cndmask (block / frame boundary tests of the frame-layout stores)", lambda op, t: op.startswith(("v_cmp", "v_cndmask"))),
    ("v_mov / v_accvgpr", lambda op, t: op.startswith(("v_mov", "v_accvgpr", "v_pk_mov"))),
    ("other VALU", lambda op, t: op.startswith("v_")),
    ("LDS reads (the LDS-DMA ring's read-back)", lambda op, t: op.startswith("ds_")),
    ("global / buffer loads (LDS-DMA)", lambda op, t: op.startswith(("global_load", "buffer_load", "flat_load"))),
    ("global / buffer stores", lambda op, t: op.startswith(("global_store", "buffer_store", "flat_store"))),
    ("s_waitcnt / s_nop", lambda op, t: op.startswith(("s_waitcnt", "s_nop"))),
    ("other scalar", lambda op, t: op.startswith("s_")),
]


def main():
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form",
                               "--save-temps=obj", "-c", os.path.join(CSRC, "decim_mfma.hip"), "-o", os.path.join(td, "dm.o")], cwd=CSRC, stderr=subprocess.DEVNULL)
        lines = open(os.path.join(td, "decim_mfma-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    want = "17decim_mfma_kernelILi%dELb1ELi4ELb%dEEE" % (L, FR)
    a = next(i for i, l in enumerate(lines) if re.match(r"^_ZN6sdrhip\w+:", l) and want in l)
    b = next(i for i in range(a + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[a:b]
    # the loop: from the label that the LAST backward branch behind the last MFMA jumps to
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    last_mfma = max(i for i, l in enumerate(body) if "v_mfma" in l)
    first_mfma = min(i for i, l in enumerate(body) if "v_mfma" in l)
    loop = None
    for i in range(last_mfma, len(body)):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", body[i])
        if m and m.group(1) in labels and labels[m.group(1)] <= first_mfma:
            loop = (labels[m.group(1)], i)
            break
    assert loop, "loop not found"
    steps = 4 << (L - 1)
    cnt = collections.Counter()
    ops = collections.Counter()
    for l in body[loop[0]:loop[1] + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        for name, f in CLASSES:
            if f(op, t):
                cnt[name] += 1
                break
        ops[op] += 1
    valu = sum(v for k, v in cnt.items() if k.startswith(("v_", "shifts", "DPP", "other VALU")) )
    total = sum(cnt.values())
    print("decim_mfma_kernel<%d, true, 4, %s>: main loop = one period of %d steps (a step = 32 raw samples of each of the wave's 8 spans = 256 input samples)" % (L, "true" if FR else "false", steps))
    print("%d instructions in the loop body, %.2f per step; VALU (without MFMA) %d = %.2f per step = %.2f wave-instructions per 64 input samples... " % (total, total / steps, valu, valu / steps, valu / steps / 4))
    print("per input sample: %.2f VALU lane-ops (64 lanes x VALU per step / 256 samples), x 2^28 samples / 64 = %.1f M VALU wave-instructions per 8 x 2^25 launch (+ warm-up 3 %%, + the VALU pieces)" %
          (valu / steps * 64 / 256, valu / steps / 256 * (1 << 28) / 1e6))
    print()
    print("%-100s %8s %9s" % ("class", "in loop", "per step"))
    for name, _ in CLASSES:
        if cnt[name]:
            print("%-100s %8d %9.2f" % (name, cnt[name], cnt[name] / steps))
    print()
    print("by opcode (>= 8 in the loop): " + ", ".join("%s %d" % kv for kv in ops.most_common() if kv[1] >= 8))


if __name__ == "__main__":
    main()
