#!/usr/bin/env python3
"""ISA-weighted issue cost of a gfx950 kernel: basic blocks of the compiled kernel with their VALU instruction counts, priced with the
per-opcode issue cycles MEASURED on the chip (profiles/valu_ceiling_r2.txt: one wave64 VALU instruction occupies its SIMD for 4 cycles,
except the short list that takes 2).  With `--trips label=count,...` (dynamic executions of a block per workgroup) it prints the
weighted VALU cycles per workgroup and the fraction of instructions that are 2-cycle — the denominator `issue_limits_pmc.valu_busy`
(which prices every instruction at 4 cycles) has to be corrected by.

    python tools/isa_weighted.py 'k_fast_cellsILi128ELi64ELb1' [--trips LBB3_7=8,...] [--all]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# 2-cycle opcodes (8 waves / SIMD column of profiles/valu_ceiling_r2.txt >= 0.40): plain 32-bit add / sub, and / or / xor, lshrrev, mov, the
# non-packed 16-bit min / max / add / sub, plain f32 mul / add / fmac; in their e32 / e64 encodings without DPP / SDWA
TWO = re.compile(r"^v_(add|sub|subrev)_(u32|co_u32)(_e32|_e64)?$|^v_(and|or|xor)_b32(_e32|_e64)?$|^v_lshrrev_b32(_e32|_e64)?$|^v_mov_b32(_e32|_e64)?$|"
                 r"^v_(min|max|add|sub)_[ui]16(_e32|_e64)?$|^v_(mul|add|sub|fmac)_f32(_e32|_e64)?$|^v_(max|min)_f16(_e32|_e64)?$")


def main():
    name = sys.argv[1]
    trips = {}
    show_all = "--all" in sys.argv
    for i, a in enumerate(sys.argv):
        if a == "--trips":
            trips = {k: float(v) for k, v in (kv.split("=") for kv in sys.argv[i + 1].split(","))}
    tmp = tempfile.mkdtemp(prefix="orbx_isa_")
    out = os.path.join(tmp, "ext.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-I", os.path.join(ROOT, "orb_slam3_modified_amd", "csrc"), os.path.join(ROOT, "orb_slam3_modified_amd", "csrc", "orbx_extractor.hip"),
                           "-o", out], stderr=subprocess.DEVNULL)
    lines, on = [], False
    for l in open(out):
        if re.match(r"^_Z\S*" + re.escape(name) + r"\S*:", l):
            on = True
        if on:
            lines.append(l.rstrip("\n"))
            if l.startswith(".Lfunc_end"):
                break
    if not lines:
        raise SystemExit(f"no kernel matching {name}")
    blocks, cur = [], ["entry", []]
    for l in lines[1:]:
        m = re.match(r"^(\.?LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1).lstrip("."), []]
            continue
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith(";") and not l.strip().startswith("."):
            cur[1].append((m.group(1), l.strip()))
    blocks.append(cur)
    labels = [b[0] for b in blocks]
    tot4 = tot2 = wcycles = 0.0
    print(f"{'block':12s} {'instr':>6s} {'valu':>6s} {'2-cyc':>6s} {'lds':>5s} {'vmem':>5s} {'salu':>5s}  loop-back / trips")
    for lab, ins in blocks:
        valu = [o for o, _ in ins if o.startswith("v_")]
        two = [o for o in valu if TWO.match(o) and "dpp" not in o and "sdwa" not in o]
        lds = [o for o, _ in ins if o.startswith("ds_")]
        vmem = [o for o, _ in ins if o.startswith(("global_", "buffer_", "flat_"))]
        salu = [o for o, _ in ins if o.startswith("s_")]
        back = [t for o, t in ins if o.startswith("s_cbranch") or o == "s_branch"]
        loops = [t.split()[-1].lstrip(".") for t in back if t.split()[-1].lstrip(".") in labels and labels.index(t.split()[-1].lstrip(".")) <= labels.index(lab)]
        n = trips.get(lab)
        if n is not None:
            tot4 += n * (len(valu) - len(two)); tot2 += n * len(two)
        if show_all or len(valu) >= 20 or loops or n is not None:
            print(f"{lab:12s} {len(ins):6d} {len(valu):6d} {len(two):6d} {len(lds):5d} {len(vmem):5d} {len(salu):5d}  {' '.join(loops)} {'' if n is None else 'x' + str(n)}")
    if trips:
        n = tot4 + tot2
        print(f"\ndynamic VALU wave-instructions per workgroup (given trips): {n:.0f}; 2-cycle: {tot2:.0f} = {100 * tot2 / n:.1f} %")
        print(f"issue cycles per workgroup at 4 / 2 cycles: {4 * tot4 + 2 * tot2:.0f}  (all priced at 4: {4 * n:.0f}; ratio {(4 * tot4 + 2 * tot2) / (4 * n):.3f})")


if __name__ == "__main__":
    main()
