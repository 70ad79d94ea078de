#!/usr/bin/env python3
"""Opcode histogram of a kernel's hot loop in the device assembly (hipcc -S --cuda-device-only), priced with the issue costs that
scripts/microbench_ops.hip measures on MI355X (ns per wave instruction and SIMD at 16 waves per CU).

usage: loop_histogram.py file.s <kernel-name substring> <hot instruction prefix>
The hot loop = the loop (blocks sharing one `in Loop: Header=`) with the most hot instructions."""
import collections
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_ashrrev_i32", "v_lshrrev_b32"}
TRANS = {"v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32"}


def main(path, kernel, hotop):
    name, loops, cur, label, fresh = None, {}, None, "", False
    best = None
    for line in open(path):
        ls = line.strip()
        m = re.match(r"^(_Z\S+):", ls)
        if m and not ls.startswith(".L"):
            if name and loops:
                break
            name = m.group(1) if kernel in m.group(1) else None
            loops, cur = {}, None
            continue
        if name is None:
            continue
        if ls.startswith(".Lfunc_end"):
            break
        m = re.match(r"^\.L(BB\d+_\d+):", ls) or re.match(r"^; %bb\.(\d+):", ls)
        if m:
            label, cur, fresh = m.group(1), None, True
        if ls.startswith(";") or m:
            if fresh:
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", ls)
                if h:
                    cur = h.group(1)
                elif "Loop Header" in ls and label.startswith("BB"):
                    cur = label
                if cur is not None:
                    loops.setdefault(cur, collections.Counter())
            continue
        fresh = False
        if cur is not None and ls and not ls.startswith((";", ".")):
            op = ls.split()[0]
            if op.endswith(("_e32", "_e64")):
                op = op[:-4]
            if "row_" in ls or "quad_perm" in ls or "wave_" in ls or " dpp" in ls:
                op += "(dpp)"
            if op.endswith("_dpp"):
                op = op[:-4] + "(dpp)"
            if op.startswith("v_") and re.search(r"[ ,\[](s\d+|s\[\d+:\d+\]|vcc|exec)\b", ls.split(None, 1)[1] if " " in ls else ""):
                op += "(sgpr)"                                                     # an SGPR / vcc operand: issues at the slow rate
            loops[cur][op] += 1
    if not name:
        sys.exit("no kernel matching " + kernel)
    best = max(loops.values(), key=lambda c: sum(n for o, n in c.items() if o.startswith(hotop)))
    valu = {o: n for o, n in best.items() if o.startswith("v_")}
    fast = sum(n for o, n in valu.items() if o in FAST)
    trans = sum(n for o, n in valu.items() if o in TRANS)
    slow = sum(valu.values()) - fast - trans
    print(name)
    for o, n in sorted(best.items(), key=lambda t: -t[1]):
        cls = "" if not o.startswith("v_") else ("fast" if o in FAST else "trans" if o in TRANS else "slow")
        print("  %-34s %5d  %s" % (o, n, cls))
    print("VALU %d = %d fast (1.1 ns) + %d slow (2.0 ns) + %d transcendental (3.55 ns) -> %.0f ns per wave and iteration" %
          (sum(valu.values()), fast, slow, trans, fast * 1.1 + slow * 2.0 + trans * 3.55))
    print("SALU %d, LDS %d, VMEM %d" % (sum(n for o, n in best.items() if o.startswith("s_")), sum(n for o, n in best.items() if o.startswith("ds_")),
                                         sum(n for o, n in best.items() if o.startswith(("buffer_", "global_", "flat_")))))


if __name__ == "__main__":
    main(*sys.argv[1:4])
