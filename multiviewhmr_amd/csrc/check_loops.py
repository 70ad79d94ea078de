#!/usr/bin/env python3
"""Build gate for the brick forward kernels: the quad loops that run behind hand-counted `s_waitcnt vmcnt(N)` (the blocks with the
ds_read_b128 tap reads) must not touch scratch -- the compiler follows a spill reload with `s_waitcnt vmcnt(0)`, which also waits for
the LDS-DMA of the next ring item that was just requested (measured: the prefetch is gone).  k_fwd_brick is held to zero scratch by
check_resources.py; k_fwd_brick_groups keeps a noinline slow path whose call frame is scratch outside the loop, so its loops are
checked here on the device assembly (hipcc -save-temps)."""
import re
import sys

bad = []
for path in sys.argv[1:]:
    name, blk, in_loop, stats = None, None, False, {}
    for line in open(path):
        ls = line.strip()
        m = re.match(r"^(_ZN5mvhmr18k_fwd_brick_groups\S*):", ls)
        if m:
            name = m.group(1)
            continue
        if ls.startswith(".Lfunc_end"):
            name = None
            continue
        if name is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ls)
        if m:
            blk = (name, m.group(1))
            in_loop = "in Loop" in ls or "Loop Header" in ls
            stats[blk] = [in_loop, 0, 0]
            continue
        if blk and ls and not ls.startswith(";"):
            if "ds_read_b128" in ls:
                stats[blk][1] += 1
            if ls.startswith("scratch_"):
                stats[blk][2] += 1
    for (kname, b), (loop, reads, scratch) in stats.items():
        if loop and reads and scratch:
            bad.append((kname, b, scratch))
for kname, b, scratch in bad:
    sys.stderr.write("loop check: %s block %s has %d scratch accesses inside the quad loop\n" % (kname, b, scratch))
sys.exit(1 if bad else 0)
