#!/usr/bin/env python3
"""Build gate: kernels that pipeline LDS-DMA behind hand-counted `s_waitcnt vmcnt(N)` must not touch scratch --
a register spill is a vector-memory operation, is counted in vmcnt, and silently breaks the count.
Reads hipcc's -Rpass-analysis=kernel-resource-usage output."""
import re
import sys

NO_SCRATCH = ("k_fwd_brick",)          # kernels with hand-counted vmcnt waits (k_bwd_brick only uses vmcnt(0))
EXEMPT = re.compile(r"k_fwd_brick_groups")   # its scratch is the frame of the cold noinline slow path; the quad loop is checked on the asm (check_loops.py)
text = open(sys.argv[1]).read()
bad = []
for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", text, flags=re.S):
    name, scratch = m.group(1), int(m.group(2))
    if any(k in name for k in NO_SCRATCH) and scratch and not EXEMPT.search(name):
        bad.append((name, scratch))
for name, scratch in bad:
    sys.stderr.write("resource check: %s uses %d bytes/lane of scratch (spills break counted vmcnt waits)\n" % (name, scratch))
errors = [l for l in text.splitlines() if " error: " in l]
sys.exit(1 if bad else 0)
