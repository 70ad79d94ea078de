#!/usr/bin/env python3
"""Build gate: kernels that pipeline LDS-DMA behind hand-counted `s_waitcnt vmcnt(N)` must not touch scratch --
a register spill is a vector-memory operation, is counted in vmcnt, and silently breaks the count.
Reads hipcc's -Rpass-analysis=kernel-resource-usage output.

k_fwd_brick is held to zero scratch here.  k_fwd_brick_groups and k_bwd_brick keep a cold noinline slow path whose call frame is
scratch OUTSIDE their hot loops; their loops are checked on the device assembly instead (check_loops.py).  k_bwd_brick's one
counted wait, wait_vmcnt(n_at) after the flush atomics, relies on vmcnt retiring in issue order: an extra vector-memory operation
the compiler adds among the wave's youngest could only make that wait cover more, never less."""
import re
import sys

NO_SCRATCH = ("k_fwd_brick",)
EXEMPT = re.compile(r"k_fwd_brick_groups")      # (k_fwd_ws: cold noinline slow path + prologue spills outside its loops; check_loops.py holds the loops)
text = open(sys.argv[1]).read()
bad = []
for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", text, flags=re.S):
    name, scratch = m.group(1), int(m.group(2))
    if any(k in name for k in NO_SCRATCH) and scratch and not EXEMPT.search(name):
        bad.append((name, scratch))
for name, scratch in bad:
    sys.stderr.write("resource check: %s uses %d bytes/lane of scratch (spills break counted vmcnt waits)\n" % (name, scratch))
sys.exit(1 if bad else 0)
