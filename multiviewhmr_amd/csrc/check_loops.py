#!/usr/bin/env python3
"""Build gate on the device assembly (hipcc -save-temps): the hot loops of the brick kernels must not touch scratch.

  k_fwd_brick_groups  quad loops that run behind hand-counted `s_waitcnt vmcnt(N)` (the loops with the ds_read_b128 tap reads): the
                      compiler follows a spill reload with `s_waitcnt vmcnt(0)`, which also waits for the LDS-DMA of the next ring
                      item that was just requested (measured: the prefetch is gone).  Its noinline slow path keeps a scratch call
                      frame outside the loop, so check_resources.py cannot hold the kernel to zero scratch.
  k_bwd_brick         the quad loop with the ds_add_u32 accumulation: same reason (its counted wait_vmcnt(n_at) after the flush
                      atomics assumes the atomics are the wave's youngest vector-memory operations), and the same noinline slow path.

usage: check_loops.py <kernel-name-prefix>:<hot instruction> [...] -- file.s [...]
A loop is every basic block that names the same `in Loop: Header=` (plus the header block itself).  Exits non-zero when a hot loop
touches scratch, and also when a prefix matches no function or a matched prefix has no hot loop at all (a rename must not turn the
gate off silently)."""
import re
import sys


def scan(path, rules):
    """rules: {mangled-name substring: hot instruction} -> (violations, functions seen per rule, hot loops per rule)"""
    bad, seen, hot = [], {k: 0 for k in rules}, {k: 0 for k in rules}
    name, rule, loops, cur, label, fresh = None, None, {}, None, "", False

    def close():
        if name is None:
            return
        for header, (n_hot, n_scratch) in loops.items():
            if n_hot:
                hot[rule] += 1
                if n_scratch:
                    bad.append((name, header, n_scratch))

    for line in open(path):
        ls = line.strip()
        m = re.match(r"^(_Z\S+):", ls)
        if m and not ls.startswith(".L"):
            close()
            name, rule, loops, cur = None, None, {}, None
            for k in rules:
                if k in m.group(1):
                    name, rule = m.group(1), k
                    seen[k] += 1
            continue
        if ls.startswith(".Lfunc_end"):
            close()
            name, rule, loops, cur = None, None, {}, None
            continue
        if name is None:
            continue
        m = re.match(r"^\.L(BB\d+_\d+):", ls) or re.match(r"^; %bb\.(\d+):", ls)
        if m:                                                                    # a new basic block (labelled or fall-through)
            label, cur, fresh = m.group(1), None, True
        if ls.startswith(";") or m:                                              # the loop note sits on the label line or on the comment line after it
            if fresh:
                h = re.search(r"in Loop: Header=(BB\d+_\d+)", ls)
                if h:
                    cur = h.group(1)
                elif "Loop Header" in ls and label.startswith("BB"):
                    cur = label
                if cur is not None:
                    loops.setdefault(cur, [0, 0])
            continue
        fresh = False
        if cur is not None and ls and not ls.startswith(";"):
            if ls.startswith(rules[rule]):
                loops[cur][0] += 1
            if ls.startswith("scratch_"):
                loops[cur][1] += 1
    close()
    return bad, seen, hot


def main(argv):
    if "--" not in argv:
        sys.stderr.write(__doc__)
        return 2
    i = argv.index("--")
    rules = dict(a.split(":", 1) for a in argv[:i])
    files = argv[i + 1:]
    bad, seen, hot = [], {k: 0 for k in rules}, {k: 0 for k in rules}
    for path in files:
        b, s, h = scan(path, rules)
        bad += b
        for k in rules:
            seen[k] += s[k]
            hot[k] += h[k]
    rc = 0
    for kname, header, scratch in bad:
        sys.stderr.write("loop check: %s loop %s has %d scratch accesses beside its %s\n" % (kname, header, scratch, "hot instructions"))
        rc = 1
    for k in rules:
        if not seen[k]:
            sys.stderr.write("loop check: no function matching '%s' in %s -- the gate would be off\n" % (k, " ".join(files)))
            rc = 1
        elif not hot[k]:
            sys.stderr.write("loop check: functions matching '%s' have no loop with %s -- the gate would be off\n" % (k, rules[k]))
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
