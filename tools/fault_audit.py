#!/usr/bin/env python3
"""Host-side audit of round 2's GPU memory access fault: NOT a reproduction attempt.

The fault: bench.py --batch 8, batch-1 leg, phase host_to_detections_latency, three runs of three, when batch 1-3 steps were
captured as SINGLE-BRANCH graphs; fault addresses 0x7eb75b9fe000 / 0x70f927ffe000 / ...fe000 = 8 KiB below a 2 MiB boundary.
This script replays bench's ALLOCATION sequence (source engine, batch-8 engine created, run and closed, batch-1 engine
created, run through the same phases on the shipped two-branch capture) and records:
  (i)   the nodes of the batch-1 step as the forked capture and as the fork-less capture build it (kernel, grid, block, launch
        argument pointers), diffed;
  (ii)  every buffer of the batch-1 handle with [base, end) - which allocations touch the page 8 KiB below a 2 MiB boundary;
  (iii) the same for the pinned staging buffers after the host_to_detections phase has allocated them.
Output: text on stdout (committed as profiles/r03_fault_audit_dump.txt)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-object-detection_amd"))


def flagged(amap):
    out = []
    for ln in amap.splitlines():
        t = ln.split()
        base, end = int(t[3], 16), int(t[5], 16)
        page = 0x1FE000
        # does [base, end) contain, start or end inside the page [2 MiB - 8 KiB, 2 MiB - 4 KiB) of some 2 MiB region?
        first = (base - page + 0x1FFFFF) // 0x200000 * 0x200000 + page if base > page else page
        touches = first < end
        ends_at = (end & 0x1FFFFF) in (0x1FE000, 0x1FF000) or (base & 0x1FFFFF) in (0x1FE000, 0x1FF000)
        if ends_at:
            out.append("BOUNDARY " + ln)
        elif touches and (end - base) < (4 << 20):
            out.append("contains " + ln)
    return out


def main():
    import yolact_amd as ya
    S = 550
    src = ya.Engine(input_size=S, max_batch=1, use_graph=False)
    blob = src.generate_weights(1)
    src.load_weights(blob)
    rng = np.random.default_rng(0)
    # bench's batch-8 leg (abridged: create, load, a few steps incl. host copies, close)
    e8 = ya.Engine(input_size=S, max_batch=8, use_graph=True)
    e8.load_weights_device(src.weights_device_ptr(), src.weights_nbytes())
    f8 = rng.integers(0, 256, (8, S, S, 3), dtype=np.uint8)
    for _ in range(3):
        e8.set_input(f8); e8.evaluate()
    e8.sync(); e8.detections(0)
    print("## batch-8 engine: allocations that touch the page 8 KiB below a 2 MiB boundary (small ones) or start / end on it")
    print("\n".join(flagged(e8.alloc_map())) or "(none)")
    e8.close()
    # the batch-1 leg
    e1 = ya.Engine(input_size=S, max_batch=1, use_graph=True)
    e1.load_weights_device(src.weights_device_ptr(), src.weights_nbytes())
    f1 = f8[:1]
    for _ in range(5):
        e1.set_input(f1); e1.evaluate(); e1.sync(); e1.detections(0, want_masks=False)
    amap = e1.alloc_map()
    print("\n## batch-1 engine: full allocation map after the host_to_detections phase has run (pinned staging allocated)")
    print(amap)
    print("## batch-1 engine: flagged")
    print("\n".join(flagged(amap)) or "(none)")
    forked = e1.graph_nodes(with_tail=True)
    e1.set_tuning(tailfork=0, headfork_maxb=0)
    single = e1.graph_nodes(with_tail=True)
    e1.reset_tuning("tailfork", "headfork_maxb")
    print("\n## (i) batch-1 step, forked capture (shipped)")
    print(forked)
    print("## (i) batch-1 step, fork-less capture (tailfork = 0, headfork_maxb = 0; carries the 4-byte side memset)")
    print(single)
    a, b = forked.splitlines()[1:], single.splitlines()[1:]
    norm = lambda ls: sorted(l.replace("partial ws_side", "partial ws_main") for l in ls if "memset dst side_word" not in l)
    na, nb = norm(a), norm(b)
    print("## (i) diff of the two node lists (split-K workspace of side-stream convolutions and the side memset normalised away)")
    if na == nb:
        print(f"IDENTICAL: {len(na)} nodes each - same kernels, grids, blocks and launch-argument pointers")
    else:
        sa, sb = set(na), set(nb)
        for l in sorted(sa - sb):
            print("only forked:   ", l)
        for l in sorted(sb - sa):
            print("only fork-less:", l)
    # (iii) every pointer in the captured launch arguments lies inside an allocation of THIS handle (none of an earlier one)
    import re
    spans = []
    for ln in e1.alloc_map().splitlines():
        t = ln.split()
        spans.append((int(t[3], 16), int(t[5], 16), t[1]))
    bad, seen = [], 0
    for ln in forked.splitlines():
        for m in re.finditer(r"0x[0-9a-f]{9,}", ln):
            v = int(m.group(0), 16)
            seen += 1
            if not any(b0 <= v < e0 for b0, e0, _ in spans):
                bad.append((hex(v), ln[:80]))
    print(f"## (iii) {seen} pointers in the captured launch arguments; outside every allocation of the batch-1 handle: {len(bad)}")
    for b_ in bad[:20]:
        print("   ", b_)
    e1.close(); src.close()


if __name__ == "__main__":
    main()
