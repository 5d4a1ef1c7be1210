"""Histogram of watershed frontier sizes per sweep from TF_WS_TRACE output on stdin (development aid)."""
import sys, collections
per_phase = collections.defaultdict(list)
for line in sys.stdin:
    if not line.startswith("ws_trace"):
        continue
    head, vals = line.split(":")
    phase = int(head.split()[2]); first = int(head.split()[4]) == 0
    v = [int(x) for x in vals.split()]
    v = v[1:-1] if first else v[:-1]          # slot 0 of the first batch is the full scan; last slot repeats as next slot 0
    for x in v:
        if x == 0:
            break
        per_phase[phase].append(x)
edges = [256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 1 << 40]
for ph, v in sorted(per_phase.items()):
    h = [0] * len(edges); tot = [0] * len(edges)
    for x in v:
        for i, e in enumerate(edges):
            if x <= e:
                h[i] += 1; tot[i] += x; break
    print(f"phase {ph}: {len(v)} sweeps, {sum(v)} entries")
    for e, n, t in zip(edges, h, tot):
        if n:
            print(f"   frontier <= {e:>8}: {n:5d} sweeps  {t:>11d} entries")
