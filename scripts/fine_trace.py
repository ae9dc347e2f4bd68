"""Fine-grained timeline of ring stages (debug build with -DWNV_FINE_TRACE).  Slots: 0 X received (after LDS + barrier) | 1 u sent
(wave 0) | 2 H sent | 3 skip sent | 4 deferred done | 5 gate value ready (wave 0) | 6 zin ready | 7 barrier after u | 8 poll that
carried every tag returned (wave 0) | 9..15 u sent by waves 1..7."""
import sys
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
steps = sorted({int(r[0]) for r in rows})
S = max(int(r[1]) for r in rows)
W = len(rows[0]) - 2
acc = {}
for t in steps[1:-1]:
    base = [int(x) for x in [r for r in rows if int(r[0]) == t and int(r[1]) == S][0][2:]][0]
    prev = None
    for pos in range(1, S):
        r = [x for x in rows if int(x[0]) == t and int(x[1]) == pos][0]
        v = [int(x) - base for x in r[2:]]
        sends = [v[1]] + ([v[k] for k in range(9, 16)] if W >= 16 else [])
        cur = dict(v=v, first=min(sends), last=max(sends))
        if prev is not None and W >= 16:
            d = dict(transport=v[8] - prev["last"], lds_barrier=v[0] - v[8], matvec_gate=v[5] - v[0], store_issue=v[1] - v[5],
                     wave_skew=cur["last"] - cur["first"], layer=v[1] - prev["v"][1])
            for k, x in d.items():
                acc.setdefault(k, []).append(x)
            if t == steps[2]:
                print(f"stage {pos:2d}: last u of stage {pos-1} sent {prev['last']:6d} -> poll hit {v[8]:6d} (+{d['transport']:4d}) -> in LDS {v[0]:6d} (+{d['lds_barrier']:3d}) "
                      f"-> gate ready {v[5]:6d} (+{d['matvec_gate']:3d}) -> u sent wave0 {v[1]:6d}, waves: first {cur['first']:6d} last {cur['last']:6d} (skew {d['wave_skew']:3d}) | zin ready {v[6]:6d}")
        prev = cur
if acc:
    print("means over stages 2..S-1 and", len(steps) - 2, "steps (ns):", {k: round(sum(x) / len(x), 1) for k, x in acc.items()})
