"""Fine-grained timeline of ring stages (debug build with -DWNV_FINE_TRACE): slots 5-7."""
import sys
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
steps = sorted({int(r[0]) for r in rows})
t = steps[2]
S = max(int(r[1]) for r in rows)
base = [int(x) for x in [r for r in rows if int(r[0]) == t and int(r[1]) == S][0][2:]][0]
for pos in range(1, S):
    r = [x for x in rows if int(x[0]) == t and int(x[1]) == pos][0]
    v = [int(x) - base for x in r[2:]]
    print(f"stage {pos:2d}: zin ready {v[6]:6d} | X received {v[0]:6d} (zin slack {v[0] - v[6]:5d}) | u sent {v[1]:6d} | barrier {v[7]:6d} | H sent {v[2]:6d} | skip sent {v[3]:6d} | deferred done {v[4]:6d}")
