import sys
rows = {(int(f[0]), int(f[1])): [int(x) for x in f[2:]] for f in (l.split() for l in open(sys.argv[1]) if not l.startswith("#"))}
L = max(k[1] for k in rows); steps = sorted({k[0] for k in rows}); t = steps[3]
base = rows[(t-1, L)][1]
print("slots: 7 early gathered | 1 zin ready | 0 chain gathered | 2 u published | 3 h published   (ns after the head's send)")
for l in range(L):
    v = [x - base if x >= 0 else None for x in rows[(t, l)]]
    print(f"group {l:2d}: early in {v[7]}  zin {v[1]}  chain in {v[0]}  u pub {v[2]}  h pub {v[3]}")
