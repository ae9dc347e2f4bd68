"""Fine-grained timeline of ring stages (trace build, -DWNV_FINE_TRACE: stamps are noted in registers and written once per step,
behind everything that is timed).  Slots per (step, stage): 0 X and zin in LDS (barrier passed) | 1 u sent (wave 0) | 2 H sent |
3 skip sent | 4 step done | 5 h_{l-1} received (wave 0) | 6 zin ready (N waves) | 7 barrier behind u passed | 8 poll that carried
every tag of the chain input returned (wave 0) | 9..11 u sent by chain waves 1..3.  Head: 0 input of the step sent | 1 skip sum in
LDS | 2 step done.

    python scripts/fine_trace.py <raw trace written by WNV_RING_TRACE>"""
import sys
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]


class N(int):
    """stamp arithmetic that survives a missing stamp (a role that does not take it): anything with None is None"""


def sub(a, b):
    return None if a is None or b is None else a - b

steps = sorted({int(r[0]) for r in rows})
S = max(int(r[1]) for r in rows)
acc = {}
def get(t, pos):
    r = [x for x in rows if int(x[0]) == t and int(x[1]) == pos][0]
    return [int(x) for x in r[2:]]
for t in steps[1:-1]:
    base = get(t, S)[0]
    st = {}
    for pos in range(0, S):
        raw = get(t, pos)
        v = [x - base if x >= 0 else None for x in raw]
        sends = [x for x in [v[1], v[9], v[10], v[11]] if x is not None]
        if sends:                                   # (position 0 is empty when the head evaluates layer 0)
            st[pos] = dict(v=v, first=min(sends), last=max(sends))
    for pos in range(2, S):
        if pos - 2 not in st:
            continue
        c, pr, pp = st[pos], st[pos - 1], st[pos - 2]
        v = c["v"]
        d = dict(transport=sub(v[8], pr["last"]), zin_ahead_of_hit=sub(v[8], v[6]), lds_barrier=sub(v[0], None if v[8] is None or v[6] is None else max(v[8], v[6])),
                 chain_phase=sub(c["first"], v[0]), wave_skew=c["last"] - c["first"], layer=c["last"] - pr["last"],
                 h_transport=sub(v[5], pp["v"][2]), n_phase=sub(v[6], v[5]), barrier_behind_u=sub(v[7], c["last"]), o_phase=sub(v[2], v[7]), skip_after_h=sub(v[3], v[2]),
                 rest=sub(v[4], None if v[3] is None or v[2] is None else max(v[3], v[2])))
        if len(v) >= 32 and v[12] is not None and v[13] is not None:
            d.update(n_barrier=v[12] - v[5], n_matvec_w4=v[6] - v[12], n_matvec_w5=v[13] - v[12])
        if len(v) >= 32 and v[30] is not None:
            d.update(o_dot_w0=v[14] - v[7], o_recv_w0=v[15] - v[14], o_store_w0=v[2] - v[15],
                     o_dot_w4=v[30] - v[7], o_recv_w4=v[31] - v[30], o_store_w4=v[18] - v[31], h_arrival_vs_bar=(pr["v"][18] if pr["v"][18] is not None else pr["v"][2]) - v[7])
        for k, x in d.items():
            if x is not None:
                acc.setdefault(k, []).append(x)
        if t == steps[2] and None not in (v[8], v[5], v[6], v[0], v[7], v[2], v[3], v[4], d["transport"], d["chain_phase"]):
            print(f"stage {pos:2d}: u[{pos-1}] sent {pr['last']:6d} -> hit {v[8]:6d} (+{d['transport']:4d}) | h recv {v[5]:6d} zin {v[6]:6d} -> in LDS {v[0]:6d} "
                  f"-> u sent {c['first']:6d}..{c['last']:6d} (+{d['chain_phase']:3d}, skew {d['wave_skew']:3d}) -> bar {v[7]:6d} H {v[2]:6d} skip {v[3]:6d} done {v[4]:6d}")
    hd = get(t, S)
    nxt = get(t + 1, S) if t + 1 in steps else None
    last = st[S - 1]
    if nxt and hd[3] >= 0 and hd[4] >= 0:
        d = dict(head_hidden=hd[3] - hd[1], head_out=hd[4] - hd[3], head_sample_send=nxt[0] - hd[4])
        if hd[5] >= 0 and hd[6] >= 0 and hd[7] >= 0:      # the categorical head: partial outputs collected | logits in LDS | class sampled
            d.update(head_collect=hd[5] - hd[4], head_logits_barrier=hd[6] - hd[5], head_sample=hd[7] - hd[6], head_send=nxt[0] - hd[7])
        for k, x in d.items():
            acc.setdefault(k, []).append(x)
    first = st[min(st)]
    e = dict(head_skip_hop=sub(hd[1] - base, last["v"][3]), head_mlp_sample=(nxt[0] - hd[1]) if nxt else None, first_hop=first["v"][8], step=(nxt[0] - hd[0]) if nxt else None)
    for k, x in e.items():
        if x is not None:
            acc.setdefault(k, []).append(x)
if acc:
    print("means (ns) over stages 2..S-1 and", len(steps) - 2, "steps:")
    for k, x in acc.items():
        print(f"  {k:18s} {sum(x) / len(x):8.1f}   (min {min(x)}, max {max(x)})")
