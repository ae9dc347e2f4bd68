"""Fine-grained timeline of one ring stage (debug build with -DWNV_FINE_TRACE): slots 5-7."""
import sys
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
steps = sorted({int(r[0]) for r in rows})
t = steps[2]
for pos in (3, 4, 10):
    r = [x for x in rows if int(x[0]) == t and int(x[1]) == pos][0]
    v = [int(x) for x in r[2:]]
    b = v[0]
    print(f"stage {pos}: recv 0 | z FMAs done (wave0) +{v[5]-b} | gate computed (wave0) +{v[6]-b} | wave7 at barrier +{v[7]-b} | barrier passed (wave0) +{v[1]-b} | out/send +{v[2]-b} | skip sent +{v[3]-b}")
