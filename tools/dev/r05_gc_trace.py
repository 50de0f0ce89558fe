"""Which garbage-collector passes run during `python bench.py` and what do they cost?  (per_call 'cold' showed 85 ms stalls in random phases)"""
import gc, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LOG = []
_t = [0.0]
def cb(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        LOG.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3, info["collected"], len(gc.get_objects()) if info["generation"] == 2 else -1))
gc.callbacks.append(cb)
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
bench.main()
g2 = [e for e in LOG if e[0] == 2]
print("collections by generation:", {g: sum(1 for e in LOG if e[0] == g) for g in (0, 1, 2)}, file=sys.stderr)
print("gen-2 passes (ms, collected, tracked objects):", [(round(e[1], 1), e[2], e[3]) for e in g2], file=sys.stderr)
import collections
c = collections.Counter(type(o).__name__ for o in gc.get_objects())
print("most common tracked types:", c.most_common(12), file=sys.stderr)
