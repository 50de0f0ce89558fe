"""Do the two length buckets always overlap?  HIP maps streams onto a few hardware queues; two streams on one queue run one after the
other.  Times the cfg3 buckets (fp32) with k dummy streams created before the bucket streams, and with a high-priority stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from pepflowww_amd import synth, buckets as bk

dev = torch.device("cuda:0")
model, sd = bench.get_model(dev, "fp32")
wl = bench.WORKLOADS["cfg3"]
batch, B, L, n_real = bench.make_batch(wl, 0)
db = {k: v.to(dev) for k, v in batch.items()}
lens = bk.sample_lengths(batch["res_mask"])
NS, K = 60, 40
noise = {k: v for k, v in synth.make_noise(B, L, 1, seed=7).items() if k != "expo"}
plan = bk.plan_length_buckets(lens)
keep = []
with torch.no_grad():
    s = bk.BucketedSampler(model, plan, B, L, NS, (True, True, True))
    s.bind(db, noise, L, 1, 0)
    s.capture()
    for trial in range(10):
        mode = os.environ.get("MODE", "default")
        if mode == "probe":
            import time as _t
            keep += [torch.cuda.Stream() for _ in range(trial % 5)]
            s._streams = [torch.cuda.Stream() for _ in s.samplers]
            def spin(sts, cyc=2_000_000):
                torch.cuda.synchronize(); t0 = _t.perf_counter()
                for st in sts:
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(cyc)
                torch.cuda.synchronize(); return _t.perf_counter() - t0
            spin(s._streams[:1]); t1 = spin(s._streams[:1]); t2 = spin(s._streams)
            # eager (no graph) run on the two streams
            def timed(use_graph):
                cur = torch.cuda.current_stream()
                torch.cuda.synchronize(); t0 = _t.perf_counter()
                s.run(8, use_graph=use_graph)
                torch.cuda.synchronize(); return (_t.perf_counter() - t0) / 8 * 1e3
            print(f"probe: one spin {t1 * 1e3:.2f} ms, two spins {t2 * 1e3:.2f} ms; eager 8 steps {timed(False):.3f} ms per step; graph 8 steps {timed(True):.3f}", flush=True)
        elif mode == "measured":
            keep += [torch.cuda.Stream() for _ in range(trial % 5)]
            for st in keep[-(trial % 5):] if trial % 5 else []:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=dev)
            bk._STREAMS.clear()
            s._streams = s._choose_streams(True)
            print("   calibration", getattr(s, "calibration", None))
        elif mode == "prio":
            s._streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
        else:
            keep += [torch.cuda.Stream() for _ in range(trial % 5)]          # shifts the pool index of the next two
            s._streams = [torch.cuda.Stream() for _ in s.samplers]
        s.run(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(K)
        torch.cuda.synchronize()
        print(f"{mode} trial {trial}: streams {[hex(st.cuda_stream) for st in s._streams]} {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step", flush=True)
