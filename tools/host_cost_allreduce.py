"""GPU (one rank, RCCL process group of size 1): host-side cost per step of what bench.py does at N > 1 -- the forward call, the
reducer's async all-reduce -- and whether a HIP graph of (forward + all-reduce) steps captures and replays."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch.distributed as dist
import bench, lsnf_amd
from lsnf_amd import parallel
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
plan = lsnf_amd.prepare([t.to(dev) for t in bench.synth_weights(1)], bench.NZ, bench.WIDTH, bench.DEPTH)
z = torch.randn(8192, bench.NZ, device=dev)
out = (torch.empty_like(z), torch.empty(8192, device=dev), torch.empty(8192, device=dev))
buf = torch.zeros(1, 3, dtype=torch.float64, device=dev)
def wall(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6
st = lsnf_amd.flow.new_stats(dev)
print("host us per call (issue only / incl. drain):")
print("  forward(8192 rows, stats)        %.1f / %.1f" % wall(lambda: lsnf_amd.forward(plan, z, out=out, stats=st)))
works = []
def ar():
    w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True); works.append(w)
    if len(works) > 2: works.pop(0).wait()
print("  all_reduce(async) + wait of -2   %.1f / %.1f" % wall(ar))
class Multi(parallel.PipelinedStatsReducer):
    def _multi(self): return True            # (force the collective path on a world of one)
red = Multi(dev, bucket=1)
def step():
    s = red.next_buffer(); lsnf_amd.forward(plan, z, out=out, stats=s); red.submit(s)
print("  bench step (forward + reducer)   %.1f / %.1f" % wall(step))
red.finish()
# the same step with several evaluations per collective, over three streams with a reducer each (bench.py's eager timed region)
for bucket in (1, 4, 8):
    side = [torch.cuda.Stream() for _ in range(3)]
    reds = [Multi(dev, bucket=bucket) for _ in side]
    outs3 = [(torch.empty_like(z), torch.empty(8192, device=dev), torch.empty(8192, device=dev)) for _ in side]
    cnt = [0]
    def step3():
        k = cnt[0] % 3; cnt[0] += 1
        with torch.cuda.stream(side[k]):
            s = reds[k].next_buffer(); lsnf_amd.forward(plan, z, out=outs3[k], stats=s); reds[k].submit(s)
    print("  3 streams, %d evaluation(s) per collective: %.1f / %.1f" % ((bucket,) + wall(step3)))
    calls = [lsnf_amd.flow.BoundForward(plan, z, o) for o in outs3]
    def step3b():
        k = cnt[0] % 3; cnt[0] += 1
        with torch.cuda.stream(side[k]):
            s = reds[k].next_buffer(); calls[k](s); reds[k].submit(s)
    print("     ... with the launch bound once (flow.BoundForward): %.1f / %.1f" % wall(step3b))
    def step3c():
        k = cnt[0] % 3; cnt[0] += 1
        r = reds[k]
        if r.touches_stream():
            with torch.cuda.stream(side[k]):
                s = r.next_buffer(); calls[k](s); r.submit(s)
        else:
            s = r.next_buffer(); calls[k](s, side[k]); r.submit(s)
    print("     ... and the stream named in the call between a bank's ends: %.1f / %.1f" % wall(step3c))
    for r_, s_ in zip(reds, side):
        with torch.cuda.stream(s_): r_.finish()
    torch.cuda.synchronize()
# graph capture of G steps
try:
    G = 20
    g = torch.cuda.CUDAGraph()
    red2 = Multi(dev, bucket=1)
    s_cap = torch.cuda.Stream()
    with torch.cuda.stream(s_cap):
        for _ in range(3): step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s_cap):
            for _ in range(G):
                s = red2.next_buffer(); lsnf_amd.forward(plan, z, out=out, stats=s); red2.submit(s)
            red2.finish()
    torch.cuda.synchronize()
    t = wall(lambda: g.replay(), 200)
    print("  graph of %d steps: replay %.1f / %.1f us per replay = %.1f us per step; sums %s" % (G, t[0], t[1], t[1] / G, red2.banks[0][0][4:7].tolist()))
except Exception as e:
    print("  graph capture failed:", type(e).__name__, str(e)[:300])
# the same over THREE side streams with a reducer each, as bench.py's eager timed region does (fork from / join to the capture
# stream).  CRASHES the process (segmentation fault inside the capture, RCCL 2.26.6 / ROCm 7.0.2): only with `threestreams`
if "threestreams" not in sys.argv:
    dist.destroy_process_group()
    sys.exit(0)
try:
    G = 30
    side = [torch.cuda.Stream() for _ in range(3)]
    reds = [Multi(dev, bucket=1) for _ in side]
    outs = [(torch.empty_like(z), torch.empty(8192, device=dev), torch.empty(8192, device=dev)) for _ in side]
    def mstep(i):
        k = i % 3
        with torch.cuda.stream(side[k]):
            s = reds[k].next_buffer(); lsnf_amd.forward(plan, z, out=outs[k], stats=s); reds[k].submit(s)
    for i in range(6): mstep(i)
    for r_, s_ in zip(reds, side):
        with torch.cuda.stream(s_): r_.finish()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(); g3 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g3, stream=cap):
            for s_ in side: s_.wait_stream(cap)
            for i in range(G): mstep(i)
            for r_, s_ in zip(reds, side):
                with torch.cuda.stream(s_): r_.finish()
            for s_ in side: cap.wait_stream(s_)
    torch.cuda.synchronize()
    t = wall(lambda: g3.replay(), 200)
    ref = reds[0].banks[0][0][4:7].tolist()
    print("  graph of %d steps over 3 streams: replay %.1f / %.1f us = %.1f us per step; sums %s (rows ok: %s)" % (G, t[0], t[1], t[1] / G, ref, ref[2] == 8192.0))
except Exception as e:
    print("  3-stream graph capture failed:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
