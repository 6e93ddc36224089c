"""Do two branches of a captured graph (fork / join through events on a second stream) run side by side in a hipGraph replay, and
what does the fork cost?  Two ~85 us spin kernels, serial vs forked, 50 replays each.  (MI355X, ROCm 7.0: 176 us serial, 109 us forked:
they do overlap, and the fork / join costs ~20 us -- more than the 10 us the weight-gradient kernel of the flow could hide beside the
summary network's backward: not built.)   usage: python tools/graph_fork_probe.py"""
import torch, time
dev = torch.device("cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.zeros(1, device=dev); b = torch.zeros(1, device=dev)
def body(fork):
    cur = torch.cuda.current_stream()
    if fork:
        e1 = torch.cuda.Event(); e1.record(cur)
        s2.wait_event(e1)
        with torch.cuda.stream(s2):
            torch.cuda._sleep(200_000)      # ~85 us
            e2 = torch.cuda.Event(); e2.record(s2)
        torch.cuda._sleep(200_000)
        cur.wait_event(e2)
    else:
        torch.cuda._sleep(200_000); torch.cuda._sleep(200_000)
    a.add_(1)
for fork in (False, True):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s1):
        body(fork); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s1):
            body(fork)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        print("fork" if fork else "serial", (time.perf_counter() - t0) / 50 * 1e6, "us per replay")
