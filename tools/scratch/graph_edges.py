"""How much does a cross-stream dependency cost inside a replayed HIP graph?  n_ops tiny kernels on the capture stream, with `forks`
fork/side-kernel pairs (side stream waits on main; joined once at the end, or after each fork with --join-each)."""
import argparse, time, torch
ap = argparse.ArgumentParser()
ap.add_argument("--ops", type=int, default=400)
ap.add_argument("--forks", type=int, default=60)
ap.add_argument("--side-ops", type=int, default=3)
ap.add_argument("--join-each", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
x = torch.zeros(4096, device=dev)
y = torch.zeros(4096, device=dev)
side = torch.cuda.Stream()
cap = torch.cuda.Stream()

def body(forks):
    main = torch.cuda.current_stream()
    every = max(a.ops // max(forks, 1), 1)
    nf = 0
    for i in range(a.ops):
        x.add_(1.0)
        if forks and i % every == 0 and nf < forks:
            nf += 1
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(a.side_ops):
                    y.add_(1.0)
            if a.join_each:
                main.wait_stream(side)
    if forks:
        main.wait_stream(side)

for forks in (0, a.forks, a.forks // 4, 4, 1):
    with torch.cuda.stream(cap):
        body(forks)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            body(forks)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    nk = a.ops + (forks * a.side_ops)
    print(f"forks={forks:4d}  kernels={nk:5d}  replay {dt*1e3:8.3f} ms  ({dt*1e6/nk:6.2f} us/kernel)", flush=True)
