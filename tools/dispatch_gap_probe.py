"""How long a chain of N tiny DEPENDENT kernels takes on one stream, issued eagerly (host far ahead)
and replayed from a hipGraph: the per-launch dispatch gap the 3D passes pay ~800 times.
    python tools/dispatch_gap_probe.py
"""
import torch


def main():
    dev = torch.device('cuda', 0)
    x = torch.zeros(4096, device=dev)
    n = 1000

    def chain():
        for _ in range(n):
            x.add_(1.0)

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3 / n

    print('eager : %.2f us per dependent launch (device time, host issues ahead)' % timed(chain))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    print('graph : %.2f us per dependent launch' % timed(g.replay))
    big = torch.zeros(64 << 20, device=dev)

    def chain_big():
        for _ in range(100):
            big.add_(1.0)
    t = timed(chain_big) * n / 100
    print('256 MiB add_: %.1f us per launch (%.0f GB/s)' % (t, 2 * big.numel() * 4 / t / 1e3))


if __name__ == '__main__':
    main()
