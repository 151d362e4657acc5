"""Time per kernel of a captured chain of dependent, trivially small kernels: the floor every extra launch of a
decode layer pays inside a hipGraph.

    python tools/launch_floor_probe.py [--n 2000]
"""
import argparse

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for numel in (1, 256 * 1536, 256 * 8960):
        x = torch.zeros(numel, device=dev, dtype=torch.bfloat16)
        y = torch.ones(numel, device=dev, dtype=torch.bfloat16)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            x.add_(y)
            with torch.cuda.graph(g, stream=side):
                for _ in range(args.n):
                    x.add_(y)
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{numel:9d} bf16 elements, x += y chained {args.n} times in a graph: "
              f"{e0.elapsed_time(e1) * 1e3 / (5 * args.n):6.2f} us per kernel", flush=True)


if __name__ == "__main__":
    main()
