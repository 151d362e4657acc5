"""Peer-memory probe: a parent process exports device buffers of several sizes, a child process maps each and
pulls it with a device-to-device copy; prints open latency, copy bandwidth and whether the bytes match.

    python tools/ipc_probe.py [--sizes-mb 64,1024,3100] [--timeout 40]
"""
import argparse
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch

    from areal_vllm_amd.ops import ipc

    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    for line in sys.stdin:
        desc = json.loads(line)
        t0 = time.perf_counter()
        peer = ipc.PeerBuffer(desc)
        t1 = time.perf_counter()
        dst = torch.empty(desc["nbytes"], dtype=torch.uint8, device="cuda")
        peer.copy_into(dst)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        peer.copy_into(dst)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(json.dumps(dict(open_ms=(t1 - t0) * 1e3, first_copy_ms=(t2 - t1) * 1e3, copy_ms=(t3 - t2) * 1e3,
                              checksum=int(dst[:: max(1, dst.numel() // 65536)].long().sum().item()))), flush=True)
        peer.close()
        del dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--sizes-mb", default="64,1024,2047,2049,3100")
    ap.add_argument("--timeout", type=float, default=40.0)
    args = ap.parse_args()
    if args.child:
        return child()
    import select

    import torch

    from areal_vllm_amd.ops import ipc

    torch.cuda.set_device(0)
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"], stdin=subprocess.PIPE,
                            stdout=subprocess.PIPE, text=True)
    try:
        for mb in [int(x) for x in args.sizes_mb.split(",")]:
            t = torch.randint(0, 255, (mb * 1024 * 1024,), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            exp = ipc.export_buffer(t)
            exp.refresh()
            want = int(t[:: max(1, t.numel() // 65536)].long().sum().item())
            proc.stdin.write(json.dumps(exp.desc) + "\n")
            proc.stdin.flush()
            ready, _, _ = select.select([proc.stdout], [], [], args.timeout)
            if not ready:
                print(f"{mb:6d} MB: NO ANSWER within {args.timeout:.0f} s (pieces {len(exp.desc['chunks'])})", flush=True)
                break
            r = json.loads(proc.stdout.readline())
            print(f"{mb:6d} MB: open {r['open_ms']:8.2f} ms  first copy {r['first_copy_ms']:8.2f} ms  copy {r['copy_ms']:8.2f} ms "
                  f"({mb / 1024 / max(r['copy_ms'], 1e-6) * 1e3:7.1f} GB/s)  match {r['checksum'] == want}  "
                  f"pieces {len(exp.desc['chunks'])}", flush=True)
            del t, exp
    finally:
        proc.kill()


if __name__ == "__main__":
    main()
