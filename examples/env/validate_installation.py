"""Check that this checkout can run: native extensions built and loadable, Python dependencies importable,
GPU visible (optional), a tiny forward/backward on the available device.

    python examples/env/validate_installation.py [--build] [--gpu]

Role of the reference's ``examples/env/validate_installation.py`` (which checks its CUDA wheels); here the
native pieces are this repository's own HIP library and C++ runtime.
"""
import argparse
import importlib
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

REQUIRED = ["torch", "numpy", "yaml", "aiohttp", "requests", "safetensors", "pybind11"]
OPTIONAL = ["transformers", "datasets", "ray", "etcd3", "wandb", "swanlab", "tensorboardX", "fastapi", "uvicorn"]


class Validator:
    def __init__(self):
        self.results = []

    def check(self, name, fn, critical=True):
        try:
            detail = fn()
            self.results.append((name, True, critical, str(detail or "")))
        except Exception as e:  # report everything, decide at the end
            self.results.append((name, False, critical, f"{type(e).__name__}: {e}"))
            if os.environ.get("AREAL_VALIDATE_TRACE"):
                traceback.print_exc()

    def report(self) -> int:
        bad = 0
        for name, ok, critical, detail in self.results:
            tag = "ok  " if ok else ("FAIL" if critical else "skip")
            print(f"[{tag}] {name:34s} {detail}")
            bad += (not ok) and critical
        print(f"\n{'installation looks good' if bad == 0 else f'{bad} critical check(s) failed'}")
        return 1 if bad else 0


def _imp(mod):
    m = importlib.import_module(mod)
    return getattr(m, "__version__", "")


def _native():
    import areal_vllm_amd.ops as ops
    from areal_vllm_amd._C import _runtime

    assert ops.hip_available(), "libareal_hip.so not loadable (run `python build_ext.py`)"
    assert hasattr(_runtime, "Scheduler")
    return "libareal_hip.so + _runtime"


def _scheduler():
    from areal_vllm_amd._C import _runtime

    sc = _runtime.SchedulerConfig()
    sc.num_pages, sc.max_running_requests, sc.max_prefill_tokens, sc.context_length = 64, 4, 1024, 256
    s = _runtime.Scheduler(sc)
    s.add_request("r0", [5, 6, 7], 4, 0, 1.0, 1.0, 1 << 30, [])
    assert s.schedule()["kind"] == 1
    return "prefill plan produced"


def _tiny_step(device):
    import torch

    from areal_vllm_amd.models.config import preset
    from areal_vllm_amd.models.transformer import Transformer

    dt = torch.bfloat16 if device == "cuda" else torch.float32
    m = Transformer(preset("tiny"), dtype=dt, device=device)
    m.init_weights(seed=0)
    ids = torch.randint(2, 500, (48,), device=device)
    pos = torch.arange(48, device=device, dtype=torch.int32)
    cu = torch.tensor([0, 48], device=device, dtype=torch.int32)
    h = m(ids, pos, cu, 48)
    lp, _ = m.logprobs(h, torch.roll(ids, -1))
    (-lp.mean()).backward()
    g = sum(float(p.grad.float().norm()) for p in m.parameters() if p.grad is not None)
    assert g == g and g > 0
    return f"loss {-float(lp.detach().mean()):.3f}, grad norm sum {g:.3f} on {device}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true", help="compile the native extensions first")
    ap.add_argument("--gpu", action="store_true", help="require an MI355X and run the GPU step")
    args = ap.parse_args()
    v = Validator()
    if args.build:
        v.check("build native extensions", lambda: __import__("build_ext").build_all(verbose=False))
    for m in REQUIRED:
        v.check(f"import {m}", lambda m=m: _imp(m))
    for m in OPTIONAL:
        v.check(f"import {m} (optional)", lambda m=m: _imp(m), critical=False)
    v.check("native extensions", _native)
    v.check("C++ scheduler", _scheduler)
    v.check("tiny forward/backward (cpu)", lambda: _tiny_step("cpu"))
    import torch

    if args.gpu or torch.cuda.is_available():
        v.check("GPU visible", lambda: torch.cuda.get_device_name(0), critical=args.gpu)
        v.check("tiny forward/backward (gpu)", lambda: _tiny_step("cuda"), critical=args.gpu)
    sys.exit(v.report())


if __name__ == "__main__":
    main()
