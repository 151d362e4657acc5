"""Device platforms (``areal/platforms/{platform,cuda,cpu}.py`` in the reference)."""
import os

import torch


class Platform:
    device_name: str = "unknown"
    device_type: str = "cpu"
    dispatch_key: str = "CPU"
    communication_backend: str = "gloo"
    device_control_env_var: str = ""

    def __getattr__(self, key):
        mod = getattr(torch, self.device_type, None)
        if mod is not None and hasattr(mod, key):
            return getattr(mod, key)
        raise AttributeError(key)

    def synchronize(self):
        pass

    def current_device(self):
        return torch.device("cpu")

    def set_device(self, idx):
        pass

    def device_count(self) -> int:
        return 0

    def clear_cublas_workspaces(self):
        pass

    def get_inference_worker_class(self) -> str:
        return "areal_vllm_amd.inference.engine.NativeInferenceEngine"

    # name kept for scripts written against the reference's platform API
    get_vllm_worker_class = get_inference_worker_class


class RocmPlatform(Platform):
    device_name = "AMD Instinct"
    device_type = "cuda"  # torch-ROCm exposes HIP devices through the torch.cuda namespace
    dispatch_key = "CUDA"
    communication_backend = "nccl"  # RCCL
    device_control_env_var = "HIP_VISIBLE_DEVICES"

    def synchronize(self):
        torch.cuda.synchronize()

    def current_device(self):
        return torch.device("cuda", torch.cuda.current_device())

    def set_device(self, idx):
        torch.cuda.set_device(idx)

    def device_count(self) -> int:
        return torch.cuda.device_count()

    def memory_allocated(self):
        return torch.cuda.memory_allocated()


class CpuPlatform(Platform):
    device_name = "CPU"
    device_type = "cpu"
    communication_backend = "gloo"
    device_control_env_var = ""

    def __getattr__(self, key):
        raise AttributeError(key)
