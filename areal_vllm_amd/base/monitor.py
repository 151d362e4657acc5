"""Throughput accounting and GPU monitoring for the worker runtime.

* FLOP formulas for decoder-only transformers (training / forward / generation), used by the master
  worker to log TFLOP/s per model function call (``realhf/base/monitor.py:288-358``).
* A kernel-time categoriser that buckets profiler kernel names into compute / collective / p2p / memory /
  misc time; the reference parses CUDA traces (``monitor.py:361-739``), here the input is a rocprofv3
  ``*_kernel_stats.csv`` or ``*_kernel_trace.csv`` and the keys are the ROCm kernel families (Tensile
  ``Cijk_*`` GEMMs, RCCL ``ncclDevKernel*``, ``__amd_rocclr_copyBuffer`` ...).
* A background GPU utilisation / memory monitor built on ``rocm-smi`` instead of NVML.
* Time marks: log lines that delimit named regions so that traces of several workers can be aligned.
"""
import csv
import dataclasses
import enum
import json
import re
import subprocess
import threading
import time
from typing import Dict, Iterable, List, Optional

from ..utils import logging

logger = logging.getLogger("monitor")


# ------------------------------------------------------------------------------- FLOP counts
def calculate_llama_forward_flops(batch_size: int, seqlens: List[int], num_layers: int, hidden_size: int,
                                  intermediate_size: int, vocab_size: int, n_q_heads: Optional[int] = None,
                                  n_kv_heads: Optional[int] = None, head_dim: Optional[int] = None) -> int:
    """Matmul FLOPs of one forward pass over packed sequences. With the head geometry given, the q/k/v/o
    projections are counted exactly for grouped-query attention; otherwise 4 h^2 per token is assumed."""
    assert len(seqlens) == batch_size
    T = sum(seqlens)
    attn_scores = sum(x * x for x in seqlens) * (hidden_size if head_dim is None else n_q_heads * head_dim)
    if head_dim is None:
        proj = 4 * T * hidden_size * hidden_size
    else:
        proj = T * hidden_size * head_dim * (2 * n_q_heads + 2 * n_kv_heads)
    return 2 * num_layers * (proj + 2 * attn_scores + 3 * T * hidden_size * intermediate_size) \
        + 4 * T * vocab_size * hidden_size


def calculate_llama_train_flops(checkpoint_activations_factor: int, batch_size: int, seqlens: List[int], **kw) -> int:
    """factor 3 = forward + backward; 4 with full activation recomputation."""
    return checkpoint_activations_factor * calculate_llama_forward_flops(batch_size, seqlens, **kw)


def calculate_llama_gen_flops(batch_size: int, prompt_lens: List[int], gen_len: int, num_layers: int,
                              hidden_size: int, intermediate_size: int, vocab_size: int) -> int:
    flops = calculate_llama_forward_flops(batch_size, prompt_lens, num_layers, hidden_size, intermediate_size,
                                          vocab_size)
    P = sum(prompt_lens)
    # sum over decode steps i of the per-step cost; the attention term grows linearly with i
    per_step_fixed = 2 * num_layers * (4 * batch_size * hidden_size ** 2 + 3 * batch_size * hidden_size
                                       * intermediate_size) + 4 * batch_size * vocab_size * hidden_size
    attn = 2 * num_layers * 2 * hidden_size * (gen_len * (P + batch_size) + batch_size * gen_len * (gen_len - 1) // 2)
    return flops + gen_len * per_step_fixed + attn


def transformer_forward_flops(n_tokens: int, avg_seqlen: float, num_layers: int, hidden_size: int,
                              intermediate_size: int, vocab_size: int, n_q_heads: Optional[int] = None,
                              n_kv_heads: Optional[int] = None, head_dim: Optional[int] = None, **_) -> float:
    """Forward FLOPs from token counts alone (`n_tokens` tokens in sequences of about `avg_seqlen`)."""
    n = max(1, int(round(n_tokens / max(1.0, avg_seqlen))))
    lens = [int(avg_seqlen)] * n
    return float(calculate_llama_forward_flops(n, lens, num_layers, hidden_size, intermediate_size, vocab_size,
                                               n_q_heads, n_kv_heads, head_dim))


caculuate_llama_forward_flops = calculate_llama_forward_flops  # spelling used by the reference's callers


# ------------------------------------------------------------------- kernel-time categories
class KernelTimeCategory(enum.Enum):
    COMPUTE = "compute"
    P2P_COMM = "p2p_comm"
    COLL_COMM = "coll_comm"
    MEM = "memoryIO"
    IDLE = "idle"
    MISC = "misc"

    @classmethod
    def from_name(cls, name: str) -> "KernelTimeCategory":
        n = name
        if any(k in n for k in ("copyBuffer", "fillBuffer", "Memcpy", "Memset", "memcpy", "memset")):
            return cls.MEM
        if any(k in n for k in ("SendRecv", "sendrecv", "Send", "Recv")) and "nccl" in n.lower():
            return cls.P2P_COMM
        if any(k in n for k in ("ncclDevKernel", "ncclKernel", "rccl", "AllReduce", "AllGather", "ReduceScatter",
                                "Broadcast", "AllToAll")):
            return cls.COLL_COMM
        if any(k in n for k in ("elementwise", "FillFunctor", "reduce_kernel", "index", "gather", "scatter", "cat",
                                "copy_kernel", "Philox", "distribution")):
            return cls.MISC
        return cls.COMPUTE


CUDAKernelTimeCategory = KernelTimeCategory  # name used by the reference's callers


@dataclasses.dataclass
class KernelTimeStat:
    """Per-category kernel time in microseconds."""
    compute: float = 0.0
    p2p_comm: float = 0.0
    coll_comm: float = 0.0
    memoryIO: float = 0.0
    idle: float = 0.0
    misc: float = 0.0

    @property
    def total(self) -> float:
        return self.compute + self.p2p_comm + self.coll_comm + self.memoryIO + self.idle + self.misc

    def percentage(self) -> Dict[str, float]:
        t = max(self.total, 1e-9)
        return {k: 100.0 * v / t for k, v in dataclasses.asdict(self).items()}

    def __add__(self, o: "KernelTimeStat") -> "KernelTimeStat":
        return KernelTimeStat(*[a + b for a, b in zip(dataclasses.astuple(self), dataclasses.astuple(o))])


def kernel_stat_from_rocprof(csv_path: str, wall_time_us: Optional[float] = None) -> KernelTimeStat:
    """Bucket a rocprofv3 kernel stats CSV (columns Name, TotalDurationNs) or kernel trace CSV (columns
    Kernel_Name, Start_Timestamp, End_Timestamp) by category; `idle` = wall time minus kernel time."""
    stat = KernelTimeStat()
    with open(csv_path, newline="") as f:
        for row in csv.DictReader(f):
            if "TotalDurationNs" in row:
                name, us = row["Name"], float(row["TotalDurationNs"]) / 1e3
            else:
                name = row.get("Kernel_Name") or row.get("Name") or ""
                us = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
            key = KernelTimeCategory.from_name(name).value
            setattr(stat, key, getattr(stat, key) + us)
    if wall_time_us is not None:
        stat.idle = max(0.0, wall_time_us - stat.total)
    return stat


# ------------------------------------------------------------------------------ GPU monitor
def read_gpu_stats() -> List[Dict[str, float]]:
    """Utilisation (%) and memory (MB) per visible GPU from `rocm-smi --json`; [] when unavailable."""
    try:
        out = subprocess.run(["rocm-smi", "--showuse", "--showmeminfo", "vram", "--json"], capture_output=True,
                             text=True, timeout=10).stdout
        data = json.loads(out)
    except Exception:
        return []
    stats = []
    for card, d in sorted(data.items()):
        if not card.startswith("card"):
            continue
        use = next((float(v) for k, v in d.items() if "GPU use" in k), 0.0)
        tot = next((float(v) for k, v in d.items() if "Total Memory" in k), 0.0)
        used = next((float(v) for k, v in d.items() if "Total Used" in k), 0.0)
        stats.append(dict(util=use, mem_total_mb=tot / 2 ** 20, mem_used_mb=used / 2 ** 20))
    return stats


def gpu_utilization_monitor(worker_idx: int, interval: float, ttl: float, stop: Optional[threading.Event] = None):
    """Log GPU utilisation every `interval` seconds for `ttl` seconds (run in a daemon thread)."""
    end = time.time() + ttl
    while time.time() < end and not (stop is not None and stop.is_set()):
        for i, s in enumerate(read_gpu_stats()):
            logger.info(f"Worker Index {worker_idx}, GPU {i}: utilization {s['util']:.0f}%, total memory "
                        f"{s['mem_total_mb']:.2f}MB, used memory {s['mem_used_mb']:.2f}MB")
        time.sleep(interval)


# ------------------------------------------------------------------------------- time marks
_MARK = re.compile(r"\*(?P<name>[^*]+)\* #(?P<id>[^#]+)# \$(?P<t>\d+)\$ ns step &(?P<step>\d+)&")


def time_mark(name: str, identifier: str, step: int = 0, t: Optional[int] = None):
    logger.info(f"*{name}* #{identifier}# ${int(time.time_ns() if t is None else t)}$ ns step &{step}&")


def parse_time_mark_in_line(line: str, name: str, step_range=None):
    m = _MARK.search(line)
    if not m or m.group("name") != name:
        return None
    step = int(m.group("step"))
    if step_range is not None and not (step_range[0] <= step < step_range[1]):
        return None
    return m.group("id"), int(m.group("t"))


def parse_time_mark_in_file(path: str, name: str, step_range=None) -> Dict[str, List[int]]:
    out: Dict[str, List[int]] = {}
    with open(path) as f:
        for line in f:
            r = parse_time_mark_in_line(line, name, step_range)
            if r is not None:
                out.setdefault(r[0], []).append(r[1])
    return out
