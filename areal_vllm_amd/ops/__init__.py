"""Hand-written gfx950 (MI355X) kernels and their autograd wrappers."""
from ._lib import available as hip_available  # noqa: F401
from .attention import (  # noqa: F401
    PAGE_SIZE,
    DecodeWorkspace,
    alloc_kv_cache,
    cache_gather,
    context_slots,
    flash_attn_paged_prefill,
    decode_num_splits,
    flash_attn_varlen,
    flash_attn_varlen_qkv,
    flash_attn_varlen_sep,
    paged_attention_decode,
    rope_and_cache_write_,
)
from .logprob import (  # noqa: F401
    gather_logprobs,
    gather_logprobs_entropy,
    lm_head_logprobs,
    logprob_stats,
    merge_shard_stats,
    vocab_shard_range,
)
from .norm_act import RotaryTable, apply_rope, qk_norm, rmsnorm, rope_inplace_, swiglu  # noqa: F401
from .optim import adam_prepare_, adamw_, scale_, sumsq_  # noqa: F401
from .rl import gae_1d_packed, gae_2d, gae_padded, merge_intervals, set_intervals, slice_intervals  # noqa: F401
from .sampling import TokenCounts, advance_step_, decode_advance_, decode_begin_, sample  # noqa: F401
from . import train_gemm  # noqa: F401
from .linear import embedding, linear  # noqa: F401
from .decode_gemm import dgemm, splitk_residual_rmsnorm, splitk_rope_cache_write  # noqa: F401
from .grouped_gemm import grouped_gemm, grouped_gemm_offsets, grouped_gemm_offsets_ok, grouped_gemm_ref  # noqa: F401
from .moe import moe_combine, moe_permute, moe_sort, moe_topk  # noqa: F401
