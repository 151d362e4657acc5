"""Vision side of Qwen2-VL style vision-language models: the ViT encoder with 2-D rotary attention and the
2x2 patch merger that projects into the language model's hidden size, plus the multimodal rotary
position ids (M-RoPE) of the language model.

The reference trains VLMs through Hugging Face model classes inside its FSDP engine
(``areal/engine/base_hf_engine.py:135-195`` loads ``AutoModelForImageTextToText``; the vision workflow is
``areal/workflow/vision_rlvr.py``, the Ulysses patch ``areal/models/transformers/qwen2_vl.py``). Here the
tower is a native module whose parameter names equal the checkpoint names below ``visual.``, so HF
checkpoints load by name; its parameters live in the same flat buffer as the language model, i.e. they
are trained, reduced and shipped to the rollout engine with everything else.

Input convention (what the HF image processor emits): ``pixel_values`` [N_patches, C * tp * P * P] with the
patches of every image already in 2x2 merge-window order and ``image_grid_thw`` [n_images, 3] = frames,
patch rows, patch columns. Attention is full (not causal) inside each frame. The encoder is a small part
of the model's FLOPs (head dim 80 for the released sizes, which the D=128 MFMA attention kernel does not
cover), so it runs on PyTorch's fused attention.
"""
import dataclasses
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclasses.dataclass
class VisionConfig:
    depth: int = 32
    embed_dim: int = 1280
    hidden_size: int = 1536  # output size = hidden size of the language model
    num_heads: int = 16
    mlp_ratio: float = 4.0
    patch_size: int = 14
    spatial_merge_size: int = 2
    temporal_patch_size: int = 2
    in_channels: int = 3

    @classmethod
    def from_dict(cls, d: Dict) -> "VisionConfig":
        known = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in known})

    def to_dict(self) -> Dict:
        return dict(dataclasses.asdict(self), model_type="qwen2_vl_vision", hidden_act="quick_gelu")


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


class _Linear(nn.Module):
    def __init__(self, i, o, dtype, device):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i, dtype=dtype, device=device))
        self.bias = nn.Parameter(torch.zeros(o, dtype=dtype, device=device))

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class _LayerNorm(nn.Module):
    def __init__(self, n, dtype, device, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n, dtype=dtype, device=device))
        self.bias = nn.Parameter(torch.zeros(n, dtype=dtype, device=device))
        self.eps = eps

    def forward(self, x):
        return F.layer_norm(x.float(), (x.shape[-1],), self.weight.float(), self.bias.float(), self.eps).to(x.dtype)


class _VisionAttention(nn.Module):
    def __init__(self, cfg: VisionConfig, dtype, device):
        super().__init__()
        self.h = cfg.num_heads
        self.qkv = _Linear(cfg.embed_dim, 3 * cfg.embed_dim, dtype, device)
        self.proj = _Linear(cfg.embed_dim, cfg.embed_dim, dtype, device)

    def forward(self, x, cos, sin, frame_lens: List[int]):
        N, E = x.shape
        q, k, v = self.qkv(x).view(N, 3, self.h, E // self.h).unbind(1)
        q = (q.float() * cos + _rotate_half(q.float()) * sin).to(x.dtype)
        k = (k.float() * cos + _rotate_half(k.float()) * sin).to(x.dtype)
        outs, o = [], 0
        for n in frame_lens:  # full attention inside a frame, nothing across frames
            qs, ks, vs = (t[o:o + n].transpose(0, 1).unsqueeze(0) for t in (q, k, v))
            outs.append(F.scaled_dot_product_attention(qs, ks, vs)[0].transpose(0, 1))
            o += n
        return self.proj(torch.cat(outs, 0).reshape(N, E))


class _VisionMLP(nn.Module):
    def __init__(self, cfg: VisionConfig, dtype, device):
        super().__init__()
        self.fc1 = _Linear(cfg.embed_dim, int(cfg.embed_dim * cfg.mlp_ratio), dtype, device)
        self.fc2 = _Linear(int(cfg.embed_dim * cfg.mlp_ratio), cfg.embed_dim, dtype, device)

    def forward(self, x):
        h = self.fc1(x)
        return self.fc2(h * torch.sigmoid(1.702 * h))  # QuickGELU


class _VisionBlock(nn.Module):
    def __init__(self, cfg, dtype, device):
        super().__init__()
        self.norm1 = _LayerNorm(cfg.embed_dim, dtype, device)
        self.norm2 = _LayerNorm(cfg.embed_dim, dtype, device)
        self.attn = _VisionAttention(cfg, dtype, device)
        self.mlp = _VisionMLP(cfg, dtype, device)

    def forward(self, x, cos, sin, frame_lens):
        x = x + self.attn(self.norm1(x), cos, sin, frame_lens)
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, cfg: VisionConfig, dtype, device):
        super().__init__()
        self.proj = nn.Module()
        # stored in the checkpoint's Conv3d shape; kernel = stride, so it is a linear map of a flat patch
        self.proj.weight = nn.Parameter(torch.empty(cfg.embed_dim, cfg.in_channels, cfg.temporal_patch_size,
                                                    cfg.patch_size, cfg.patch_size, dtype=dtype, device=device))

    def forward(self, pixel_values):
        w = self.proj.weight
        return F.linear(pixel_values.to(w.dtype), w.view(w.shape[0], -1))


class _Merger(nn.Module):
    def __init__(self, cfg: VisionConfig, dtype, device):
        super().__init__()
        m = cfg.spatial_merge_size ** 2
        self.m = m
        self.ln_q = _LayerNorm(cfg.embed_dim, dtype, device)
        self.mlp = nn.ModuleList([_Linear(cfg.embed_dim * m, cfg.embed_dim * m, dtype, device), nn.GELU(),
                                  _Linear(cfg.embed_dim * m, cfg.hidden_size, dtype, device)])

    def forward(self, x):
        x = self.ln_q(x).view(-1, x.shape[-1] * self.m)
        for layer in self.mlp:
            x = layer(x)
        return x


class VisionTower(nn.Module):
    def __init__(self, cfg: VisionConfig, dtype=torch.bfloat16, device="cpu"):
        super().__init__()
        self.cfg = cfg
        self.patch_embed = _PatchEmbed(cfg, dtype, device)
        self.blocks = nn.ModuleList([_VisionBlock(cfg, dtype, device) for _ in range(cfg.depth)])
        self.merger = _Merger(cfg, dtype, device)

    def rotary(self, grid_thw: torch.Tensor, device) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos / sin [N, 1, head_dim]: first half of the rotary pairs follows the patch row, second half
        the patch column, both enumerated in merge-window order."""
        cfg = self.cfg
        hd = cfg.embed_dim // cfg.num_heads
        m = cfg.spatial_merge_size
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32, device=device) / (hd // 2)))
        parts = []
        for t, h, w in grid_thw.tolist():
            hp = torch.arange(h, device=device).view(h, 1).expand(h, w)
            wp = torch.arange(w, device=device).view(1, w).expand(h, w)
            order = lambda p: p.reshape(h // m, m, w // m, m).permute(0, 2, 1, 3).reshape(-1)  # noqa: E731
            pos = torch.stack([order(hp), order(wp)], -1).repeat(t, 1).float()  # [t*h*w, 2]
            parts.append((pos.unsqueeze(-1) * inv).flatten(1))  # [n, hd/2]
        ang = torch.cat(parts, 0)
        emb = torch.cat([ang, ang], -1).unsqueeze(1)
        return emb.cos(), emb.sin()

    def forward(self, pixel_values: torch.Tensor, grid_thw: torch.Tensor) -> torch.Tensor:
        """-> [N_patches / merge^2, hidden_size] image embeddings in placeholder-token order."""
        x = self.patch_embed(pixel_values)
        cos, sin = self.rotary(grid_thw, x.device)
        frame_lens = [h * w for t, h, w in grid_thw.tolist() for _ in range(t)]
        for blk in self.blocks:
            x = blk(x, cos, sin, frame_lens)
        return self.merger(x)

    def n_output_tokens(self, grid_thw: torch.Tensor) -> int:
        return int((grid_thw[:, 0] * grid_thw[:, 1] * grid_thw[:, 2]).sum()) // self.cfg.spatial_merge_size ** 2


# ------------------------------------------------------------------------------------ M-RoPE
def mrope_position_ids(input_ids: Sequence[int], grids: Sequence[Sequence[int]], image_token_ids: Sequence[int],
                       spatial_merge_size: int, return_mask: bool = False):
    """3-D rotary positions [3, S] (temporal, row, column) of one sequence and its position delta
    (``max position + 1 - S``: what generated text tokens add to their sequence index).

    Text tokens count up on all three axes; every run of image placeholder tokens takes the next grid
    from `grids` and is numbered by (frame, merged row, merged column) relative to the running position,
    which then advances by max(rows, columns) of the merged grid. Placeholder ids that appear after all
    grids are used up (a policy can sample them) are ordinary text. With `return_mask` the boolean mask of
    the positions that take image features is returned as a third value."""
    ids = list(input_ids)
    S = len(ids)
    img = set(int(x) for x in image_token_ids)
    out = torch.empty(3, S, dtype=torch.long)
    mask = torch.zeros(S, dtype=torch.bool)
    cur, i, g = 0, 0, 0
    while i < S:
        j = i
        if ids[i] in img and g < len(grids):
            t, h, w = (int(x) for x in grids[g])
            g += 1
            hh, ww = h // spatial_merge_size, w // spatial_merge_size
            n = t * hh * ww
            assert all(x in img for x in ids[i:i + n]) and i + n <= S, "image placeholders do not match image_grid_thw"
            tt = torch.arange(t).view(t, 1, 1).expand(t, hh, ww).reshape(-1)
            hp = torch.arange(hh).view(1, hh, 1).expand(t, hh, ww).reshape(-1)
            wp = torch.arange(ww).view(1, 1, ww).expand(t, hh, ww).reshape(-1)
            out[:, i:i + n] = torch.stack([tt, hp, wp]) + cur
            mask[i:i + n] = True
            cur += max(hh, ww)
            i += n
        else:
            j += 1
            while j < S and not (ids[j] in img and g < len(grids)):
                j += 1
            out[:, i:j] = torch.arange(j - i).view(1, -1) + cur
            cur += j - i
            i = j
    delta = int(out.max()) + 1 - S if S else 0
    if return_mask:
        return out, delta, mask
    return out, delta


def mrope_table(positions3: torch.Tensor, head_dim: int, theta: float, mrope_section: Sequence[int]):
    """Per-token cos / sin [T, head_dim / 2] for 3-D positions [3, T]: rotary pair i uses the temporal, row
    or column position according to which of the `mrope_section` ranges it falls into."""
    half = head_dim // 2
    assert sum(mrope_section) == half, (mrope_section, half)
    dev = positions3.device
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float64, device=dev) / head_dim))
    axis = torch.repeat_interleave(torch.arange(3, device=dev), torch.tensor(list(mrope_section), device=dev))
    pos = positions3.to(torch.float64)[axis]  # [half, T]
    ang = (pos * inv.unsqueeze(1)).t()
    return ang.cos().float().contiguous(), ang.sin().float().contiguous()
