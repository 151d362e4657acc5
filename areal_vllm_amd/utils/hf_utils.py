"""Tokenizer / processor loading helpers (``areal/utils/hf_utils.py`` in the reference)."""
from typing import Optional


def load_hf_tokenizer(model_name_or_path: str, fast_tokenizer: bool = True, padding_side: Optional[str] = None):
    import transformers

    kw = {}
    if padding_side is not None:
        kw["padding_side"] = padding_side
    tok = transformers.AutoTokenizer.from_pretrained(model_name_or_path, fast_tokenizer=fast_tokenizer,
                                                     trust_remote_code=True, force_download=False, **kw)
    if tok.pad_token_id is None:
        tok.pad_token_id = tok.eos_token_id
    return tok


def load_hf_processor_and_tokenizer(model_name_or_path: str, fast_tokenizer: bool = True,
                                    padding_side: Optional[str] = None):
    import transformers

    tok = load_hf_tokenizer(model_name_or_path, fast_tokenizer, padding_side)
    try:
        proc = transformers.AutoProcessor.from_pretrained(model_name_or_path, trust_remote_code=True,
                                                          force_download=False)
    except Exception:
        proc = None
    return proc, tok
