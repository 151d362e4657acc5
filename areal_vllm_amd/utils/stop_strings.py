"""Stop strings matched on DECODED text.

The engine matches a stop string as the token-id sequence its tokenizer gives the string in isolation. A model can
spell the same characters with other tokens (a piece merged with its neighbour, a different split), which only the
decoded text shows; the helpers here find the first token count whose text contains a stop string. Used by the HTTP
server while tokens arrive and by the client loop on every answer."""
from typing import List, Optional, Sequence


def window(tokenizer, stops: Sequence[str]) -> int:
    """How many tokens before the new ones have to be decoded along with them so that a stop string that straddles
    the boundary is seen."""
    return max(len(tokenizer.encode(x, add_special_tokens=False)) for x in stops) + 8


def first_cut(tokenizer, tokens: List[int], stops: Sequence[str], first_new: int = 0,
              lookback: Optional[int] = None) -> Optional[int]:
    """Smallest n > first_new such that the text of tokens[:n] contains one of `stops` (None if it never does).
    Tokens before `first_new` are known not to contain one."""
    stops = [x for x in stops if x]
    if not stops or len(tokens) <= first_new:
        return None
    lo = max(0, first_new - (window(tokenizer, stops) if lookback is None else lookback))
    if not any(x in tokenizer.decode(tokens[lo:]) for x in stops):
        return None
    for n in range(first_new + 1, len(tokens) + 1):
        if any(x in tokenizer.decode(tokens[lo:n]) for x in stops):
            return n
    return None
