"""Token-level similarity of files of this repository against their namesakes in a reference tree
(comments and docstrings stripped; difflib ratio over Python token strings).

    python tools/similarity_check.py /root/reference areal_vllm_amd/api/workflow_api.py:areal/api/workflow_api.py ...
"""
import difflib
import io
import sys
import tokenize


def toks(path):
    out = []
    prev = None
    with open(path, "rb") as f:
        for t in tokenize.tokenize(f.readline):
            if t.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT,
                          tokenize.ENCODING, tokenize.ENDMARKER):
                prev = t
                continue
            if t.type == tokenize.STRING and (prev is None or prev.type in (tokenize.INDENT, tokenize.NEWLINE, tokenize.NL,
                                                                               tokenize.ENCODING, tokenize.DEDENT)):
                prev = t
                continue  # docstring
            out.append(t.string)
            prev = t
    return out


def main():
    ref_root = sys.argv[1]
    for pair in sys.argv[2:]:
        mine, ref = pair.split(":")
        a, b = toks(mine), toks(f"{ref_root}/{ref}")
        print(f"{difflib.SequenceMatcher(None, a, b, autojunk=False).ratio():.2f}  {mine}  ({len(a)} vs {len(b)} tokens)")


if __name__ == "__main__":
    main()
