"""Write the countdown train / test JSONL files.

python examples/countdown/countdown.py --out /tmp/areal/data/countdown --train 50000 --test 1024
"""
import argparse
import os

from areal_vllm_amd.dataset.countdown import generate, write_jsonl

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/areal/data/countdown")
    ap.add_argument("--train", type=int, default=50000)
    ap.add_argument("--test", type=int, default=1024)
    ap.add_argument("--min-numbers", type=int, default=3)
    ap.add_argument("--max-numbers", type=int, default=4)
    args = ap.parse_args()
    write_jsonl(os.path.join(args.out, "train.jsonl"), generate(args.train, 0, args.min_numbers, args.max_numbers))
    write_jsonl(os.path.join(args.out, "test.jsonl"), generate(args.test, 1, args.min_numbers, args.max_numbers))
    print(f"wrote {args.train} + {args.test} rows under {args.out}")
