"""`python -m srcs.train` entry point: forwards to the MI355X implementation (ladiffcodec_amd/train_loop.py)."""
from ladiffcodec_amd.train_loop import build_parser, main, run  # noqa: F401

if __name__ == "__main__":
    main()
