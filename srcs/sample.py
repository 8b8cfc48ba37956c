"""`python -m srcs.sample` entry point: forwards to the MI355X implementation (ladiffcodec_amd/sample.py)."""
from ladiffcodec_amd.sample import build_parser, main, synthesis  # noqa: F401

if __name__ == "__main__":
    main()
