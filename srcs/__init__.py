"""Drop-in module path of the reference CLI: `python -m srcs.sample ...` (reference README.md:35,39)."""
