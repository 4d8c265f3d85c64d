"""The closed-loop harness lives in the package (esvo_amd/closed_loop.py: bench.py reports it as an operating point)."""
from esvo_amd.closed_loop import *  # noqa: F401,F403
from esvo_amd.closed_loop import run, register, cayley2rot, orth, TICK_NS  # noqa: F401
