"""CPU oracle for the MinLZ block codec — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/minlz_oracle.h for the pinning status and the rules.
"""
from .oracle import *  # noqa: F401,F403
