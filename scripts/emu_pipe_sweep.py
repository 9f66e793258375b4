#!/usr/bin/env python
"""dev tool: frames-per-emission-launch sweep of the decoupled pipeline, hipGraph vs plain launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from emu_ab import run
from bench import DEFAULT_KW
for E in (4, 8, 16, 32):
    os.environ["V2E_AMD_PIPE_E"] = str(E)
    run("E=%d graph" % E, dict(DEFAULT_KW), 1)
    run("E=%d plain launches" % E, dict(DEFAULT_KW), 0)
