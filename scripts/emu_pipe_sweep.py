#!/usr/bin/env python
"""dev tool: frames-per-emission-launch sweep of the default device-resident pipeline (hipGraph)."""
import sys, os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = "import sys; sys.path.insert(0, %r); from emu_ab import run; from bench import DEFAULT_KW; run(%%r, dict(DEFAULT_KW), 1, steps=20)" % here
for E in (8, 16, 24, 32):
    e = dict(os.environ); e["V2E_AMD_PIPE_E"] = str(E)
    subprocess.run([sys.executable, "-c", code % ("E=%d" % E)], env=e)
