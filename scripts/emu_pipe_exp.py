import sys, os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = "import sys; sys.path.insert(0, %r); from emu_ab import run; from bench import DEFAULT_KW; run(%%r, dict(DEFAULT_KW), %%d, steps=20)" % here
P = {"V2E_EXP_STEP_PRIO": "1"}
for label, env in [("base", {}), ("step prio", P), ("prio + lds 16K", dict(P, V2E_EXP_EMIT_LDS="16000")),
                   ("prio + lds 24K", dict(P, V2E_EXP_EMIT_LDS="24000")), ("prio + lds 32K", dict(P, V2E_EXP_EMIT_LDS="32000")),
                   ("prio + lds 40K", dict(P, V2E_EXP_EMIT_LDS="40000")), ("base again", {})]:
    e = dict(os.environ); e.update(env); e["V2E_AMD_PIPE_E"] = "16"
    for ug in (1, 0):
        subprocess.run([sys.executable, "-c", code % (label + (" graph" if ug else " plain"), ug)], env=e)
