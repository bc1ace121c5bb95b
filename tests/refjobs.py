"""The reference's OWN job functions, unmodified: /root/reference/src/cactus/paf/local_alignment.py imported with stand-ins for the
modules it imports but that are not installable here (toil, sonLib, Bio) and for the parts of the cactus package
that are out of scope (cactus.shared.common.cactus_call -> a local-binaries Popen that puts our front ends on PATH and records every
argv).  Test infrastructure: only usable where /root/reference exists (this container) -- the GPU box replays the recorded argv from
tests/golden/ref_argv.json instead (tests/test_parity_gpu.py).

    ref = load()                      # module object of the reference file
    ref.run_lastz(job, ...)           # the reference's statements, our bin/lastz
    calls                             # [(argv or [argv, ...], kwargs)] in call order
"""
from __future__ import annotations

import importlib.util
import logging
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FILE = "/root/reference/src/cactus/paf/local_alignment.py"

calls = []                 # every cactus_call of the reference's functions: (parameters, outfile is not None)
path_dirs = [os.path.join(ROOT, "bin")]


def available() -> bool:
    return os.path.exists(REF_FILE)


def cactus_call(parameters, outfile=None, work_dir=None, returnStdErr=False, gpus=None, cpus=None, job_memory=None, outappend=False,
                check_output=False, **kw):
    """stand-in for cactus.shared.common.cactus_call (local binaries mode, common.py:732-994): the commands -- one, or a list piped
    into each other -- with path_dirs first on PATH, stdout to outfile / returned, RuntimeError on a non-zero exit"""
    commands = [parameters] if isinstance(parameters[0], str) else list(parameters)
    calls.append(([list(c) for c in commands], dict(outfile=bool(outfile), outappend=outappend, returnStdErr=returnStdErr, check_output=check_output)))
    env = dict(os.environ, PATH=os.pathsep.join(path_dirs) + os.pathsep + os.environ.get("PATH", ""))
    fh = open(outfile, "ab" if outappend else "wb") if outfile else None
    procs, errs = [], []
    try:
        for k, cmd in enumerate(commands):
            last = k == len(commands) - 1
            procs.append(subprocess.Popen(cmd, stdin=procs[-1].stdout if procs else None, stdout=(fh or subprocess.PIPE) if last else subprocess.PIPE,
                                          stderr=subprocess.PIPE, cwd=work_dir, env=env))
            if len(procs) > 1:
                procs[-2].stdout.close()
        out, err = procs[-1].communicate()
        errs = [p.stderr.read() if p is not procs[-1] else err for p in procs]
        for p in procs:
            p.wait()
    finally:
        if fh:
            fh.close()
    for p, cmd, e in zip(procs, commands, errs):
        if p.returncode != 0:
            raise RuntimeError("Command {} exited {}: stderr={}".format(cmd, p.returncode, (e or b"").decode(errors="replace")))
    if check_output:
        return (out or b"").decode()
    if returnStdErr:
        return (err or b"").decode()
    return None


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """the reference module, freshly executed"""
    from cactus_amd.shared.common import cactus_clamp_memory, getOptionalAttrib
    log = logging.getLogger("refjobs")

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from localjob import LocalJob

    class Job(LocalJob):                                   # toil.job.Job: a bare Job() is given its file store by addChild (local_alignment.py:432-433)
        def __init__(self):
            self.fileStore, self.cores, self.memory = None, 1, 2_000_000_000
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.split(".")[0] in ("toil", "sonLib", "Bio", "cactus")}
    for k in saved:
        del sys.modules[k]
    try:
        _module("toil"); _module("toil.job", Job=Job); _module("toil.statsAndLogging", logger=log)
        _module("toil.lib"); _module("toil.lib.bioio", getLogLevelString=lambda: "INFO")
        _module("toil.realtimeLogger", RealtimeLogger=log)
        _module("sonLib"); _module("sonLib.bioio", newickTreeParser=lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("sonLib is not installed")))
        _module("Bio", SeqIO=None)
        _module("cactus"); _module("cactus.paf")
        nope = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("out of scope"))      # noqa: E731
        _module("cactus.paf.paf", get_event_pairs=nope, get_leaves=nope, get_node=nope, get_distances=nope)
        _module("cactus.shared"); _module("cactus.shared.common", cactus_call=cactus_call, getOptionalAttrib=getOptionalAttrib, zip_gz=nope, cactus_clamp_memory=cactus_clamp_memory)
        _module("cactus.preprocessor"); _module("cactus.preprocessor.checkUniqueHeaders", sanitize_fasta_headers=nope)
        _module("cactus.preprocessor.unmasking", unmask_contigs_all=nope)
        _module("cactus.preprocessor.cactus_preprocessor", clean_if_different=nope)
        spec = importlib.util.spec_from_file_location("ref_local_alignment", REF_FILE)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("toil", "sonLib", "Bio", "cactus")]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
