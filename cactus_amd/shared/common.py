"""Process boundary of the blast path: the local-binaries slice of cactus_call
(/root/reference/src/cactus/shared/common.py:732-994) and getOptionalAttrib (:226), with the same
argument names, stdout->outfile behaviour and RuntimeError-on-non-zero-exit convention
(:962-988).  Docker / singularity modes, memory accounting and realtime logging are out of
scope (orchestration, unchanged).  The MI355X front ends live in <repo>/bin and
are put first on PATH, which is how CACTUS_BINARIES_MODE=local finds `lastz` and `paffy` (:793-795)."""
from __future__ import annotations

import os
import subprocess
import threading

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN_DIR = os.path.join(REPO_ROOT, "bin")


def getOptionalAttrib(node, attribName, typeFn=None, default=None, errorIfNotPresent=False):
    """Same contract as common.py:226: typed attribute lookup with default; bool('0') is False."""
    if node is not None and attribName in node.attrib:
        if typeFn is not None:
            if typeFn == bool:
                aname = node.attrib[attribName].lower()
                if aname == "false":
                    return False
                if aname == "true":
                    return True
                return bool(int(node.attrib[attribName]))
            return typeFn(node.attrib[attribName])
        return node.attrib[attribName]
    if errorIfNotPresent:
        raise RuntimeError("Could not find attribute %s in %s node" % (attribName, node))
    return default


def cactus_call(parameters, outfile=None, work_dir=None, returnStdErr=False, gpus=None, cpus=None, job_memory=None,
                outappend=False, check_output=False, env=None):
    """Runs one command locally, or -- when `parameters` is a list of commands -- the commands piped into each other as the
    reference's cactus_call does for the chaining stage (local_alignment.py:684-691).  stdout of the last command goes to
    `outfile` (or is returned when check_output), stderr is captured; a non-zero exit of any command raises RuntimeError
    carrying the command line and stderr."""
    assert parameters
    commands = [parameters] if isinstance(parameters[0], str) else list(parameters)
    call_env = dict(os.environ if env is None else env)
    call_env["PATH"] = BIN_DIR + os.pathsep + call_env.get("PATH", "")
    stdout = subprocess.PIPE if check_output else None
    fh = None
    if outfile is not None:
        fh = open(outfile, "ab" if outappend else "wb")
        stdout = fh
    procs = []
    try:
        for k, cmd in enumerate(commands):
            last = k == len(commands) - 1
            procs.append(subprocess.Popen(cmd, stdin=procs[-1].stdout if procs else None, stdout=stdout if last else subprocess.PIPE,
                                          stderr=subprocess.PIPE, cwd=work_dir, env=call_env))
            if len(procs) > 1:
                procs[-2].stdout.close()                       # the reader owns the pipe now (SIGPIPE reaches the writer)
        # stderr of the intermediate stages is drained while they run: a stage that writes more than a pipe buffer of
        # diagnostics would otherwise block for ever with the last stage waiting on its output
        drained = [b""] * len(procs)

        def drain(i):
            drained[i] = procs[i].stderr.read()

        readers = [threading.Thread(target=drain, args=(i,), daemon=True) for i in range(len(procs) - 1)]
        for t in readers:
            t.start()
        out, err = procs[-1].communicate()
        for t in readers:
            t.join()
        errs = drained[:-1] + [err]
        for p in procs[:-1]:
            p.wait()
    finally:
        if fh is not None:
            fh.close()
    err_text = "".join(e.decode(errors="replace") for e in errs if e)
    for p, cmd in zip(procs, commands):
        if p.returncode != 0:
            raise RuntimeError("Command {} exited {}: stderr={}".format(cmd, p.returncode, err_text))
    if check_output:
        return out.decode()
    if returnStdErr:
        return err_text
    return None


def cactus_clamp_memory(memory, floor=2 ** 28, ceiling=2 ** 40):
    """common.py's clamp of a job's memory request to a sane range"""
    return int(min(max(memory, floor), ceiling))


def getLogLevelString():
    return "INFO"
