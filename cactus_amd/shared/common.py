"""Process boundary of the blast path: the local-binaries slice of cactus_call
(/root/reference/src/cactus/shared/common.py:732-994) and getOptionalAttrib (:226), with the same
argument names, stdout->outfile behaviour and RuntimeError-on-non-zero-exit convention
(:962-988).  The MI355X front ends live in <repo>/bin and are put first on PATH, which is how
CACTUS_BINARIES_MODE=local finds `lastz` and `paffy` (:793-795).  The container modes are here as far as the GPU job needs them:
dockerCommand / singularityCommand build the argv of common.py:536-560 / :646-705 with the AMD way of handing a container its GPUs
(the reference's `--gpus N` / `--nv` are NVIDIA's; SURVEY 8b, last row) and cactus_call wraps a command in them under
CACTUS_BINARIES_MODE=docker|singularity.  Image pulls, memory accounting and realtime logging stay with the reference
(orchestration, unchanged)."""
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


# ---- containers on an AMD node (common.py:536-560 singularityCommand, :646-705 dockerCommand) -----------------------------------------
# A ROCm container sees a GPU through the kernel driver's device nodes, not through a runtime hook: /dev/kfd (compute) and the render
# nodes under /dev/dri, with the groups that own them; WHICH GPUs of the node is ROCR_VISIBLE_DEVICES inside the container (the role
# `--gpus "device=..."` plays for NVIDIA under Slurm, common.py:668-671).  Singularity / Apptainer bind the same with --rocm.
AMD_DOCKER_DEVICE_ARGS = ['--device=/dev/kfd', '--device=/dev/dri', '--group-add', 'video', '--group-add', 'render',
                          '--security-opt', 'seccomp=unconfined']


def amd_visible_devices(gpus, environ=None):
    """The value of ROCR_VISIBLE_DEVICES for a job of `gpus` GPUs: what the scheduler assigned if it says (Slurm's SLURM_JOB_GPUS, or a
    ROCR_ / HIP_VISIBLE_DEVICES already set for the worker), else the first `gpus` ordinals."""
    environ = os.environ if environ is None else environ
    for name in ('SLURM_JOB_GPUS', 'ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES'):
        if environ.get(name):
            return environ[name]
    return ','.join(str(i) for i in range(int(gpus)))


def dockerCommand(tool=None, work_dir=None, parameters=None, rm=True, port=None, dockstore=None, entrypoint=None, gpus=None, cpus=None,
                  environ=None):
    """argv of `docker run` for one tool invocation: the reference's call (:660-705) with the GPUs of the job handed over the AMD way.
    `tool` is the image (the reference resolves it with getDockerImage(gpu=...): the `-gpu` tag suffix of :383-397 is its concern)."""
    work_dir = os.getcwd() if work_dir is None else work_dir
    call = ['docker', 'run', '--interactive', '--net=host', '--log-driver=none', '-u', '%s:%s' % (os.getuid(), os.getgid()),
            '-v', '{}:/data'.format(os.path.abspath(work_dir))]
    if gpus:
        call += AMD_DOCKER_DEVICE_ARGS + ['-e', 'ROCR_VISIBLE_DEVICES=' + amd_visible_devices(gpus, environ), '-e', 'HSA_ENABLE_IPC_MODE_LEGACY=0']
    if cpus:
        call += ['--cpus', str(cpus)]
    call += ['--entrypoint', entrypoint if entrypoint is not None else '/opt/cactus/wrapper.sh']
    if port is not None:
        call += ['-p', '{0}:{0}'.format(port)]
    if rm:
        call += ['--rm']
    return call + [tool] + list(parameters or [])


def singularityCommand(tool=None, work_dir=None, parameters=None, port=None, file_store=None, gpus=None, cpus=None):
    """argv of `singularity exec` (:536-585): --rocm where the reference passes --nv; `tool` is the sandbox / image path."""
    work_dir = os.getcwd() if work_dir is None else work_dir
    call = ['singularity', '--silent', 'exec', '-u', '-B', '{}:{}'.format(os.path.abspath(work_dir), '/mnt'), '--pwd', '/mnt']
    if gpus:
        call += ['--rocm']
    return call + [tool] + list(parameters or [])


def container_work_dir(work_dir, parameters):
    """What the reference's prepareWorkDir does (common.py:695-730): a container only sees the ONE directory that is mounted into it
    (/data for docker, /mnt for singularity -- the container's working directory), so (1) without a work_dir the directory is derived from
    the arguments that name existing files or directories (their common prefix when they lie in several), falling back to the current
    directory, and (2) every argument loses that directory prefix -- also inside an argument that carries several paths -- so that the
    tool opens the files relative to the mount.  Returns (work_dir, rewritten parameters)."""
    if not work_dir:
        dirs = {os.path.dirname(par) for par in parameters if os.path.isfile(par) or os.path.isdir(par)}
        if len(dirs) > 1:
            work_dir = os.path.commonprefix(sorted(dirs))
        elif dirs:
            work_dir = dirs.pop()
    if not work_dir:
        work_dir = os.getcwd()
    if work_dir == '.' or os.environ.get('CACTUS_DOCKER_MODE', 1) == "0":
        return work_dir, list(parameters)
    prefix = work_dir if work_dir.endswith('/') else work_dir + '/'
    return work_dir, [par.replace(prefix, '') for par in parameters]


def cactus_call(parameters, outfile=None, work_dir=None, returnStdErr=False, gpus=None, cpus=None, job_memory=None,
                outappend=False, check_output=False, env=None):
    """Runs one command locally, or -- when `parameters` is a list of commands -- the commands piped into each other as the
    reference's cactus_call does for the chaining stage (local_alignment.py:684-691).  stdout of the last command goes to
    `outfile` (or is returned when check_output), stderr is captured; a non-zero exit of any command raises RuntimeError
    carrying the command line and stderr."""
    assert parameters
    commands = [parameters] if isinstance(parameters[0], str) else list(parameters)
    call_env = dict(os.environ if env is None else env)
    mode = call_env.get("CACTUS_BINARIES_MODE", "local")
    if mode in ("docker", "singularity"):
        # the tool runs inside the image named by CACTUS_DOCKER_IMAGE / CACTUS_SINGULARITY_IMG (the reference resolves these itself:
        # getDockerImage, importSingularityImage); the job's GPUs go in the AMD way
        image = call_env.get("CACTUS_DOCKER_IMAGE" if mode == "docker" else "CACTUS_SINGULARITY_IMG")
        if not image:
            raise RuntimeError("CACTUS_BINARIES_MODE={} needs {}".format(mode, "CACTUS_DOCKER_IMAGE" if mode == "docker" else "CACTUS_SINGULARITY_IMG"))
        # ONE container per call, as in the reference (common.py:764-778): piped commands become `bash -c 'set -eo pipefail && a | b'` inside it
        # (docker: bash as the entry point), and the arguments are rewritten relative to the mounted work directory
        entrypoint = None
        if len(commands) > 1:
            import shlex
            flat = [x for c in commands for x in c]
            work_dir, _ = container_work_dir(work_dir, flat)
            inner = ['bash', '-c', 'set -eo pipefail && ' + ' | '.join(' '.join(shlex.quote(x) for x in c) for c in commands)]
            if mode == "docker":
                entrypoint, inner = '/bin/bash', inner[1:]
        else:
            inner = commands[0]
        work_dir, inner = container_work_dir(work_dir, inner)
        if mode == "docker":
            commands = [dockerCommand(tool=image, work_dir=work_dir, parameters=inner, gpus=gpus, cpus=cpus, entrypoint=entrypoint)]
        else:
            commands = [singularityCommand(tool=image, work_dir=work_dir, parameters=inner, gpus=gpus, cpus=cpus)]
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
