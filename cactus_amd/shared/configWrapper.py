"""GPU plumbing of ConfigWrapper.initLastz (/root/reference/src/cactus/shared/configWrapper.py:281-393)
for AMD hardware: the reference counts devices with toil's count_nvidia_gpus (:19,:307) and asks Toil
for 'cuda:N' accelerators (local_alignment.py:393); here the count comes from the HIP runtime through
libmiblast and the accelerator string is 'rocm:N'."""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blast_config.xml")


def load_config(path: str = DEFAULT_CONFIG):
    return ET.parse(path).getroot()


def count_amd_gpus() -> int:
    from cactus_amd import miblast
    return miblast.device_count()


def accelerator_string(gpu: int):
    return "rocm:{}".format(gpu) if gpu else None


def cactus_cpu_count() -> int:
    """cores this process may use (common.py's cactus_cpu_count: the affinity mask when there is one)"""
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def initLastz(config_root, options):
    """ConfigWrapper.initLastz (configWrapper.py:281-393) for the blast phase on AMD GPUs, FastGA branch left out.
    `options` carries what the reference reads from the command line: gpu (None | N | 'all'), batchSystem, maxCores,
    lastzCores, lastzMemory, latest (any object with those attributes, or a bare --gpu value for the simple case).

      * --gpu overrides <blast gpu> and every lastzRepeatMask <preprocessor gpu> (:292-297); 'all' / legacy 'true' become the
        number of visible devices (count_amd_gpus in place of count_nvidia_gpus, single_machine only, :307-316); anything that
        is not a non-negative integer raises; asking for more devices than are visible raises instead of falling back;
      * with GPUs on, the aligner gets every core of a single machine unless --lastzCores says otherwise (:350-363), because one
        process drives all GPUs of the job; other batch systems must pass --lastzCores (:369);
      * --lastzCores / --lastzMemory land in <blast cpu> / <blast lastz_memory> (read by make_chunked_alignments) and in the
        lastzRepeatMask preprocessors (:374-387);
      * realign is switched off with GPUs (:389-393)."""
    import types
    if not hasattr(options, "__dict__") and not isinstance(options, types.SimpleNamespace):
        options = types.SimpleNamespace(gpu=options)
    opt = lambda name, default=None: getattr(options, name, default)                      # noqa: E731
    blast = config_root.find("blast")
    pp_nodes = [n for n in config_root.findall("preprocessor") if n.attrib.get("preprocessJob") == "lastzRepeatMask"]
    if opt("gpu"):
        blast.attrib["gpu"] = str(opt("gpu"))
        for node in pp_nodes:
            node.attrib["gpu"] = str(opt("gpu"))
        if opt("latest"):
            raise RuntimeError('--latest cannot be used with --gpu')
    batch = str(opt("batchSystem", "single_machine")).lower()

    def get_gpu_count():
        if batch in ('single_machine', 'singlemachine'):
            n = count_amd_gpus()
            if not n:
                raise RuntimeError('Unable to automatically determine number of GPUs: Please set with --gpu N')
            return n
        raise RuntimeError('--gpu N required to set number of GPUs on non single_machine batch systems')

    def resolve(value, what):
        value = str(value)
        if value.lower() == 'false':
            value = "0"
        elif value.lower() == 'true':
            value = 'all'
        if value == 'all':
            return get_gpu_count()
        if not value.isdigit() or int(value) < 0:
            raise RuntimeError('Invalid value for {} gpu count, {}. Please specify a numeric value with --gpu'.format(what, value))
        n = int(value)
        if n > 0 and batch in ('single_machine', 'singlemachine') and n > count_amd_gpus():
            raise RuntimeError("--gpu {} requested but only {} visible".format(n, count_amd_gpus()))
        return n

    blast.attrib["gpu"] = str(resolve(blast.attrib.get("gpu", "0"), "blast"))
    for node in pp_nodes:
        node.attrib["gpu"] = str(resolve(node.attrib.get("gpu", "0"), "repeatmask"))

    lastz_cores = opt("lastzCores") or None
    if int(blast.attrib["gpu"]) and not lastz_cores:
        if batch in ('single_machine', 'singlemachine'):
            max_cores = opt("maxCores")
            lastz_cores = max_cores if max_cores is not None and max_cores < 2 ** 20 else cactus_cpu_count()
        else:
            raise RuntimeError('--lastzCores must be used with --gpu on non-singlemachine batch systems')
    if lastz_cores:
        blast.attrib["cpu"] = str(lastz_cores)
    if opt("lastzMemory"):
        blast.attrib["lastz_memory"] = str(opt("lastzMemory"))
    for node in pp_nodes:
        if lastz_cores:
            node.attrib["cpu"] = str(lastz_cores)
        if opt("lastzMemory"):
            node.attrib["lastz_memory"] = str(opt("lastzMemory"))
    if int(blast.attrib["gpu"]) and blast.attrib.get("realign", "0").lower() not in ("0", "false"):
        blast.attrib["realign"] = "0"
    return config_root
