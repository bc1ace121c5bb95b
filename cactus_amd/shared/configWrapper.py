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


def initLastz(config_root, gpu_option):
    """Resolves --gpu {N|all} into <blast gpu="N"> like initLastz does (:296-317): 'all' means every
    visible device; asking for more than are visible is an error instead of a silent fallback."""
    blast = config_root.find("blast")
    if gpu_option is None:
        return config_root
    if str(gpu_option) == "all":
        n = count_amd_gpus()
        if n == 0:
            raise RuntimeError("--gpu all requested but no AMD GPU is visible")
    else:
        n = int(gpu_option)
        if n > 0 and n > count_amd_gpus():
            raise RuntimeError("--gpu {} requested but only {} visible".format(n, count_amd_gpus()))
    blast.attrib["gpu"] = str(n)
    return config_root
