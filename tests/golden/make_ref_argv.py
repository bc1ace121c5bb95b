"""Writes tests/golden/ref_argv.json: the command lines the REFERENCE's unmodified job functions build for this path
(/root/reference/src/cactus/paf/local_alignment.py imported through tests/refjobs.py; run_lastz per divergence class, CPU and GPU
branch; make_chunked_alignments; the outgroup chain; trim_unaligned_sequences), with work-directory paths reduced to base names.
tests/test_parity_gpu.py replays the lastz / run_kegalign lines against the real front ends on the MI355X -- /root/reference does not
exist there.  usage (in the build container): python tests/golden/make_ref_argv.py"""
import json
import os
import sys
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MIPAF_NATIVE"] = "1"
import refjobs  # noqa: E402
import test_reference_jobs_cpu as t  # noqa: E402
from localjob import LocalJob  # noqa: E402


def main():
    import stat
    tmp = Path(tempfile.mkdtemp(prefix="ref_argv_"))
    shim = tmp / "shim"; shim.mkdir()
    for name in ("lastz", "run_kegalign"):
        p = shim / name
        p.write_text("#!/bin/bash\nargs=()\nwhile [ $# -gt 0 ]; do case \"$1\" in --num_gpu|--num_threads) shift 2;; *) args+=(\"$1\"); shift;; esac; done\nexec %s/oracle/oracle_lastz \"${args[@]}\"\n" % ROOT)
        p.chmod(p.stat().st_mode | stat.S_IEXEC)
    refjobs.path_dirs[:] = [str(shim), os.path.join(ROOT, "bin")]
    ref = refjobs.load()
    out = {"source": "/root/reference/src/cactus/paf/local_alignment.py (unmodified) + /root/reference/src/cactus/cactus_progressive_config.xml", "run_lastz": [], "flows": {}}
    for distance, gpu in ((0.03, 0), (0.07, 0), (0.12, 0), (0.2, 0), (0.24, 0), (0.6, 0), (0.03, 2), (0.2, 1), (0.6, 4)):
        job = LocalJob()
        a, b = t.genome_files(job, tmp, 20000, 7, 1)
        refjobs.calls.clear()
        ref.run_lastz(job, "A", a, "B", b, distance, t.params(gpu=gpu) if gpu else t.params())
        out["run_lastz"].append({"distance": distance, "gpu": gpu, "argv": refjobs.calls[0][0][0]})

    import re

    def norm(cmds):
        base = [[os.path.basename(x) if x.startswith("/") else x for x in c] for c in cmds]
        return [[re.sub(r"^g_[0-9a-f]{32}$", "<global file>", re.sub(r"^tmp[a-z0-9_]{8}$", "<temp dir>", x)) for x in c] for c in base]
    job = LocalJob()
    a, b = t.genome_files(job, tmp, 30000, 9, 2)
    refjobs.calls.clear()
    ref.make_chunked_alignments(job, "A", a, "B", b, 0.6, t.params(chunkSize=12000, overlapSize=500))
    seen = []
    for cmds, kw in refjobs.calls:
        n = norm(cmds)
        key = [c[:2] for c in n]
        if key not in [s[0] for s in seen]:
            seen.append((key, n, kw))
    out["flows"]["make_chunked_alignments"] = [{"argv": n, "how": kw} for _, n, kw in seen]
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "ref_argv.json"), "w"), indent=1)
    print("wrote ref_argv.json:", len(out["run_lastz"]), "run_lastz lines")


if __name__ == "__main__":
    main()
