"""The second call site of lastz / run_kegalign in Cactus: the lastz repeat masker (SURVEY.md section 8 row f3;
/root/reference/src/cactus/preprocessor/lastzRepeatMasking/cactus_lastzRepeatMask.py:66-180).  Off by default in
the reference (cactus_progressive_config.xml:36 active="0"), kept here because it reuses exactly the seed +
ungapped kernels of the blast path:

    getFragments          :66-77    cactus_fasta_fragments.py  --fragment=F --step=F/2 --origin=zero
    alignFastaFragments   :79-133   lastz target[multiple] fragments --ungapped --queryhsplimit=keep,nowarn:N
                                    --querydepth=keep,nowarn:P --format=general:name1,zstart1,end1,name2,zstart2+,end2+ --markend
    maskCoveredIntervals  :135-162  cactus_covered_intervals --origin=one M=2*period --queryoffsets
                                    cactus_fasta_softmask_intervals.py --origin=one

The three helper programs are in-tree in the reference (preprocessor/lastzRepeatMasking/); their documented
behaviour is restated with numpy (depth counting is a histogram -- difference array + cumsum -- instead of the
sliding byte window of cactus_covered_intervals.c:392-394, identical output for sorted-by-query input)."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from cactus_amd.shared.common import cactus_call


@dataclass
class RepeatMaskOptions:                       # cactus_lastzRepeatMask.py:28-45
    fragment: int = 200
    minPeriod: int = 50
    lastzOpts: str = "--step=3 --ambiguous=iupac,100,100 --ungapped --queryhsplimit=keep,nowarn:1500"
    gpu: int = 0
    unmaskInput: bool = False
    unmaskOutput: bool = False
    eventName: str = "seq"

    @property
    def period(self):
        return self.minPeriod


def _fasta_records(text: str):
    name, parts = None, []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith(">"):
            if name is not None:
                yield name, "".join(parts)
            name, parts = (line[1:].strip().split() or [""])[0], []
        elif name is not None:
            parts.append(line)
    if name is not None:
        yield name, "".join(parts)


def fasta_fragments(fasta_text: str, fragment: int = 100, step: int = 50, origin: str = "one") -> str:
    """cactus_fasta_fragments.py:45-109: overlapping upper-cased fragments named NAME_<start>; all-N fragments dropped."""
    out = []
    all_n = "N" * fragment
    for name, seq in _fasta_records(fasta_text):
        seq = seq.upper()
        for ix in range(0, len(seq), step):
            frag = seq[ix:min(ix + fragment, len(seq))]
            if frag == all_n:
                continue
            out.append(">%s_%d\n%s\n" % (name, ix if origin == "zero" else ix + 1, frag))
    return "".join(out)


def covered_intervals(general_lines, M: int = 1, origin_one: bool = False, query_offsets: bool = False, markend: bool = False) -> str:
    """cactus_covered_intervals.c: input lines `<ref> <rstart> <rend> <q>[_<offset>] <qstart+> <qend+>` (origin-zero,
    half-open); reports query intervals covered by at least M alignments (depth saturates at 255, :56,:393);
    self-alignments are skipped (:355); output `chrom<TAB>start<TAB>end`, origin-one closed if origin_one (:440-470)."""
    per_chrom = {}
    order = []
    for line in general_lines:
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        f = line.split()
        rchrom, rs, re_, qchrom, qs, qe = f[0], int(f[1]), int(f[2]), f[3], int(f[4]), int(f[5])
        if query_offsets:
            qchrom, off = qchrom.rsplit("_", 1)
            qs += int(off); qe += int(off)
        if qchrom == rchrom and qs == rs and qe == re_:
            continue
        if qchrom not in per_chrom:
            per_chrom[qchrom] = ([], [])
            order.append(qchrom)
        per_chrom[qchrom][0].append(qs)
        per_chrom[qchrom][1].append(qe)
    out = []
    o = 1 if origin_one else 0
    for chrom in order:
        starts = np.asarray(per_chrom[chrom][0], dtype=np.int64)
        ends = np.asarray(per_chrom[chrom][1], dtype=np.int64)
        n = int(ends.max()) if len(ends) else 0
        diff = np.zeros(n + 1, dtype=np.int64)
        np.add.at(diff, starts, 1)
        np.add.at(diff, ends, -1)
        depth = np.minimum(np.cumsum(diff[:-1]), 255)
        hit = depth >= M
        edges = np.diff(np.concatenate(([0], hit.astype(np.int8), [0])))
        for s, e in zip(np.nonzero(edges == 1)[0], np.nonzero(edges == -1)[0]):
            out.append("%s\t%d\t%d\n" % (chrom, s + o, e))
    if markend:
        out.append("# covered_intervals end-of-file\n")
    return "".join(out)


def softmask_intervals(fasta_text: str, interval_lines, origin_one: bool = False, unmask: bool = False, wrap: int = 100, min_length=None) -> str:
    """cactus_fasta_softmask_intervals.py:80-155: lower-case the listed intervals (`chrom start end`, origin-zero half-open, or
    origin-one closed) of each sequence; output wrapped at `wrap`.  Error behaviour as the script's asserts: a malformed or empty
    interval, two sequences of one name, or intervals for a sequence the FASTA does not hold raise AssertionError."""
    by_chrom = {}
    for n, line in enumerate(interval_lines, 1):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        f = line.split()
        assert len(f) >= 3, "not enough fields (line %s): %s" % (n, line)
        try:
            c, s, e = f[0], int(f[1]), int(f[2])
            if origin_one:
                s -= 1
            if s < 0 or s >= e:
                raise ValueError
        except ValueError:
            raise AssertionError("bad line (line %s): %s" % (n, line))
        by_chrom.setdefault(c, [])
        if min_length is None or e - s >= min_length:
            by_chrom[c].append((s, e))
    out = []
    seen = set()
    for name, seq in _fasta_records(fasta_text):
        assert name not in seen, "more than one sequence is named %s" % name
        seen.add(name)
        if unmask:
            seq = seq.upper()
        arr = np.frombuffer(seq.encode(), dtype=np.uint8).copy()
        for s, e in by_chrom.get(name, []):
            seg = arr[s:e]
            up = (seg >= 65) & (seg <= 90)
            seg[up] += 32
        seq = arr.tobytes().decode()
        out.append(">%s\n" % name)
        out.extend(seq[i:i + wrap] + "\n" for i in range(0, len(seq), wrap))
    missing = [c for c in by_chrom if c not in seen]
    assert missing == [], "missing fasta sequence %s" % (", ".join(missing))
    return "".join(out)


class LastzRepeatMaskJob:
    """Same three steps as the reference job (cactus_lastzRepeatMask.py:164-180), run eagerly on a LocalJob."""

    def __init__(self, repeatMaskOptions: RepeatMaskOptions, queryID, targetIDs):
        self.repeatMaskOptions, self.queryID, self.targetIDs = repeatMaskOptions, queryID, targetIDs

    def getFragments(self, queryFile):
        fragments = os.path.join(self.work_dir, self.repeatMaskOptions.eventName + '_frag')
        with open(queryFile) as f, open(fragments, "w") as out:
            out.write(fasta_fragments(f.read(), self.repeatMaskOptions.fragment, self.repeatMaskOptions.fragment // 2, "zero"))
        return fragments

    def alignFastaFragments(self, targetFiles, fragments):
        o = self.repeatMaskOptions
        target = os.path.join(self.work_dir, o.eventName + '.fa')
        with open(target, "w") as out:
            for t in targetFiles:
                out.write(open(t).read())
        handling = ['%s[multiple][nameparse=darkspace]' % os.path.basename(target), '%s[nameparse=darkspace]' % os.path.basename(fragments)]
        if o.unmaskInput:
            handling = ['%s[multiple,unmask][nameparse=darkspace]' % os.path.basename(target), '%s[unmask][nameparse=darkspace]' % os.path.basename(fragments)]
        if o.gpu:
            assert not o.unmaskInput
            handling = [os.path.basename(target), os.path.basename(fragments)]
        alignment = os.path.join(self.work_dir, o.eventName + '.cigar')
        tool = 'run_kegalign' if o.gpu else 'lastz'
        lastz_cmd = [tool] + handling + o.lastzOpts.split() + \
            ["--querydepth=keep,nowarn:%i" % (o.period + 3),
             "--format=general:name1,zstart1,end1,name2,zstart2+,end2+",
             "--markend"]
        if o.gpu:
            lastz_cmd += ['--num_threads', '1']
        messages = cactus_call(outfile=alignment, work_dir=self.work_dir, parameters=lastz_cmd, returnStdErr=True)
        if o.gpu:
            for line in (messages or "").lower().split("\n"):
                for keyword in ['terminate', 'error', 'fail', 'assert', 'signal', 'abort', 'segmentation', 'sigsegv', 'kill']:
                    if keyword in line and 'signals' not in line:
                        raise RuntimeError('{} exited 0 but keyword "{}" found in stderr'.format(lastz_cmd, keyword))
        return alignment

    def maskCoveredIntervals(self, queryFile, alignment):
        o = self.repeatMaskOptions
        with open(alignment) as f:
            maskInfo = covered_intervals(f, M=int(o.period * 2), origin_one=True, query_offsets=True)
        with open(queryFile) as f:
            masked = softmask_intervals(f.read(), maskInfo.splitlines(), origin_one=True, unmask=o.unmaskOutput)
        maskedQuery = os.path.join(self.work_dir, o.eventName + '.maskedQeury')
        with open(maskedQuery, "w") as out:
            out.write(masked)
        return maskedQuery

    def run(self, fileStore):
        assert len(self.targetIDs) >= 1 and self.repeatMaskOptions.fragment > 1
        self.work_dir = fileStore.getLocalTempDir()
        queryFile = os.path.join(self.work_dir, self.repeatMaskOptions.eventName + '.query')
        fileStore.readGlobalFile(self.queryID, queryFile)
        targetFiles = [os.path.join(self.work_dir, '{}_{}.tgt'.format(self.repeatMaskOptions.eventName, i)) for i in range(len(self.targetIDs))]
        for targetFile, fileID in zip(targetFiles, self.targetIDs):
            fileStore.readGlobalFile(fileID, targetFile)
        fragments = self.getFragments(queryFile)
        alignment = self.alignFastaFragments(targetFiles, fragments)
        return fileStore.writeGlobalFile(self.maskCoveredIntervals(queryFile, alignment))
