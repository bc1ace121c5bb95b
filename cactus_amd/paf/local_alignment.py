"""Host side of the blast phase where it differs from the reference's: the job functions of
/root/reference/src/cactus/paf/local_alignment.py that sit on the hot path, with the reference's names, arguments, return values and
error behaviour, written around the MI355X front ends.

The reference's OWN function bodies are what the CPU suite runs against <repo>/bin (tests/refjobs.py imports the file unmodified:
run_lastz, make_chunked_alignments, combine_chunks, the outgroup chain, trim_unaligned_sequences); this module exists for what that
cannot cover -- the GPU box has no /root/reference, a deployment has no Toil stand-ins -- and holds only what the AMD path needs:

    select_lastz_params       :41-53    distance -> option string
    lastz_command             :54-68    argv of one chunk-pair job (bin/lastz / bin/run_kegalign take the reference's grammar)
    run_lastz                 :29-97    one chunk-pair job -> PAF file id; MIBLAST_INPROCESS=1: the same job through the C ABI
    make_chunked_alignments   :370-408  chunk both genomes, one run_lastz job per chunk pair ('rocm:N' accelerators), combine
    combine_chunks            :336-367  dechunk + concatenate (batched)
    make_ingroup_to_outgroup_alignments_0..3  :421-526  the outgroup chain with trimming between the calls
    chain_alignments, chain_tile_trim_filter_one_contig  :607-727  paffy chain | tile | trim | filter

faffy / paffy's text steps are cactus_amd.paf.chunking (or the native text code behind bin/faffy, bin/paffy); the chaining stage's
`paffy` is <repo>/bin/paffy.  trim_unaligned_sequences (:861-904) is at the end of this file: the three tools of its body are bin/paffy
to_bed, bin/faffy extract, bin/paffy upconvert (the reference's own body drives them in the CPU suite; cactus_amd.paf.chunking.trim_to_aligned
is the same on text in process).  cactus_consolidated and everything after it are untouched.
"""
from __future__ import annotations

import glob
import math
import os
import shutil

from cactus_amd.paf import chunking
from cactus_amd.shared.common import cactus_call, cactus_clamp_memory, getLogLevelString, getOptionalAttrib
from cactus_amd.shared.configWrapper import accelerator_string

STDERR_KEYWORDS = ['terminate', 'error', 'fail', 'assert', 'signal', 'abort', 'segmentation', 'sigsegv', 'kill']
DIVERGENCE_CLASSES = ("one", "two", "three", "four", "five")


def _blast(params):
    return params.find("blast")


def _blast_int(params, name, default):
    return getOptionalAttrib(_blast(params), name, typeFn=int, default=default)


def select_lastz_params(distance, params, gpu):
    """The option set of a pair at `distance` (:44-51): the first divergence class whose bound is not below it, else "default";
    always "default" with <divergences useDefault>."""
    option_sets = _blast(params).find("kegalignArguments" if gpu else "lastzArguments").attrib
    bounds = params.find("constants").find("divergences")
    if getOptionalAttrib(bounds, 'useDefault', typeFn=bool, default=False):
        return option_sets["default"]
    return next((option_sets[c] for c in DIVERGENCE_CLASSES if distance <= float(bounds.attrib[c])), option_sets["default"])


def lastz_command(target_fa, query_fa, options, gpu, cores):
    """argv of the chunk-pair job (:54-68).  CPU form: file names carry lastz's modifiers; GPU form: bare names, and the devices /
    host threads of the job go last as separate tokens."""
    if gpu:
        assert gpu > 0
        return ['run_kegalign', target_fa, query_fa, '--format=paf:wfmash'] + options.split(' ') + ['--num_gpu', str(gpu), '--num_threads', str(cores)]
    return ['lastz', target_fa + '[multiple][nameparse=darkspace]', query_fa + '[nameparse=darkspace]', '--format=paf:wfmash'] + options.split(' ')


def stderr_offence(messages):
    """(keyword, line) of the first stderr line the GPU branch treats as a failure although the exit code was 0 (:75-83), or None."""
    for line in (messages or "").lower().split("\n"):
        if line.startswith("signals delivered"):
            continue
        hit = next((k for k in STDERR_KEYWORDS if k in line and 'signals' not in line), None)
        if hit:
            return hit, line
    return None


def run_lastz(job, name_A, genome_A, name_B, genome_B, distance, params):
    scratch = job.fileStore.getLocalTempDir()
    local = {name: os.path.join(scratch, name + '.fa') for name in (name_A, name_B)}
    job.fileStore.readGlobalFile(genome_A, local[name_A])
    job.fileStore.readGlobalFile(genome_B, local[name_B])
    paf_path = os.path.join(scratch, name_A + '_' + name_B + '.paf')
    gpu, cpu = _blast_int(params, 'gpu', 0), _blast_int(params, 'cpu', None)
    command = lastz_command(name_A + '.fa', name_B + '.fa', select_lastz_params(distance, params, gpu), gpu, job.cores)
    if os.environ.get("MIBLAST_INPROCESS") == "1":
        messages = _run_inprocess(command, scratch, paf_path)
    else:
        messages = cactus_call(parameters=command, outfile=paf_path, work_dir=scratch, returnStdErr=True, gpus=gpu, cpus=cpu, job_memory=job.memory)
    offence = stderr_offence(messages) if gpu else None            # (our front ends keep stderr empty on success)
    if offence:
        job.fileStore.logToMaster("KegAlign offending line: " + offence[1])
        raise RuntimeError('{} exited 0 but keyword "{}" found in stderr'.format(command, offence[0]))
    return job.fileStore.writeGlobalFile(paf_path)


def _run_inprocess(lastz_cmd, work_dir, alignment_file):
    """The same job through libmiblast's C ABI instead of a subprocess."""
    import ctypes as C
    from cactus_amd import miblast
    lib = miblast.load()
    argv = [a.encode() for a in lastz_cmd]
    arr = (C.c_char_p * len(argv))(*argv)
    p = miblast.Params()
    files = (C.c_char_p * 2)()
    ng, nt = C.c_int(), C.c_int()
    rc = lib.miblast_params_from_argv(len(argv), arr, C.byref(p), files, C.byref(ng), C.byref(nt))
    if rc != 0:
        raise RuntimeError("Command {} exited {}: stderr={}".format(lastz_cmd, 2, lib.miblast_last_error().decode()))
    # one context per GPU of the job (--num_gpu of the run_kegalign form, 1 for lastz): the block pairs of the two files are
    # dealt to them (include/miblast.h, miblast_multi)
    try:
        ctx = miblast.Multi(max(1, ng.value))
    except miblast.MiblastError as e:
        raise RuntimeError("Command {} exited {}: stderr={}".format(lastz_cmd, 3, e))
    try:
        fd = os.open(alignment_file, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            rc = lib.miblast_multi_align_files(ctx._h, os.path.join(work_dir, files[0].decode()).encode(),
                                               os.path.join(work_dir, files[1].decode()).encode(), C.byref(p), fd, None)
        finally:
            os.close(fd)
        if rc != 0:
            raise RuntimeError("Command {} exited {}: stderr={}".format(lastz_cmd, 1, lib.miblast_last_error().decode()))
    finally:
        ctx.close()
    return ""


def combine_chunks(job, chunked_alignment_files, batch_size):
    """:336-356 -- every chunk pair's PAF back in genome coordinates (paffy dechunk), end to end; from two batches' worth of files on,
    one child job per batch and a merge behind them."""
    n = len(chunked_alignment_files)
    if n >= 2 * batch_size:
        parts = [job.addChildJobFn(combine_chunks, chunked_alignment_files[at:at + batch_size], batch_size).rv() for at in range(0, n, batch_size)]
        assert len(parts) == math.ceil(n / batch_size)
        return job.addFollowOnJobFn(merge_combined_chunks, parts).rv()
    combined = job.fileStore.getLocalTempFile()
    for file_id in chunked_alignment_files:
        chunking.paf_dechunk(job.fileStore.readGlobalFile(file_id), combined, append=True)
        job.fileStore.deleteGlobalFile(file_id)
    return job.fileStore.writeGlobalFile(combined)


def merge_combined_chunks(job, combined_chunks):
    merged = job.fileStore.getLocalTempFile()
    concat_global_files(job, combined_chunks, merged)
    return job.fileStore.writeGlobalFile(merged)


def make_chunked_alignments(job, event_a, genome_a, event_b, genome_b, distance, params):
    blast = _blast(params).attrib
    gpu, cores, fixed_memory = _blast_int(params, 'gpu', 0), _blast_int(params, 'cpu', None), _blast_int(params, 'lastz_memory', None)
    chunk_size, overlap = int(blast['bigChunkSize' if gpu else 'chunkSize']), int(blast["overlapSize"])

    def chunks_of(genome):
        files = chunking.fasta_chunk(job.fileStore.readGlobalFile(genome), job.fileStore.getLocalTempDir(), chunk_size, overlap)
        return [job.fileStore.writeGlobalFile(f, cleanup=True) for f in files]

    pair_pafs = []
    pieces_a, pieces_b = chunks_of(genome_a), chunks_of(genome_b)
    for i, a in enumerate(pieces_a):
        for j, b in enumerate(pieces_b):
            both = a.size + b.size
            memory = fixed_memory or max(200000000, 15 * both)
            pair_pafs.append(job.addChildJobFn(run_lastz, '{}_{}'.format(event_a, i), a, '{}_{}'.format(event_b, j), b, distance, params,
                                               cores=cores, disk=max(4 * both, memory), memory=memory,
                                               accelerators=accelerator_string(gpu)).rv())        # AMD GPUs: 'rocm:N' (the reference asks for 'cuda:N')
    return job.addFollowOnJobFn(combine_chunks, pair_pafs, _blast_int(params, 'dechunkBatchSize', int(1e9))).rv()


# ---- ingroup -> outgroup alignments with progressive trimming (:411-526) ----------------------------------------------------
def invert_alignments(job, alignment_file):
    """paffy invert (:411-418): query and target of every record swapped"""
    inverted = job.fileStore.getLocalTempFile()
    with open(job.fileStore.readGlobalFile(alignment_file)) as records, open(inverted, "w") as out:
        out.writelines(chunking.paf_invert_line(r) for r in records if r.strip())
    job.fileStore.deleteGlobalFile(alignment_file)
    return job.fileStore.writeGlobalFile(inverted)


def make_ingroup_to_outgroup_alignments_0(job, ingroup_event, outgroup_events, event_names_to_sequences, distances, params):
    chain = job.addChildJobFn(make_ingroup_to_outgroup_alignments_1, ingroup_event, outgroup_events, event_names_to_sequences, distances, params).rv()
    return job.addFollowOnJobFn(invert_alignments, chain).rv()         # (the outgroup becomes the query)


def make_ingroup_to_outgroup_alignments_1(job, ingroup_event, outgroup_events, event_names_to_sequences, distances, params):
    """events are plain names here (the reference passes tree nodes and uses .iD); distances is keyed by (ingroup, outgroup)"""
    nearest, later = outgroup_events[0], outgroup_events[1:]
    found = job.addChildJobFn(make_chunked_alignments, nearest, event_names_to_sequences[nearest], ingroup_event,
                              event_names_to_sequences[ingroup_event], distances[ingroup_event, nearest], params).rv()
    if not later:
        return found
    return job.addFollowOnJobFn(make_ingroup_to_outgroup_alignments_2, found, ingroup_event, later, dict(event_names_to_sequences), distances, params).rv()


def make_ingroup_to_outgroup_alignments_2(job, alignments, ingroup_event, outgroup_events, event_names_to_sequences, distances, params):
    """what of the ingroup stayed unaligned (stretches of trimMinSize and more, trimFlanking bases around them: :451-489) goes to the
    next outgroup"""
    scratch = job.fileStore.getLocalTempDir()
    paf, fasta, rest = (os.path.join(scratch, ingroup_event + ext) for ext in ('.paf', '.fa', '_subseq.fa'))
    job.fileStore.readGlobalFile(alignments, paf)
    job.fileStore.readGlobalFile(event_names_to_sequences[ingroup_event], fasta)
    blast = _blast(params).attrib
    chunking.fasta_extract(chunking.paf_to_bed_unaligned(paf, fasta, int(blast['trimMinSize'])), fasta, rest, int(blast['trimFlanking']))
    event_names_to_sequences[ingroup_event] = job.fileStore.writeGlobalFile(rest)
    if os.path.getsize(rest) == 0:
        return alignments
    more = job.addChildJobFn(make_ingroup_to_outgroup_alignments_1, ingroup_event, outgroup_events, event_names_to_sequences, distances, params).rv()
    return job.addFollowOnJobFn(make_ingroup_to_outgroup_alignments_3, ingroup_event, event_names_to_sequences[ingroup_event], alignments, more).rv()


def make_ingroup_to_outgroup_alignments_3(job, ingroup_event, ingroup_seq_file, alignments, alignments2, has_resources=False):
    """the later outgroups' records back in the ingroup's own coordinates (paffy dechunk --query, :515), behind the first outgroup's"""
    together = job.fileStore.getLocalTempFile()
    shutil.copyfile(job.fileStore.readGlobalFile(alignments), together)
    chunking.paf_dechunk(job.fileStore.readGlobalFile(alignments2), together, query_only=True, append=True)
    job.fileStore.deleteGlobalFile(ingroup_seq_file)
    return job.fileStore.writeGlobalFile(together)


# ---- chaining stage (:594-734): chain -> tile -> trim -> filter -> chain -> filter ------------------------------------------
# Same job functions, arguments, file flow and thresholds as the reference; written around two small helpers (the paffy command
# lines of one per-contig job, and "run these piped commands into this file").
def concat_global_files(job, file_ids, output_path):
    """:594-604 -- the global files end to end in output_path; they are deleted from the store as they are consumed."""
    with open(output_path, 'wb') as sink:
        for fid in file_ids:
            with open(job.fileStore.readGlobalFile(fid), 'rb') as src:
                shutil.copyfileobj(src, sink)
            job.fileStore.deleteGlobalFile(fid)


def _blast_attrib(params, name, default=None):
    attrib = params.find("blast").attrib
    return attrib[name] if default is None else attrib.get(name, default)


def _paffy_commands(params):
    """The command lines of chain_tile_trim_filter_one_contig (:672-681), keyed by role."""
    level = getLogLevelString()
    chain = ['paffy', 'chain']
    for option, attribute in (('--maxGapLength', 'chainMaxGapLength'), ('--chainGapOpen', 'chainGapOpen'),
                              ('--chainGapExtend', 'chainGapExtend'), ('--trimFraction', 'chainTrimFraction')):
        chain += [option, _blast_attrib(params, attribute)]
    return {'chain': chain + ['--logLevel', level],
            'tile': ['paffy', 'tile', '--logLevel', level],
            'trim': ['paffy', 'trim', '--trimIdentity', _blast_attrib(params, 'pafTrimIdentity')],
            'primary': ['paffy', 'filter', '--maxTileLevel', '1'],
            'score': ['paffy', 'filter', '--minChainScore', _blast_attrib(params, 'minPrimaryChainScore')]}


def _pipe_to(job, commands, path, append=False):
    cactus_call(parameters=commands, outfile=path, outappend=append, job_memory=job.memory)


def chain_alignments(job, alignment_files, alignment_names, reference_event_name, params,
                     include_inverted_alignments=True, total_sequence_size=0):
    """:607-657 -- merge the PAF files, append their inverted copy, and chain the lot: in one job when the merged file is at most
    chainSplitMinSize bytes, else split by query contig (paffy split_file, parts of >= chainContigGroupSize bases) with one job
    per part and a merge at the end."""
    scratch = job.fileStore.getLocalTempDir()
    merged = os.path.join(scratch, 'merged.paf')
    concat_global_files(job, alignment_files, merged)
    if include_inverted_alignments:
        snapshot = os.path.join(scratch, 'merged_copy.paf')            # invert reads a copy while its output is appended to the original
        shutil.copyfile(merged, snapshot)
        _pipe_to(job, ['paffy', 'invert', '--inputFile', snapshot], merged, append=True)
        os.remove(snapshot)

    def one_job(path):
        size = os.path.getsize(path)
        return job.addChildJobFn(chain_tile_trim_filter_one_contig, job.fileStore.writeGlobalFile(path), reference_event_name, params,
                                 disk=4 * size, memory=cactus_clamp_memory(4 * size)).rv()

    if os.path.getsize(merged) <= int(_blast_attrib(params, "chainSplitMinSize", "1000000000")):
        return one_job(merged)
    prefix = os.path.join(scratch, 'split_')
    cactus_call(parameters=['paffy', 'split_file', '--inputFile', merged, '--query', '--prefix', prefix,
                            '--minLength', str(int(_blast_attrib(params, "chainContigGroupSize", "10000000"))),
                            '--logLevel', getLogLevelString()], job_memory=job.memory)
    parts = sorted(glob.glob(prefix + '*.paf'), key=lambda path: int(path[len(prefix):-len('.paf')]))
    return job.addFollowOnJobFn(merge_processed_alignments, [one_job(path) for path in parts]).rv()


def chain_tile_trim_filter_one_contig(job, split_file_id, reference_event_name, params):
    """:660-727 -- chain | tile | trim | filter --maxTileLevel 1 | chain | filter --minChainScore on one part.  With
    outputSecondaryAlignments the reference's second branch: what `filter --maxTileLevel 1 --invert` finds in the filtered file,
    then the primaries that keep their chain score, then the demoted ones relabelled tp:A:S / tl:i:2.  With MIBLAST_INPROCESS=1 the
    job is one mipaf_chain_tile_trim_filter call (one device context instead of one per piped process); the bytes are the same
    either way (tests/test_zz_chain_gpu.py)."""
    scratch = job.fileStore.getLocalTempDir()
    source = os.path.join(scratch, 'input.paf')
    result = os.path.join(scratch, 'output.paf')
    job.fileStore.readGlobalFile(split_file_id, source)
    secondary = int(_blast_attrib(params, "outputSecondaryAlignments")) != 0

    if os.environ.get("MIBLAST_INPROCESS") == "1":
        from cactus_amd import miblast, mipaf
        ctx = miblast.Context(0)
        try:
            cp = mipaf.default_chain_params(max_gap_length=int(_blast_attrib(params, "chainMaxGapLength")), gap_open=int(_blast_attrib(params, "chainGapOpen")),
                                            gap_extend=int(_blast_attrib(params, "chainGapExtend")), trim_fraction=float(_blast_attrib(params, "chainTrimFraction")))
            s = mipaf.PafSet.from_file(source)
            s.chain_tile_trim_filter(ctx, cp, _blast_attrib(params, "pafTrimIdentity"), int(_blast_attrib(params, "minPrimaryChainScore")),
                                     output_secondary=secondary)
            s.write(result)
            s.close()
        finally:
            ctx.close()
    else:
        cmd = _paffy_commands(params)
        first_pass = [cmd['chain'] + ['--inputFile', source], cmd['tile'], cmd['trim'], cmd['primary']]
        if not secondary:
            _pipe_to(job, first_pass + [cmd['chain'], cmd['score']], result)
        else:
            filtered = os.path.join(scratch, 'filter.paf')
            rechained = os.path.join(scratch, 'primary_chain.paf')
            _pipe_to(job, first_pass, filtered)
            _pipe_to(job, [cmd['primary'] + ['--inputFile', filtered, '--invert']], result)
            _pipe_to(job, [cmd['chain'] + ['--inputFile', filtered]], rechained)
            _pipe_to(job, [cmd['score'] + ['--inputFile', rechained]], result, append=True)
            cactus_call(parameters=[cmd['score'] + ['--inputFile', rechained, '--invert'], ['sed', 's/tp:A:P/tp:A:S/'], ['sed', 's/tl:i:1/tl:i:2/']],
                        outfile=result, outappend=True)

    processed = job.fileStore.writeGlobalFile(result)
    job.fileStore.deleteGlobalFile(split_file_id)
    return processed


def merge_processed_alignments(job, processed_file_ids):
    """:730-737 -- the per-part outputs end to end."""
    final = os.path.join(job.fileStore.getLocalTempDir(), 'final.paf')
    concat_global_files(job, processed_file_ids, final)
    return job.fileStore.writeGlobalFile(final)


# ---- the genomes cut down to what is aligned (:861-904; cactus_progressive.py:182 calls it when outgroups are trimmed for the next node) -------
def trim_unaligned_sequences(job, sequences, alignments, params, has_resources=False):
    """-> ([trimmed sequence file ids], trimmed alignment file id).  Without resources of its own the job re-issues itself as a child with
    disk / memory for `paffy to_bed --includeInverted`'s two bytes per base of every sequence plus the alignments (:865-869)."""
    if not has_resources:
        need = 4 * sum(seq.size for seq in sequences) + 2 * alignments.size
        return job.addChildJobFn(trim_unaligned_sequences, sequences, alignments, params, has_resources=True,
                                 disk=need, memory=cactus_clamp_memory(need)).rv()
    scratch = job.fileStore.getLocalTempDir()
    paf = os.path.join(scratch, 'alignments.paf')
    job.fileStore.readGlobalFile(alignments, paf)
    log = ["--logLevel", getLogLevelString()]
    bed = paf + '.bed'                                     # what the alignments cover, either side (binary: covered or not)
    cactus_call(parameters=['paffy', 'to_bed', "--binary", "--excludeUnaligned", "--includeInverted", '-i', paf] + log,
                outfile=bed, returnStdErr=True, job_memory=job.memory)
    flank = _blast(params).attrib["trimOutgroupFlanking"]
    kept = []
    for k, sequence in enumerate(sequences):               # every genome reduced to the covered stretches (+ flank), sub-sequence names NAME|LEN|START
        fasta = os.path.join(scratch, '{}.fa'.format(k))
        job.fileStore.readGlobalFile(sequence, fasta)
        cactus_call(parameters=['faffy', 'extract', "-i", bed, fasta, "--skipMissing", "--minSize", "1", "--flank", flank] + log,
                    outfile=fasta + '.trim', returnStdErr=True, job_memory=job.memory)
        kept.append(fasta + '.trim')
    cactus_call(parameters=['paffy', 'upconvert', "-i", paf] + log + kept, outfile=paf + '.trim', returnStdErr=True)     # the alignments in the reduced sequences' coordinates
    return [job.fileStore.writeGlobalFile(f) for f in kept], job.fileStore.writeGlobalFile(paf + '.trim')
