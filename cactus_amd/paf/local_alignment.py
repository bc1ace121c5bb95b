"""Host side of the blast phase: the Toil job functions of
/root/reference/src/cactus/paf/local_alignment.py that sit directly on the hot path, kept with the
same names, arguments, return values and error behaviour so the parity tests read like the
reference's callers:

    run_lastz                 :29-97    one chunk-pair job  -> PAF file id
    make_chunked_alignments   :370-408  chunk both genomes, all-vs-all run_lastz, then combine
    combine_chunks            :336-356  dechunk + concatenate
    chain_alignments          :607-657  merge, add the inverted copies, split by query contig when large
    chain_tile_trim_filter_one_contig :660-727  paffy chain | tile | trim | filter | chain | filter
    trim_unaligned_sequences  :861-904  paffy to_bed | faffy extract | paffy upconvert (second half)

What differs, on purpose: the `lastz` / `run_kegalign` found on PATH are the MI355X front ends in
<repo>/bin (or, with MIBLAST_INPROCESS=1, the same code through the C ABI via ctypes), AMD GPUs are
requested as 'rocm:N', faffy and paffy's chunk/dechunk are replaced by cactus_amd.paf.chunking, and the `paffy`
of the chaining stage is <repo>/bin/paffy (chain, tile and trim run on the GPU; with MIBLAST_INPROCESS=1 the whole
per-contig job is one call through include/mipaf.h).  cactus_consolidated and everything after it are untouched.
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


def select_lastz_params(distance, params, gpu):
    """distance -> lastz option string, exactly the rule of local_alignment.py:44-51: the first of
    one..five whose divergence bound is >= distance, else "default" (or always default if useDefault)."""
    lastz_params_node = params.find("blast")
    lastz_divergence_node = lastz_params_node.find("kegalignArguments" if gpu else "lastzArguments")
    divergences = params.find("constants").find("divergences")
    lastz_params = lastz_divergence_node.attrib["default"]
    if not getOptionalAttrib(divergences, 'useDefault', typeFn=bool, default=False):
        for i in "one", "two", "three", "four", "five":
            if distance <= float(divergences.attrib[i]):
                lastz_params = lastz_divergence_node.attrib[i]
                break
    return lastz_params


def run_lastz(job, name_A, genome_A, name_B, genome_B, distance, params):
    work_dir = job.fileStore.getLocalTempDir()
    alignment_file = os.path.join(work_dir, '{}_{}.paf'.format(name_A, name_B))
    genome_a_file = os.path.join(work_dir, '{}.fa'.format(name_A))
    genome_b_file = os.path.join(work_dir, '{}.fa'.format(name_B))
    job.fileStore.readGlobalFile(genome_A, genome_a_file)
    job.fileStore.readGlobalFile(genome_B, genome_b_file)

    lastz_params_node = params.find("blast")
    gpu = getOptionalAttrib(lastz_params_node, 'gpu', typeFn=int, default=0)
    cpu = getOptionalAttrib(lastz_params_node, 'cpu', typeFn=int, default=None)
    lastz_params = select_lastz_params(distance, params, gpu)
    if gpu:
        lastz_bin = 'run_kegalign'
        suffix_a, suffix_b = '', ''
        assert gpu > 0
        lastz_params += ' --num_gpu {} --num_threads {}'.format(gpu, job.cores)
    else:
        lastz_bin = 'lastz'
        suffix_a = '[multiple][nameparse=darkspace]'
        suffix_b = '[nameparse=darkspace]'

    lastz_cmd = [lastz_bin,
                 '{}{}'.format(os.path.basename(genome_a_file), suffix_a),
                 '{}{}'.format(os.path.basename(genome_b_file), suffix_b),
                 '--format=paf:wfmash'] + lastz_params.split(' ')

    if os.environ.get("MIBLAST_INPROCESS") == "1":
        messages = _run_inprocess(lastz_cmd, work_dir, alignment_file)
    else:
        messages = cactus_call(parameters=lastz_cmd, outfile=alignment_file, work_dir=work_dir, returnStdErr=True,
                               gpus=gpu, cpus=cpu, job_memory=job.memory)

    if gpu:
        # same guard as local_alignment.py:75-83 -- our front end keeps stderr empty on success
        for line in (messages or "").lower().split("\n"):
            if not line.startswith("signals delivered"):
                for keyword in STDERR_KEYWORDS:
                    if keyword in line and 'signals' not in line:
                        job.fileStore.logToMaster("KegAlign offending line: " + line)
                        raise RuntimeError('{} exited 0 but keyword "{}" found in stderr'.format(lastz_cmd, keyword))
    return job.fileStore.writeGlobalFile(alignment_file)


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
    if len(chunked_alignment_files) >= 2 * batch_size:
        batch_results = []
        for chunk_idx in range(math.ceil(len(chunked_alignment_files) / batch_size)):
            batch = chunked_alignment_files[chunk_idx * batch_size: chunk_idx * batch_size + batch_size]
            batch_results.append(job.addChildJobFn(combine_chunks, batch, batch_size).rv())
        return job.addFollowOnJobFn(merge_combined_chunks, batch_results).rv()
    alignment_file = job.fileStore.getLocalTempFile()
    for chunk in chunked_alignment_files:
        chunking.paf_dechunk(job.fileStore.readGlobalFile(chunk), alignment_file, append=True)
        job.fileStore.deleteGlobalFile(chunk)
    return job.fileStore.writeGlobalFile(alignment_file)


def merge_combined_chunks(job, combined_chunks):
    output_path = job.fileStore.getLocalTempFile()
    with open(output_path, 'a') as output_file:
        for chunk in combined_chunks:
            with open(job.fileStore.readGlobalFile(chunk, mutable=True), 'r') as chunk_file:
                output_file.write(chunk_file.read())
            job.fileStore.deleteGlobalFile(chunk)
    return job.fileStore.writeGlobalFile(output_path)


def make_chunked_alignments(job, event_a, genome_a, event_b, genome_b, distance, params):
    lastz_params_node = params.find("blast")
    gpu = getOptionalAttrib(lastz_params_node, 'gpu', typeFn=int, default=0)
    lastz_cores = getOptionalAttrib(lastz_params_node, 'cpu', typeFn=int, default=None)
    lastz_memory = getOptionalAttrib(lastz_params_node, 'lastz_memory', typeFn=int, default=None)
    chunk_attr = 'bigChunkSize' if gpu else 'chunkSize'

    def make_chunks(genome):
        output_chunks_dir = job.fileStore.getLocalTempDir()
        chunk_files = chunking.fasta_chunk(job.fileStore.readGlobalFile(genome), output_chunks_dir,
                                           int(params.find("blast").attrib[chunk_attr]),
                                           int(params.find("blast").attrib["overlapSize"]))
        return [job.fileStore.writeGlobalFile(chunk, cleanup=True) for chunk in chunk_files]

    chunks_a = make_chunks(genome_a)
    chunks_b = make_chunks(genome_b)
    accelerators = accelerator_string(gpu)
    chunked_alignment_files = []
    for i, chunk_a in enumerate(chunks_a):
        for j, chunk_b in enumerate(chunks_b):
            memory = lastz_memory if lastz_memory else max(200000000, 15 * (chunk_a.size + chunk_b.size))
            chunked_alignment_files.append(job.addChildJobFn(run_lastz, '{}_{}'.format(event_a, i), chunk_a,
                                                             '{}_{}'.format(event_b, j), chunk_b, distance, params,
                                                             cores=lastz_cores,
                                                             disk=max(4 * (chunk_a.size + chunk_b.size), memory),
                                                             memory=memory, accelerators=accelerators).rv())
    dechunk_batch_size = getOptionalAttrib(lastz_params_node, 'dechunkBatchSize', typeFn=int, default=int(1e9))
    return job.addFollowOnJobFn(combine_chunks, chunked_alignment_files, dechunk_batch_size).rv()


# ---- ingroup -> outgroup alignments with progressive trimming (local_alignment.py:411-526) -----------------------------
def invert_alignments(job, alignment_file):
    """ Invert the pafs in the alignment_file (paffy invert, :411-418) """
    src = job.fileStore.readGlobalFile(alignment_file)
    dst = job.fileStore.getLocalTempFile()
    with open(src) as fin, open(dst, "w") as fout:
        for line in fin:
            if line.strip():
                fout.write(chunking.paf_invert_line(line))
    job.fileStore.deleteGlobalFile(alignment_file)
    return job.fileStore.writeGlobalFile(dst)


def make_ingroup_to_outgroup_alignments_0(job, ingroup_event, outgroup_events, event_names_to_sequences, distances, params):
    alignment_file = job.addChildJobFn(make_ingroup_to_outgroup_alignments_1, ingroup_event, outgroup_events,
                                       event_names_to_sequences, distances, params).rv()
    # Invert the final alignment so that the query is the outgroup and the target is the ingroup
    return job.addFollowOnJobFn(invert_alignments, alignment_file).rv()


def make_ingroup_to_outgroup_alignments_1(job, ingroup_event, outgroup_events, event_names_to_sequences, distances, params):
    """events are plain names here (the reference passes tree nodes and uses .iD); distances is keyed by (ingroup, outgroup)"""
    outgroup = outgroup_events[0]
    alignment = job.addChildJobFn(make_chunked_alignments, outgroup, event_names_to_sequences[outgroup],
                                  ingroup_event, event_names_to_sequences[ingroup_event], distances[ingroup_event, outgroup], params).rv()
    if len(outgroup_events) > 1:
        return job.addFollowOnJobFn(make_ingroup_to_outgroup_alignments_2, alignment, ingroup_event, outgroup_events[1:],
                                    dict(event_names_to_sequences), distances, params).rv()
    return alignment


def make_ingroup_to_outgroup_alignments_2(job, alignments, ingroup_event, outgroup_events, event_names_to_sequences, distances, params):
    # identify all ingroup sub-sequences that remain unaligned longer than a threshold (:451-475)
    work_dir = job.fileStore.getLocalTempDir()
    alignments_file = os.path.join(work_dir, '{}.paf'.format(ingroup_event))
    job.fileStore.readGlobalFile(alignments, alignments_file)
    ingroup_seq_file = os.path.join(work_dir, '{}.fa'.format(ingroup_event))
    job.fileStore.readGlobalFile(event_names_to_sequences[ingroup_event], ingroup_seq_file)
    bed = chunking.paf_to_bed_unaligned(alignments_file, ingroup_seq_file, int(params.find("blast").attrib['trimMinSize']))
    seq_file = os.path.join(work_dir, '{}_subseq.fa'.format(ingroup_event))
    chunking.fasta_extract(bed, ingroup_seq_file, seq_file, int(params.find("blast").attrib['trimFlanking']))
    # replace the ingroup sequences with remaining sequences and recurse over the remaining outgroups
    event_names_to_sequences[ingroup_event] = job.fileStore.writeGlobalFile(seq_file)
    if os.path.getsize(seq_file) == 0:
        return alignments
    alignments2 = job.addChildJobFn(make_ingroup_to_outgroup_alignments_1, ingroup_event, outgroup_events,
                                    event_names_to_sequences, distances, params).rv()
    return job.addFollowOnJobFn(make_ingroup_to_outgroup_alignments_3, ingroup_event, event_names_to_sequences[ingroup_event],
                                alignments, alignments2).rv()


def make_ingroup_to_outgroup_alignments_3(job, ingroup_event, ingroup_seq_file, alignments, alignments2, has_resources=False):
    # use paffy dechunk --query to correct the subsequence coordinates of alignments2, then cat (:515-520)
    a1 = job.fileStore.readGlobalFile(alignments)
    a2 = job.fileStore.readGlobalFile(alignments2)
    merged = job.fileStore.getLocalTempFile()
    with open(merged, "w") as out, open(a1) as f1:
        out.write(f1.read())
    chunking.paf_dechunk(a2, merged, query_only=True, append=True)
    job.fileStore.deleteGlobalFile(ingroup_seq_file)
    return job.fileStore.writeGlobalFile(merged)


def trim_unaligned_sequences(job, sequences, alignments, params, has_resources=False):
    """:861-904 -- the genomes cut down to what the blast alignments cover (+ trimOutgroupFlanking), and the alignments rewritten to
    those sub-sequences.  The three tools run as the reference runs them -- bin/paffy to_bed / upconvert and bin/faffy extract are the
    native text code of libmiblast (mp_text.cpp); the same steps in process: cactus_amd.paf.chunking.trim_to_aligned."""
    work_dir = job.fileStore.getLocalTempDir()
    alignments_file = os.path.join(work_dir, 'alignments.paf')
    job.fileStore.readGlobalFile(alignments, alignments_file)
    bed_file = alignments_file + '.bed'
    cactus_call(parameters=['paffy', 'to_bed', "--binary", "--excludeUnaligned", "--includeInverted",
                            '-i', alignments_file, "--logLevel", getLogLevelString()], outfile=bed_file, returnStdErr=True, job_memory=job.memory)
    trimmed_sequence_files = []
    for i, sequence in enumerate(sequences):
        seq_file = os.path.join(work_dir, '{}.fa'.format(i))
        job.fileStore.readGlobalFile(sequence, seq_file)
        trimmed_seq_file = seq_file + '.trim'
        cactus_call(parameters=['faffy', 'extract', "-i", bed_file, seq_file, "--skipMissing", "--minSize", "1",
                                "--flank", params.find("blast").attrib["trimOutgroupFlanking"], "--logLevel", getLogLevelString()],
                    outfile=trimmed_seq_file, returnStdErr=True, job_memory=job.memory)
        trimmed_sequence_files.append(trimmed_seq_file)
    trimmed_alignments = alignments_file + '.trim'
    cactus_call(parameters=['paffy', 'upconvert', "-i", alignments_file, "--logLevel", getLogLevelString()] + trimmed_sequence_files,
                outfile=trimmed_alignments, returnStdErr=True)
    return [job.fileStore.writeGlobalFile(i) for i in trimmed_sequence_files], job.fileStore.writeGlobalFile(trimmed_alignments)


# ---- chaining stage (local_alignment.py:594-734): chain -> tile -> trim -> filter -> chain -> filter ---------------------------
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
    work_dir = job.fileStore.getLocalTempDir()
    merged = os.path.join(work_dir, 'merged.paf')
    concat_global_files(job, alignment_files, merged)
    if include_inverted_alignments:
        snapshot = os.path.join(work_dir, 'merged_copy.paf')           # invert reads a copy while its output is appended to the original
        shutil.copyfile(merged, snapshot)
        _pipe_to(job, ['paffy', 'invert', '--inputFile', snapshot], merged, append=True)
        os.remove(snapshot)

    def one_job(path):
        size = os.path.getsize(path)
        return job.addChildJobFn(chain_tile_trim_filter_one_contig, job.fileStore.writeGlobalFile(path), reference_event_name, params,
                                 disk=4 * size, memory=cactus_clamp_memory(4 * size)).rv()

    if os.path.getsize(merged) <= int(_blast_attrib(params, "chainSplitMinSize", "1000000000")):
        return one_job(merged)
    prefix = os.path.join(work_dir, 'split_')
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
    work_dir = job.fileStore.getLocalTempDir()
    source = os.path.join(work_dir, 'input.paf')
    result = os.path.join(work_dir, 'output.paf')
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
            filtered = os.path.join(work_dir, 'filter.paf')
            rechained = os.path.join(work_dir, 'primary_chain.paf')
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
