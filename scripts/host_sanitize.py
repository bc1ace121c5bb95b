#!/usr/bin/env python3
"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU involved): the text tools (mp_text.cpp, bin/faffy's main), the PAF
reader and the chaining stage's host orchestration with its kernels emulated (tests/emu), the FASTA parser (mb_seq.cpp).  Builds the
instrumented binaries under a temporary directory and drives them with random, mutated and malformed input: every run must end with a
result or a refusal -- never with a sanitizer report or a signal.  --kernels: also the kernels' own sources under the host emulation
(DP evaluator, LDS body, hand-over check, traceback, ungapped kernels, both seed stages, set kernels), instrumented the same way.
   python scripts/host_sanitize.py [n_rounds] [--kernels]"""
import os, random, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, EMU = os.path.join(ROOT, "cactus_amd", "csrc"), os.path.join(ROOT, "tests", "emu")
FLAGS = ["-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + EMU, "-I" + SRC, "-I" + os.path.join(ROOT, "include"),
         "-Wno-unknown-pragmas"]
sys.path.insert(0, ROOT)
os.environ["ASAN_OPTIONS"] = "detect_leaks=0"


def sh(cmd, **kw):
    p = subprocess.run(cmd, capture_output=True, **kw)
    return p


def reported(p):
    return p.returncode < 0 or b"Sanitizer" in p.stderr or b"runtime error" in p.stderr


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
    random.seed(3)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        # ---- builds
        k = os.path.join(d, "emu_kernels.cpp")
        open(k, "w").write(open(os.path.join(SRC, "mp_kernels.hip")).read().replace("extern __shared__ unsigned hist[];", "extern unsigned hist[];"))
        paffy, faffy, fasta = os.path.join(d, "paffy"), os.path.join(d, "faffy"), os.path.join(d, "fasta")
        open(os.path.join(d, "stub.cpp"), "w").write("namespace mb { void chain_cache_destroy(void *) {} }\n")
        open(os.path.join(d, "fuzz_fasta.cpp"), "w").write(r'''
#include "mb_common.h"
#include <random>
#include <string>
#include <cstdio>
namespace mb { void set_error(const std::string &) {} int parse_fasta(const char *buf, size_t len, SeqSet &out); }
int main() {
    std::mt19937 rng(7);
    const char alpha[] = "ACGTacgtNnRYKMSWBDHVryxX-*> \t\r\n\n\n>>;|0123";
    for (int it = 0; it < 20000; it++) {
        std::string s;
        const int n = rng() % 400;
        if (rng() % 3) s += ">";
        for (int i = 0; i < n; i++) s += (rng() % 50 == 0) ? (char)(rng() % 256) : alpha[rng() % (sizeof(alpha) - 1)];
        if (rng() % 4 == 0) s += "\n>last";
        mb::SeqSet S;
        if (mb::parse_fasta(s.data(), s.size(), S) == 0) { long long tot = 0; for (size_t k = 0; k < S.lens.size(); k++) tot += S.lens[k]; if (!S.codes.empty() && tot > (long long)S.codes.size()) return 1; }
    }
    return 0;
}''')
        for out, srcs in ((paffy, [k, os.path.join(SRC, "mp_chain.cpp"), os.path.join(SRC, "mp_text.cpp"), os.path.join(SRC, "mp_paffy_main.cpp"), os.path.join(EMU, "emu_runtime.cpp"),
                                   os.path.join(EMU, "emu_launch.cpp")]),
                          (faffy, [os.path.join(SRC, "mp_faffy_main.cpp"), os.path.join(SRC, "mp_text.cpp"), os.path.join(EMU, "emu_runtime.cpp"), os.path.join(d, "stub.cpp")]),
                          (fasta, [os.path.join(d, "fuzz_fasta.cpp"), os.path.join(SRC, "mb_seq.cpp")])):
            p = sh(["g++", *FLAGS, "-o", out, *srcs])
            if p.returncode:
                print(p.stderr.decode()[-2000:]); return 1
        # ---- FASTA parser
        p = sh([fasta])
        if p.returncode or reported(p):
            print("FASTA parser:", p.stderr.decode()[:1500]); bad += 1
        # ---- PAF reader + commands on mutated records
        good = "q1\t1000\t10\t200\t+\tt1\t5000\t100\t290\t180\t190\t255\tAS:i:1500\tcg:Z:100=2X88=\n"

        def mutate(line):
            b = bytearray(line.encode())
            for _ in range(random.randint(1, 6)):
                at, r = random.randrange(len(b)), random.random()
                if r < 0.3: b[at] = random.randrange(256)
                elif r < 0.6: del b[at]
                else: b.insert(at, random.choice(b"\t0123456789=XIDM-+:\n "))
            return bytes(b)
        cmds = (("invert", []), ("chain", ["--maxGapLength", "1000000", "--chainGapOpen", "5000", "--chainGapExtend", "1", "--trimFraction", "1.0"]),
                ("to_bed", ["--binary", "--excludeAligned"]), ("tile", []), ("trim", ["--trimIdentity", "0.9"]))
        done = refused = 0
        for _ in range(n):
            text = b"".join(mutate(good) if random.random() < 0.7 else good.encode() for _ in range(random.randint(1, 6)))
            for cmd, args in cmds:
                p = sh([paffy, cmd, *args], input=text, env=dict(os.environ, MIPAF_CHAIN_THREADS="64"))
                if reported(p):
                    print("paffy", cmd, p.stderr.decode(errors="replace")[:1200]); bad += 1
                done += p.returncode == 0; refused += p.returncode != 0
        print("paffy on mutated PAF: %d results, %d refusals" % (done, refused))
        # ---- faffy chunk / extract against the product's binary
        import numpy as np
        from cactus_amd import gen
        for seed in range(max(3, n // 40)):
            rng = np.random.default_rng(seed)
            recs = [("c%d extra words" % i, gen.random_sequence(int(rng.integers(1, 5000)), rng)) for i in range(int(rng.integers(1, 14)))]
            fa = os.path.join(d, "g%d.fa" % seed)
            open(fa, "wb").write(gen.fasta_bytes(recs))
            c, o = str(int(rng.integers(500, 4000))), str(int(rng.integers(0, 300)))
            outs = []
            for exe in (faffy, os.path.join(ROOT, "bin", "faffy")):
                od = os.path.join(d, "chunks_%d_%d" % (seed, len(outs))); os.makedirs(od)
                p = sh([exe, "chunk", "-c", c, "-o", o, "--dir", od, fa])
                if p.returncode or reported(p):
                    print("faffy chunk", p.stderr.decode()[:800]); bad += 1
                outs.append({x: open(os.path.join(od, x), "rb").read() for x in sorted(os.listdir(od))})
            bad += outs[0] != outs[1]
            bed = os.path.join(d, "x%d.bed" % seed)
            open(bed, "w").write("".join("c%d\t%d\t%d\n" % (i, int(rng.integers(0, max(1, len(recs[i][1]) // 2))), len(recs[i][1])) for i in range(len(recs))) + "missing\t0\t10\n")
            ex = []
            for exe in (faffy, os.path.join(ROOT, "bin", "faffy")):
                p = sh([exe, "extract", "-i", bed, fa, "--flank", "10", "--minSize", "1", "--skipMissing"])
                if p.returncode or reported(p):
                    print("faffy extract", p.stderr.decode()[:800]); bad += 1
                ex.append(p.stdout)
            bad += ex[0] != ex[1]
        # ---- the kernels' own sources under the host emulation (tests/emu), instrumented: out-of-bounds accesses of the rings, planes, snapshots
        #      and trace blocks, shifts and overflows (unaligned 4- and 8-byte loads are meant: -fno-sanitize=alignment)
        if "--kernels" in sys.argv:
            for name, runs in (("emu_ydrop", (["3", "2"], ["5", "3", "relay"])), ("emu_ungapped", (["3", "2", "ux"], ["9", "2", "lane"], ["5", "2"])), ("emu_seed_dense", (["7", "6"],)),
                               ("emu_seed_batch", (["11", "2"],)), ("emu_sets", (["5", "20"],))):
                exe = os.path.join(d, name)
                p = sh(["g++", *FLAGS, "-fno-sanitize=alignment", "-Wno-attributes", "-o", exe, os.path.join(EMU, name + ".cpp"), os.path.join(EMU, "emu_launch.cpp")])
                if p.returncode:
                    print(p.stderr.decode()[-2000:]); return 1
                for args in runs:
                    p = sh([exe, *args])
                    if p.returncode or reported(p) or b"MISMATCH" in p.stdout:
                        print(name, args, p.stdout.decode()[-600:], p.stderr.decode()[:1500]); bad += 1
                print(name, "instrumented: ok" if not bad else "instrumented: see above")
    print("host_sanitize: %d problem(s)" % bad)
    return 1 if bad else 0


sys.exit(main())
