"""CPU-side edge-case fuzz of the chaining stage: degenerate PAF records (empty cigars, gap-only and mismatch-only cigars, zero-length
intervals, duplicates and pile-ups, negative / missing / equal scores, both strands) through the HOST build of the product's sources
(tests/emu/emu_paffy: emulated work-groups) against oracle/oracle_paffy, for chain, tile (both back ends) and trim.
usage: python scripts/emu_edge_fuzz.py [cases] [first_seed]      (needs `make -C tests/emu` and `make -C oracle`)"""
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "emu_paffy")
ORACLE = os.path.join(ROOT, "oracle", "oracle_paffy")


def run(exe, cmd, text, *args):
    p = subprocess.run([exe, cmd, *args], input=text.encode(), capture_output=True, env=dict(os.environ, MIPAF_CHAIN_THREADS="128"))
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def record(rng, length=5000):
    qn, tn, strand = rng.choice(["q1", "q2"]), rng.choice(["t1", "t2", "q1"]), rng.choice("+-")
    kind = rng.random()
    ops = []
    if kind < 0.1:
        ops = []                                             # empty cigar, zero-length alignment
    elif kind < 0.2:
        ops = [(rng.randint(1, 30), rng.choice("ID"))]
    elif kind < 0.3:
        ops = [(rng.randint(1, 30), "X")]
    else:
        for _ in range(rng.randint(1, 6)):
            o = rng.choice("==XIDM")
            if not ops or ops[-1][1] != o:
                ops.append((rng.randint(1, 60), o))
    qspan = sum(n for n, o in ops if o != "D")
    tspan = sum(n for n, o in ops if o != "I")
    qs = rng.choice([0, rng.randint(0, length - qspan), length - qspan])
    ts = rng.choice([0, rng.randint(0, length - tspan), length - tspan])
    if rng.random() < 0.3:
        qs, ts = 100, 200                                    # pile-ups and duplicates
    c = [qn, length, qs, qs + qspan, strand, tn, length, ts, ts + tspan, sum(n for n, o in ops if o in "=M"), sum(n for n, _ in ops), 255]
    if rng.random() < 0.8:
        c.append("AS:i:%d" % rng.choice([0, -5, 1, 100, 100, 5000, 12000, rng.randint(-50, 20000)]))
    if rng.random() < 0.9:
        c.append("cg:Z:" + "".join("%d%s" % (n, o) for n, o in ops))
    return "\t".join(str(x) for x in c) + "\n"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for case in range(first, first + n):
        rng = random.Random(case)
        text = "".join(record(rng) for _ in range(rng.randint(1, 25)))
        cargs = ["--maxGapLength", rng.choice(["0", "50", "1000000"]), "--chainGapOpen", rng.choice(["0", "10", "5000"]),
                 "--chainGapExtend", rng.choice(["0", "1"]), "--trimFraction", rng.choice(["0", "0.5", "1.0"])]
        x = rng.choice(["0", "0.2", "0.5", "1"])
        rc, chained, err = run(ORACLE, "chain", text, *cargs)
        assert rc == 0, err
        rc, tiled, err = run(ORACLE, "tile", chained)
        assert rc == 0, err
        for cmd, inp, oargs, extra in (("chain", text, cargs, []), ("tile", chained, [], []), ("tile", chained, [], ["--mipaf-hist-bins", "2"]),
                                       ("trim", tiled, ["--trimIdentity", x], [])):
            want, got = run(ORACLE, cmd, inp, *oargs), run(EMU, cmd, inp, *oargs, *extra)
            if want[0] != 0 or got[0] != 0 or want[1] != got[1]:
                bad += 1
                print(f"case {case}: {cmd} {oargs + extra}: rc {want[0]} / {got[0]} {got[2][:200]}", flush=True)
                break
    print(f"{n} cases, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
