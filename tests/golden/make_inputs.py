"""Regenerates the golden INPUTS (G1, G2, G3 of SURVEY.md Appendix A) with CPython's `random`.

The expected OUTPUTS are the reference program's own results: the survey session ran upstream
`sbwt build` / `sbwt map` on exactly these inputs and recorded the md5 sums kept in golden.json
(`ref_*` keys).  g1.sam / g2.sam in this directory are byte-identical to those outputs (same md5).
G3 (4.6 Mbp genome, 1 M reads, 108 MB of FASTQ) is too large to commit; the CPU test that uses it
regenerates it with this script.   usage: make_inputs.py OUTDIR [g1] [g2] [g3]
"""
import hashlib
import json
import os
import random
import sys

COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


def g1(out):
    random.seed(1)
    seqs = {}
    for name, n in (("chrA", 20000), ("chrB", 15000)):
        s = "".join(random.choice("ACGT") for _ in range(n))
        if name == "chrA":
            s = s[:5000] + "N" * 137 + s[5137:]
        seqs[name] = s
    with open(os.path.join(out, "g1.fa"), "w") as f:
        for name in ("chrA", "chrB"):
            f.write(">" + name + "\n")
            s = seqs[name]
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    with open(os.path.join(out, "g1.fq"), "w") as f:
        for i in range(20):
            name = random.choice(["chrA", "chrB"])
            s = seqs[name]
            while True:
                p = random.randrange(0, len(s) - 50)
                r = s[p:p + 50]
                if "N" not in r:
                    break
            st = random.choice("+-")
            if st == "-":
                r = rc(r)
            q = "".join(chr(33 + (j % 40)) for j in range(50))
            f.write("@r%d_%s_%d_%s\n%s\n+\n%s\n" % (i, name, p + 1, st, r, q))


def g2(out):
    random.seed(7)
    rep = "".join(random.choice("ACGT") for _ in range(300))
    a = "".join(random.choice("ACGT") for _ in range(3000))
    b = "".join(random.choice("ACGT") for _ in range(3000))
    chr1 = a[:1000] + rep + a[1000:2000] + rep + a[2000:] + "N" * 10 + rc(rep)[:120]
    chr2 = "N" * 5 + b[:1500] + rep[:100] + b[1500:]
    with open(os.path.join(out, "g2.fa"), "w") as f:
        f.write(">chr1\n" + chr1 + "\n>chr2\n" + chr2.lower() + "\n")
    reads = [("multi_rep40", rep[50:90]), ("multi_rep30rc", rc(rep[10:40])), ("len12", a[100:112]),
             ("len13", a[100:113]), ("nohit", "ACGT" * 10), ("span_chr", chr1[-20:] + b[:20]),
             ("polyT_start", "T" * 12 + a[300:320]), ("uniq_b", b[700:750])]
    with open(os.path.join(out, "g2.fq"), "w") as f:
        for n, s in reads:
            f.write("@%s extra\n%s\n+\n%s\n" % (n, s, "I" * len(s)))


def g3(out):
    random.seed(4600000)
    n = 4641652
    s = "".join(random.choices("ACGT", k=n))
    with open(os.path.join(out, "g3.fa"), "w") as f:
        f.write(">synEcoli\n")
        for i in range(0, n, 70):
            f.write(s[i:i + 70] + "\n")
    random.seed(50)
    rows = []
    for i in range(1000000):
        p = random.randrange(0, n - 50)
        r = s[p:p + 50]
        if random.random() < 0.5:
            r = rc(r)
        rows.append("@r%d\n%s\n+\n%s\n" % (i, r, "I" * 50))
    with open(os.path.join(out, "g3.fq"), "w") as f:
        f.write("".join(rows))


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


if __name__ == "__main__":
    out = sys.argv[1]
    which = sys.argv[2:] or ["g1", "g2"]
    os.makedirs(out, exist_ok=True)
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")))
    for g in which:
        {"g1": g1, "g2": g2, "g3": g3}[g](out)
        for ext in ("fa", "fq"):
            got = md5(os.path.join(out, "%s.%s" % (g, ext)))
            want = gold[g]["input_%s_md5" % ext]
            print(g, ext, got, "OK" if got == want else "MISMATCH (expected %s)" % want)
