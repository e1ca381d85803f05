import gzip
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_fasta(path):
    op = gzip.open if path.endswith(".gz") else open
    names, seqs = [], []
    with op(path, "rt") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                names.append(line[1:].split()[0])
                seqs.append([])
            else:
                seqs[-1].append(line)
    return names, ["".join(s) for s in seqs]


def load_golden(name):
    p = os.path.join(GOLDEN, name)
    op = gzip.open if p.endswith(".gz") else open
    with op(p, "rt") as f:
        return json.load(f)


def rows_to_lists(rows):
    """PafRow -> plain list (JSON-comparable): the bit-exact contract of SURVEY.md section 8b."""
    return [[r.qname, r.qlen, r.qs, r.qe, r.strand, r.tname, r.tlen, r.rs, r.re, r.mlen, r.blen, r.mapq, r.AS, repr(r.de), r.cg, r.n_ambi, r.inv]
            for r in rows]


def plasmid_names(n):
    # decimal BlockId-like names whose strcmp order differs from their numeric order
    return [str(1000 + i * 7919) for i in range(n)]
