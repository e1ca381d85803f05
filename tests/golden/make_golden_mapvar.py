"""Writes tests/golden/nuc_matrix.json: the letter order of the reference's Nuc alphabet and its 16 x 16 match table, read from
/root/reference/packages/pangraph/src/align/nextclade/{alphabet/nuc.rs:10-30, align/score_matrix_nuc.rs:6-26} (data, not source:
256 zeros and ones and sixteen letters).  Run in the build container (the reference is not on the GPU box)."""
import json
import os
import re

R = "/root/reference/packages/pangraph/src/align/nextclade"
t = open(os.path.join(R, "align/score_matrix_nuc.rs")).read()
body = re.sub(r"/\*.*?\*/", "", t[t.index("= &["):t.index("];")], flags=re.S)
v = [int(x) for x in re.findall(r"\d+", body)]
assert len(v) == 256
n = open(os.path.join(R, "alphabet/nuc.rs")).read()
arms = re.findall(r"Nuc::(\w+) => '(.)'", n[n.index("pub const fn from_nuc"):])
enum = re.findall(r"^\s+(\w+),?$", n[n.index("pub enum Nuc"):n.index("impl Nuc")], flags=re.M)
names = [e for e in enum if e[0].isupper()]
letter = dict(arms)
letters = "".join(letter[x] for x in names)
assert len(letters) == 16
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nuc_matrix.json")
json.dump({"letters": letters, "matrix": [v[16 * i:16 * i + 16] for i in range(16)]}, open(out, "w"))
print(letters)
