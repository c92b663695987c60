import sys, json, numpy as np
sys.path.insert(0, '.')
from badger_amd import _native, synth
from oracle import pyoracle as orc
ctx = _native.Context(0)
ext = json.load(open('tests/golden/extract_rows.json'))
seqs = [r['seq'] for r in ext['reads']]
for lo, hi in ((48, 56), (51, 52), (50, 52), (49, 52), (51, 53), (40, 60)):
    bases, off = synth.list_to_reads(seqs[lo:hi])
    got = ctx.extract_batch(bases, off, 12)
    want = orc.extract_batch(bases, off, 12, threads=4)
    bad = np.nonzero(got != want)[0]
    print(lo, hi, 'lens', [len(s) for s in seqs[lo:hi]][:8], 'mismatch idx', bad.tolist(), [(int(got[b]['polyT']), int(want[b]['polyT'])) for b in bad])
s = seqs[51]
print(s[60:130])
