#!/usr/bin/env python3
"""Round 6, VERDICT item 3: would six block-phase hypotheses in round 0 of k_jpeg_huff shorten its chain of synchronisation
rounds?  On the model of tools/sim/huff_model.py, per fixture:
  base      the kernel as it is: rounds after the first pass, segment decodes repeated
  hyp       per segment: does ANY of the bpm hypotheses (entry = segment start, block h, k = 0) end in the true exit state;
            does the one whose h is the TRUE entry's block (what "select by the predecessor's exit state" can know)
  A         hypotheses + selection by the predecessor's exit block, one verifying pass, then the kernel's rounds
  C         hypotheses + one pass from every DISTINCT exit of the predecessor's hypotheses (exact maps candidate -> candidate,
            composable by a scan) -- what is left are the segments whose true entry is no candidate: each costs one
            sequential decode, "longest run" of them in a row
  steps     decode steps of the scan for first-level tables of 10 .. 13 bits with up to 2 / 3 symbols per entry
    python3 tools/sim/huff_hypotheses.py [files, default 10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from huff_model import FILES, Dec, segments, simulate, truth

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for f in FILES[:nfiles]:
    d = Dec(open(f, 'rb').read())
    S, n = segments(d)
    end = lambda i: min((i + 1) * S, d.nbits + 32)
    tex = truth(d)
    (r0, w0, ex0) = simulate(d)
    assert ex0 == tex
    tent = [(0, 0, 0)] + tex[:-1]
    E = [[d.run(i * S, h, 0, end(i))[0] for h in range(d.bpm)] for i in range(n)]
    any_true = sum(1 for i in range(1, n) if tex[i] in E[i])
    own_true = sum(1 for i in range(1, n) if E[i][tent[i][1]] == tex[i])
    # A: chain the hypotheses by the exit block, verify, then the kernel's rounds
    x = [E[0][0]]
    for i in range(1, n):
        x.append(E[i][x[i - 1][1]])
    entry = [(0, 0, 0)] + x[:-1]
    ex = [E[0][0]] + [d.run(*entry[i], end(i))[0] for i in range(1, n)]
    wa = [n]
    while True:
        ch = [i for i in range(1, n) if ex[i - 1] != entry[i]]
        if not ch:
            break
        wa.append(len(ch))
        new = list(ex)
        for i in ch:
            entry[i] = ex[i - 1]
            new[i] = d.run(*entry[i], end(i))[0]
        ex = new
    assert ex == tex
    # C: exact maps from every distinct candidate entry
    distinct = [len(set(E[i - 1])) for i in range(1, n)]
    M = [None] + [{e: d.run(*e, end(i))[0] for e in set(E[i - 1])} for i in range(1, n)]
    t = E[0][0]
    breaks = run = longest = 0
    for i in range(1, n):
        if t in M[i]:
            t = M[i][t]
            run = 0
        else:
            t = d.run(*t, end(i))[0]
            breaks += 1
            run += 1
            longest = max(longest, run)
        assert t == tex[i]
    st = ' '.join('%d/%d:%d' % (tb, ns, d.steps(tb, ns)[0]) for tb in (10, 11, 12, 13) for ns in (2, 3))
    print('%-34s %3d segments of %4d bits | base: %3d rounds, %4d decodes repeated | hyp: any true %3d, own block true %3d of %d | '
          'A: %3d rounds after the hypotheses (%s ...) | C: second pass %.2f decodes per segment (max %d), then %2d sequential decodes, longest run %d | steps %s'
          % (os.path.basename(f), n, S, r0, sum(w0), any_true, own_true, n - 1, len(wa), ' '.join(map(str, wa[:6])), sum(distinct) / (n - 1), max(distinct),
             breaks, longest, st), flush=True)
