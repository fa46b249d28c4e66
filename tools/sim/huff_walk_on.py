"""Experiment on the CPU model of k_jpeg_huff's rounds (huff_sync_sim.py): the critical path of the rounds -- the sum
over the rounds of the longest lane's symbol count -- as the kernel does it (every changed segment decoded once per
round) and with WALK-ON: the lane that decoded the last segment of a run of changed segments goes on into the next
segments for as long as its exit differs from the stored one and nobody else decodes that segment this round.
    python3 tools/sim/huff_walk_on.py [files]"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from huff_sync_sim import Dec

def run_count(dec, p, blk, k, p_end):
    n = 0; nblk = 0
    while p < p_end:
        td, ta = dec.layout[blk]
        if k == 0:
            l, s = dec.sym(p, 0, td); p += l + s; k = 1
        else:
            l, s = dec.sym(p, 1, ta); r = s >> 4; sz = s & 15
            p += l + sz
            k = (k + 16 if r == 15 else 64) if sz == 0 else k + r + 1
        if k >= 64:
            k = 0; nblk += 1; blk = (blk + 1) % dec.bpm
        n += 1
    return (p, blk, k), n

def simulate(dec, walk, T=512, late=10**9):
    bits = dec.nbits
    S = 32 * (max(8, (bits + 32 * T - 1) // (32 * T)) | 1)
    nseg = (bits + S - 1) // S
    end = lambda i: min((i + 1) * S, bits + 32)
    entry = [(i * S, 0, 0) for i in range(nseg)]
    ex = []; cost0 = 0
    for i in range(nseg):
        e, n = run_count(dec, *entry[i], end(i)); ex.append(e); cost0 = max(cost0, n)
    rounds = 0; path = 0; total = 0
    while True:
        ch = [i for i in range(1, nseg) if ex[i - 1] != entry[i]]
        if not ch: break
        rounds += 1
        inch = set(ch)
        newex = list(ex); newentry = list(entry); longest = 0
        # 'first': only the leftmost run of changed segments walks on (everything to its left is final, so it carries the
        # TRUE state); 'late': every run, but only once at most `late` segments changed
        first_run_end = ch[0]
        while first_run_end + 1 in inch: first_run_end += 1
        for i in ch:
            newentry[i] = ex[i - 1]
            e, n = run_count(dec, *newentry[i], end(i)); lane = n; total += n
            j = i
            may = walk == 'all' or (walk == 'first' and i == first_run_end) or (walk == 'late' and len(ch) <= late)
            while may and e != ex[j] and j + 1 < nseg and (j + 1) not in inch:
                newex[j] = e
                j += 1
                newentry[j] = e
                e, n = run_count(dec, *newentry[j], end(j)); lane += n; total += n
            newex[j] = e
            longest = max(longest, lane)
        ex = newex; entry = newentry
        path += longest
    return rounds, path, total, cost0, nseg

if __name__ == '__main__':
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, 'tests', 'golden', 'sample-images1', '*.jpg')))[::5][:16]
    tot = [0, 0, 0, 0, 0]
    for f in files:
        d = Dec(open(f, 'rb').read())
        a = simulate(d, False)
        vs = [simulate(d, 'all'), simulate(d, 'first'), simulate(d, 'late', late=8), simulate(d, 'late', late=64)]
        tot[0] += a[1]
        for (k, b) in enumerate(vs): tot[k + 1] += b[1]
        print('%-28s nseg %3d | rounds %3d path %5d | ' % (os.path.basename(f), a[4], a[0], a[1]) +
              ' | '.join('%s: %3d r %5d (%.2f)' % (nm, b[0], b[1], b[1] / max(a[1], 1)) for (nm, b) in zip(('all', 'first', 'late8', 'late64'), vs)))
    print('sum of the paths: %d | ' % tot[0] + ' | '.join('%d (%.2f)' % (t, t / tot[0]) for t in tot[1:]))
